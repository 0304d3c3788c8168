#!/usr/bin/env python3
"""Scans the device code of the built library for the instruction form that gfx950 executes wrongly beside another queue's MFMA waves
(found in round 5, tools/ubench/two_queue_pk.hip):

    v_pk_fma_f32 vD, vA, vD, vC  op_sel:[_,1,_]      (destination pair == src1 pair, LOW result taken from src1's HIGH register)

Alone on its SIMD the instruction is exact; with MFMA-issuing waves of another kernel on the same SIMD the low results of lanes 48-63
are computed from the already written high result.  hipcc emits the form freely for float2 code that broadcasts the second element
of a pair.  `python tools/check_pk_hazard.py [lib.so]` prints every occurrence (kernel, instruction) and exits 1 if there is any;
tests/test_cabi_cpu.py runs it on the built library.

What is and is not flagged: tools/ubench/two_queue_pk.hip ran every member of the wider class "packed fp32 op whose
destination pair is also a source pair read hi -> lo" beside an MFMA grid - v_pk_fma_f32 with dst == src0 and dst == src2,
v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32 with the overlap, and dst == src1 with op_sel_hi instead of op_sel: 0 mismatches each;
only dst == src1 + op_sel[1] = 1 miscomputes (466 896 mismatches).  So only that form fails the check; `--wide` lists the other
members (tools/pk_hazard_scan_external.py has the classifier) for the record."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


PK = re.compile(r"\bv_pk_fma_f32\s+v\[(\d+):(\d+)\],\s*([^,]+),\s*v\[(\d+):(\d+)\],\s*[^ ]+(.*)")


def hazards(lib):
    found, n_pk = [], 0
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f.name], capture_output=True, text=True).stdout
        kernel = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                kernel = m.group(1)
                continue
            m = PK.search(line)
            if not m:
                continue
            n_pk += 1
            d0, s1_0, mods = int(m.group(1)), int(m.group(4)), m.group(6)
            sel = re.search(r"op_sel:\[(\d),(\d),(\d)\]", mods)
            if d0 == s1_0 and sel and sel.group(2) == "1":
                found.append((kernel, line.split("//")[0].strip()))
    return found, n_pk


def wide(lib):
    """[(kernel, instruction)] of the wider class (see the module docstring) - informational."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("pk_ext", os.path.join(os.path.dirname(os.path.abspath(__file__)), "pk_hazard_scan_external.py"))
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    out = []
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f.name], capture_output=True, text=True).stdout
        kernel = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                kernel = m.group(1)
            elif "v_pk_" in line:
                _, h, w = ext.classify(line)
                if w and not h:
                    out.append((kernel, line.split("//")[0].strip()))
    return out


def main():
    if "--wide" in sys.argv:
        sys.argv.remove("--wide")
        lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mindaudio_amd",
                                                                  "lib", "libmindaudio_amd.so")
        hits = wide(lib)
        per = {}
        for k, _ in hits:
            per[k] = per.get(k, 0) + 1
        for k, n in sorted(per.items(), key=lambda kv: -kv[1]):
            print("%6d  %s" % (n, k))
        print("%d instructions of the wider (measured-exact) class in %d kernels" % (len(hits), len(per)))
        return 0
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mindaudio_amd",
                                                              "lib", "libmindaudio_amd.so")
    found, n_pk = hazards(lib)
    for k, ins in found:
        print("%s: %s" % (k, ins))
    print("%d v_pk_fma_f32 in %s, %d of the hazardous form" % (n_pk, os.path.basename(lib), len(found)))
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
