#!/usr/bin/env python3
"""The packed-FMA hazard scan of tools/check_pk_hazard.py over libraries that are NOT ours but run beside our MFMA grids in the
training step: RCCL's reduction kernels (N > 1: on their own hardware queue while the backward pass runs) and the torch HIP kernels
the step still launches (a fill, an elementwise add, a reduce: profiles/r05_train_step_census.txt).

    python tools/pk_hazard_scan_external.py [--out profiles/r06_pk_hazard_scan.txt] [lib.so ...]

Default libraries: torch's librccl.so and libtorch_hip.so.  Their device code is a `.hip_fatbin` section of COMPRESSED offload
bundles ("CCOB", zstd): every bundle is cut out by the size in its header, unbundled for gfx950 with clang-offload-bundler, and the
code object is disassembled as a stream (the text of librccl's gfx950 object is several GB: never held in memory).  Per library and
kernel the scan counts

    pk_fma        every v_pk_fma_f32
    hazard        v_pk_fma_f32 vD, vA, vD, vC op_sel:[_,1,_]   - the form tools/ubench/two_queue_pk.hip shows miscomputing lanes
                  48-63 beside another queue's MFMA waves (DESIGN: two-queue hazard)
    wide          ANY v_pk_{fma,mul,add}_f32 whose destination pair is also a source pair read with op_sel 1 (hi -> lo) - the wider
                  class ADVICE r5 asks about; the micro-benchmark measured the mul / add / dst = src0 / dst = src2 members exact, so
                  these are listed, not failed on

Exit status 1 if any `hazard` is found in a kernel on the watch list (the kernels that appear in our step census / RCCL's
all-reduce), 0 otherwise; tests/test_cabi_cpu.py::test_external_pk_hazard_scan_is_current_and_clean checks the committed summary
against the installed libraries' sizes and asserts it is clean."""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile
import time

LLVM = "/opt/rocm/lib/llvm/bin"
RAW = b"__CLANG_OFFLOAD_BUNDLE__"
PK = re.compile(r"\bv_pk_(fma|mul|add)_f32\s+v\[(\d+):\d+\],\s*(\S+),\s*(\S+?)(?:,\s*(\S+?))?(\s.*)?$")
REG = re.compile(r"v\[(\d+):\d+\]")
SEL = re.compile(r"op_sel:\[([01,]+)\]")


def default_libs():
    import importlib.util

    spec = importlib.util.find_spec("torch")
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    return [os.path.join(d, "librccl.so"), os.path.join(d, "libtorch_hip.so")]


def bundles(blob):
    """Yield (kind, bytes) for every offload bundle in a .hip_fatbin section: 'ccob' (compressed) or 'raw'."""
    pos, n = 0, len(blob)
    while pos < n:
        a, b = blob.find(b"CCOB", pos), blob.find(RAW, pos)
        if a < 0 and b < 0:
            return
        if b >= 0 and (a < 0 or b < a):
            nxt = min([x for x in (blob.find(b"CCOB", b + 1), blob.find(RAW, b + 1)) if x >= 0] or [n])
            yield "raw", blob[b:nxt]
            pos = nxt
            continue
        ver = struct.unpack_from("<H", blob, a + 4)[0]
        if ver >= 3:
            size = struct.unpack_from("<Q", blob, a + 8)[0]
        else:
            size = struct.unpack_from("<I", blob, a + 8)[0]
        if size <= 0 or a + size > n:
            pos = a + 4
            continue
        yield "ccob", blob[a:a + size]
        pos = a + size


def gfx950_objects(lib, td):
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    os.unlink(fat)
    bundler = os.path.join(LLVM, "clang-offload-bundler")
    for k, (kind, data) in enumerate(bundles(blob)):
        src = os.path.join(td, "b%d.bin" % k)
        with open(src, "wb") as f:
            f.write(data)
        ls = subprocess.run([bundler, "--list", "--type=o", "--input=" + src], capture_output=True, text=True)
        targets = [t for t in ls.stdout.split() if "gfx950" in t]
        for t in targets:
            out = os.path.join(td, "b%d.co" % k)
            r = subprocess.run([bundler, "--unbundle", "--type=o", "--targets=" + t, "--input=" + src, "--output=" + out],
                               capture_output=True, text=True)
            if r.returncode == 0 and os.path.exists(out) and os.path.getsize(out):
                yield out
                os.unlink(out)
        os.unlink(src)


def classify(line):
    """(is_pk_fma, is_hazard, is_wide) of one disassembly line."""
    m = PK.search(line.split("//")[0].rstrip())
    if not m:
        return False, False, False
    op, d0 = m.group(1), int(m.group(2))
    srcs = [m.group(3), m.group(4)] + ([m.group(5)] if m.group(5) else [])
    mods = m.group(6) or ""
    sel = SEL.search(mods)
    sels = [int(x) for x in sel.group(1).split(",")] if sel else [0] * len(srcs)
    hazard = wide = False
    for i, s in enumerate(srcs):
        r = REG.match(s.rstrip(","))
        if r and int(r.group(1)) == d0 and i < len(sels) and sels[i] == 1:
            wide = True
            if op == "fma" and i == 1:
                hazard = True
    return op == "fma", hazard, wide


def kernel_symbols(co):
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", co], capture_output=True, text=True).stdout
    syms = []
    for ln in out.splitlines():  # Num: Value Size Type Bind Vis Ndx Name
        f = ln.split()
        if len(f) >= 8 and f[3] == "FUNC" and f[6] != "UND":
            syms.append(f[7])
    return syms


def scan(lib, log=None, only_watched=False):
    """{kernel: [pk_fma, hazard, wide, examples]}, number of gfx950 code objects, number of kernels looked at.  only_watched:
    disassemble just the watch-list kernels (seconds instead of minutes: what the test suite runs)."""
    per_kernel, n_obj, n_kern, t0 = {}, 0, 0, time.time()
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        for co in gfx950_objects(lib, td):
            n_obj += 1
            cmd = [os.path.join(LLVM, "llvm-objdump"), "-d", co]
            syms = kernel_symbols(co)
            if only_watched:
                dm = demangle(syms)
                syms = [k for k in syms if watched(dm.get(k, k))]
                if not syms:
                    continue
                cmd.insert(2, "--disassemble-symbols=" + ",".join(syms))
            n_kern += len(syms)
            pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, bufsize=1 << 20)
            kernel = None
            for line in pr.stdout:
                if line[:1] in "0123456789abcdef" and line.rstrip().endswith(">:"):
                    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                    if m:
                        kernel = m.group(1)
                    continue
                if "v_pk_" not in line:
                    continue
                f, h, w = classify(line)
                if f or w:
                    c = per_kernel.setdefault(kernel, [0, 0, 0, []])
                    c[0] += f
                    c[1] += h
                    c[2] += w
                    if (h or w) and len(c[3]) < 3:
                        c[3].append(line.split("//")[0].strip())
            pr.wait()
            if log and n_obj % 20 == 0:
                log("  %s: %d gfx950 code objects, %.0f s" % (os.path.basename(lib), n_obj, time.time() - t0))
    return per_kernel, n_obj, n_kern


# Kernels of these libraries that run inside / beside our training step: RCCL's device kernels (every protocol and data type: they
# all live in a few generic entry points) and the torch kernels of the step census (profiles/r05_train_step_census.txt and
# _hybrid.txt): float fill, float unary / binary add / mul, the float sum reduction, the float / integer copy-cast kernels.  Complex,
# half and double instantiations of the same templates are not launched by the step and are not on the list.
WATCH_RCCL = re.compile(r"ncclDevKernel|ncclDevFunc|oneRankReduce|rcclDev|ncclKernel|MSCCL|mscclKernel", re.I)
WATCH_TORCH = re.compile(r"FillFunctor<float>|CUDAFunctor_add<float>|AUnaryFunctor<float, float, float|BUnaryFunctor<float, float, float|"
                         r"BinaryFunctor<float, float, float|reduce_kernel<512, 1, at::native::ReduceOp<float, at::native::func_wrapper_t<float, "
                         r"at::native::sum_functor|elementwise_kernel_manual_unroll<128, 4|direct_copy_kernel|"
                         r"bfloat16_copy_kernel|float_copy|LoadWithCast|unrolled_elementwise_kernel<at::native::direct_copy")


def watched(name):
    if "complex" in name or "c10::Half" in name or "double" in name:
        return bool(WATCH_RCCL.search(name))
    return bool(WATCH_RCCL.search(name) or WATCH_TORCH.search(name))


def demangle(names):
    if not names:
        return {}
    try:
        exe = next(e for e in (os.path.join(LLVM, "llvm-cxxfilt"), "/usr/bin/c++filt", "c++filt") if e == "c++filt" or os.path.exists(e))
        out = subprocess.run([exe], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except (OSError, StopIteration):
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--out")
    ap.add_argument("--only-watched", action="store_true", help="disassemble only the watch-list kernels (fast)")
    a = ap.parse_args()
    libs = a.libs or default_libs()
    lines, bad = [], 0
    lines.append("packed-FMA hazard scan outside libmindaudio_amd.so (tools/pk_hazard_scan_external.py)")
    lines.append("form `hazard`: v_pk_fma_f32 vD, vA, vD, vC op_sel:[_,1,_]; `wide`: any v_pk_{fma,mul,add}_f32 with dst pair == a source pair "
                 "read with op_sel 1")
    for lib in libs:
        log = lambda m: print(m, file=sys.stderr, flush=True)  # noqa: E731
        log("scanning %s" % lib)
        per_kernel, n_obj, n_kern = scan(lib, log, a.only_watched)
        dm = demangle([k for k in per_kernel if k])
        tot = [sum(c[i] for c in per_kernel.values()) for i in range(3)]
        watched_k = {k: c for k, c in per_kernel.items() if k and watched(dm.get(k, k))}
        wtot = [sum(c[i] for c in watched_k.values()) for i in range(3)]
        lines.append("")
        lines.append("library %s  size %d bytes  gfx950 code objects %d  kernels disassembled %d%s" % (
            os.path.basename(lib), os.path.getsize(lib), n_obj, n_kern, "  (watch list only)" if a.only_watched else ""))
        lines.append("  all kernels:      v_pk_fma_f32 %d   hazard %d   wide %d   (kernels with a packed fp32 op: %d)" % (tot[0], tot[1], tot[2], len(per_kernel)))
        lines.append("  watched kernels:  v_pk_fma_f32 %d   hazard %d   wide %d   (kernels with a packed fp32 op: %d)" % (wtot[0], wtot[1], wtot[2], len(watched_k)))
        haz = sorted(((k, c) for k, c in per_kernel.items() if c[1]), key=lambda kc: -kc[1][1])
        if haz:
            lines.append("  kernels with the hazardous form (all of them):")
        for k, c in haz:
            name = dm.get(k, k) or "?"
            lines.append("    %s%s  x%d: %s" % ("WATCHED " if k in watched_k else "", name[:170], c[1], c[3][0] if c[3] else ""))
        wide = sorted(((k, c) for k, c in per_kernel.items() if c[2] and not c[1]), key=lambda kc: -kc[1][2])
        if wide:
            lines.append("  kernels with only the wider class (measured exact by tools/ubench/two_queue_pk.hip; first 12 of %d):" % len(wide))
        for k, c in wide[:12]:
            name = dm.get(k, k) or "?"
            lines.append("    %s%s  x%d: %s" % ("WATCHED " if k in watched_k else "", name[:170], c[2], c[3][0] if c[3] else ""))
        bad += wtot[1]
    lines.append("")
    lines.append("verdict: %s" % ("CLEAN - no kernel on the watch list contains the hazardous form" if bad == 0 else
                                  "%d hazardous instructions in watched kernels" % bad))
    text = "\n".join(lines) + "\n"
    sys.stdout.write(text)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
