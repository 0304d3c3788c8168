"""Phase timeline of ffn_pc_kernel (pair + qkv form) from a -DMA_FFN_PROF build of ffn_pc.hip
(SRC=ffn_pc.hip bash tools/ffn_variants.sh build "prof:-DMA_FFN_PROF"; run with MINDAUDIO_AMD_LIB=.../variants/prof.so):
s_memtime stamps of O-wave 0 and S-wave 4 of three workgroups, cycles since the wave's entry."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mindaudio_amd import _lib, ops

lib = _lib.load()
m, d, hid = 64 * 249, 256, 2048
r = lambda *sh: torch.randn(*sh, device="cuda")
packs = [(ops.ffn_pack_weights((r(hid, d) / 16).bfloat16(), (r(d, hid) / 45).bfloat16()), r(hid) * 0.3, r(d) * 0.3) for _ in range(12)]
lns = [(1 + 0.1 * r(d), 0.1 * r(d)) for _ in range(4)]
pq, bq = ops.ffn_qkv_pack((r(768, d) / 16).bfloat16()), r(768) * 0.3
x = r(m, d)
k = [0]


def pair():
    a, b = packs[k[0] % 12], packs[(k[0] + 1) % 12]
    k[0] += 2
    return ops.ffn_packed_pair(a[0], a[1], a[2], b[0], b[1], b[2], x, lns[0], lns[1], lns[2], lns[3], qkv=(pq, bq))


for _ in range(6):
    pair()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40):
    pair()
e1.record()
torch.cuda.synchronize()
print("launch %.1f us (back to back, instrumented build)" % (e0.elapsed_time(e1) / 40 * 1e3))
lib.ma_debug_ffn_pc_prof.argtypes = [ctypes.c_void_p]
acc = []
for it in range(10):
    pair()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 192)()
    assert lib.ma_debug_ffn_pc_prof(buf) == 0
    acc.append(np.array(buf[:], dtype=np.int64).reshape(2, 3, 32))
a = np.median(np.stack(acc), axis=0)
labels = {0: "entry", 1: "tile staged", 2: "stage 0 loop start", 5: "  period 4: top", 6: "  S: job done / O: barrier passed", 7: "  S: barrier passed / O: period done",
          8: "  S: job done / O: barrier passed", 9: "  S: barrier passed / O: period done", 3: "stage 0 loop end", 4: "stage 0 epilogue end", 10: "stage 1 loop start",
          11: "stage 1 loop end", 12: "stage 1 epilogue end", 20: "role done", 21: "tail done"}
for role, name in ((0, "O-wave 0"), (1, "S-wave 4")):
    print("== %s: cycles since entry, workgroups 0 / 97 / 248" % name)
    prev = None
    for kk in (0, 1, 2, 5, 6, 7, 8, 9, 3, 4, 10, 11, 12, 20, 21):
        if a[role, 0, kk] == 0:
            continue
        v = a[role, :, kk] - a[role, :, 0]
        dd = "" if prev is None else "  (+%s)" % " / ".join("%6d" % int(q) for q in (v - prev))
        print("%-38s %s%s" % (labels[kk], " / ".join("%7d" % int(q) for q in v), dd))
        prev = v
