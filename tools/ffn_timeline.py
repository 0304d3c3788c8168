"""Phase timeline of the FFN pair (+ qkv) launch from a -DMA_FFN_PROF build (tools/ffn_variants.sh build "prof:-DMA_FFN_PROF"):
wall_clock64 stamps (100 MHz) of wave 0 of three workgroups."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mindaudio_amd import _lib, ops

lib = _lib.load()
m, hid, dev = 64 * 249, 2048, "cuda"
r = lambda *sh: torch.randn(*sh, device=dev)
ln = (torch.ones(256, device=dev), torch.zeros(256, device=dev))
x = r(m, 256)
pq, bq = ops.ffn_qkv_pack((r(768, 256) / 16).bfloat16()), r(768)
pa = ops.ffn_pack_weights((r(hid, 256) / 16).bfloat16(), (r(256, hid) / 45).bfloat16())
pb = ops.ffn_pack_weights((r(hid, 256) / 16).bfloat16(), (r(256, hid) / 45).bfloat16())
b1, b2 = r(hid), r(256)
fn = lambda: ops.ffn_packed_pair(pa, b1, b2, pb, b1, b2, x, ln, ln, ln, ln, qkv=(pq, bq))
for _ in range(20):
    fn()
torch.cuda.synchronize()
names = {0: "s0 start", 1: "s0 staged", 2: "s0 prologue done", 3: "s0 loop done", 4: "s0 tile free", 5: "s0 reduced", 6: "s0 epilogue done",
         10: "s1 start", 11: "s1 barrier", 12: "s1 prologue done", 13: "s1 loop done", 14: "s1 tile free", 15: "s1 reduced", 16: "s1 epilogue done",
         26: "tail start", 27: "tail loop done", 28: "end"}
acc = {}
N = 20
for it in range(N):
    fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 96)()
    lib.ma_debug_ffn_prof.argtypes = [ctypes.c_void_p]
    assert lib.ma_debug_ffn_prof(buf) == 0
    a = np.array(buf[:], dtype=np.int64).reshape(3, 32)
    t0 = a[:, 0].min()
    for w in range(3):
        for k in names:
            acc.setdefault((w, k), []).append((a[w, k] - t0) / 100.0)
print("%-20s %10s %10s %10s   (us since the first workgroup's start; median of %d launches; +delta)" % ("phase", "wg 0", "wg 97", "wg 248", N))
prev = [0, 0, 0]
for k in sorted(names):
    med = [float(np.median(acc[(w, k)])) for w in range(3)]
    print("%-20s " % names[k] + " ".join("%6.2f(+%5.2f)" % (med[w], med[w] - prev[w]) for w in range(3)))
    prev = med
