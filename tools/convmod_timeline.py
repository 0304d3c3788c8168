"""Phase timeline of convmodule_kernel<OPROJ> from a -DMA_CM_PROF build of convmid_pw2.hip (tools/cm_variants.sh): wall_clock64
stamps (100 MHz) of wave 0 of three workgroups (first, middle, last)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mindaudio_amd import _lib, ops

lib = _lib.load()
dev = "cuda"
r = lambda *sh: torch.randn(*sh, device=dev)
B, T, ks = 64, 249, 15
ctx = r(B * T, 256).bfloat16()
p1, p2 = ops.gemm_k256_pack((r(512, 256) / 16).bfloat16()), ops.gemm_k256_pack((r(256, 256) / 16).bfloat16())
wo = ops.gemm_k256_pack((r(256, 256) / 16).bfloat16())
b1, b2, bo = r(512), r(256), r(256)
dw = r(256, ks) * 0.3
sc, sh = 1 + 0.1 * r(256), 0.1 * r(256)
lg, lb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
x = r(B * T, 256)
mask = torch.ones(B * T, device=dev)
xo = torch.empty_like(x)
fn = lambda: ops.attn_out_convmodule(ctx, wo, bo, lg, lb, p1, b1, dw, sc, sh, p2, b2, mask, x, B, T, out=xo)
for _ in range(10):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    fn()
e1.record()
torch.cuda.synchronize()
print("launch %.1f us (back to back, instrumented build)" % (e0.elapsed_time(e1) / 50 * 1e3))
names = ["entry", "ctx tile + taps staged", "out-projection MFMAs", "+ residual, LN statistics", "a-tile written", "pw1 + GLU (y tile)",
         "depthwise + BN + Swish", "z tile + barrier", "pw2 MFMAs", "end"]
acc = {}
N = 20
lib.ma_debug_cm_prof.argtypes = [ctypes.c_void_p]
for it in range(N):
    fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 48)()
    assert lib.ma_debug_cm_prof(buf) == 0
    a = np.array(buf[:], dtype=np.int64).reshape(3, 16)
    t0 = a[:, 0].min()
    for w in range(3):
        for k in range(10):
            acc.setdefault((w, k), []).append((a[w, k] - t0) / 100.0)
print("%-28s %14s %14s %14s   (us since the first start; median of %d; +delta)" % ("phase", "wg 0", "wg 200", "wg 511", N))
prev = [0, 0, 0]
for k in range(10):
    med = [float(np.median(acc[(w, k)])) for w in range(3)]
    print("%-28s " % names[k] + " ".join("%6.2f(+%5.2f)" % (med[w], med[w] - prev[w]) for w in range(3)))
    prev = med
