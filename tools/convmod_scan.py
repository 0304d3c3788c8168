"""convmodule_kernel launch time: kernel size 15 vs 3 (depthwise/VALU share) and batch 64 vs 32 (one resident round vs half)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd import ops

dev = "cuda"
r = lambda *sh: torch.randn(*sh, device=dev)


def t(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


for B in (64, 32):
    for ks in (15, 3):
        T = 249
        a = r(B * T, 256).bfloat16()
        p1, p2 = ops.gemm_k256_pack((r(512, 256) / 16).bfloat16()), ops.gemm_k256_pack((r(256, 256) / 16).bfloat16())
        b1, b2 = r(512), r(256)
        dw = r(256, ks) * 0.3
        sc, sh = 1 + 0.1 * r(256), 0.1 * r(256)
        x = r(B * T, 256)
        mask = torch.ones(B * T, device=dev)
        us = t(lambda: ops.convmodule(a, p1, b1, dw, sc, sh, p2, b2, mask, x, B, T))
        wo = ops.gemm_k256_pack((r(256, 256) / 16).bfloat16())
        lg, lb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
        us2 = t(lambda: ops.gemm_packed_ln(a, wo, lg, lb, ln_row_scale=mask, bias=b2, residual=x, out=x))
        print("B=%d ks=%d: convmodule %.1f us   out-projection + LN %.1f us" % (B, ks, us, us2))
