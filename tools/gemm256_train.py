#!/usr/bin/env python3
"""K = 256 layers at the training step's M = 10 200: general GEMM kernel against the packed K = 256 kernel of the evaluation path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd import ops
m = 10200
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(4): g.replay()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / (4 * reps) * 1e3
a = torch.randn(m, 256, device="cuda").bfloat16()
for n in (2048, 768, 512, 256):
    w = (torch.randn(n, 256, device="cuda") / 16).bfloat16(); b = torch.randn(n, device="cuda"); pk = ops.gemm_k256_pack(w)
    o = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    u0 = t(lambda: ops.gemm(a, w, bias=b, out=o)); u1 = t(lambda: ops.gemm_packed(a, pk, bias=b, out=o))
    up = t(lambda: ops.gemm_k256_pack(w))
    print("N=%4d: general %.1f us, packed %.1f us (pack %.1f us)" % (n, u0, u1, up))
# the K = 2048 -> 256 layers (w_2 forward, dX of w_1): general kernel against rows_packed (64 rows x 256 outputs, weights straight to registers)
import math
k, n = 2048, 256
a2 = torch.randn(m, k, device="cuda").bfloat16(); w2 = (torch.randn(n, k, device="cuda") / math.sqrt(k)).bfloat16(); b2 = torch.randn(n, device="cuda")
pk2 = ops.gemm_rows_pack(w2)
o2 = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
print("K=2048 N=256: general %.1f us (f32 out %.1f us), rows_packed %.1f us, pack %.1f us" % (
    t(lambda: ops.gemm(a2, w2, bias=b2, out=o2)), t(lambda: ops.gemm(a2, w2, bias=b2, out_dtype=torch.float32)),
    t(lambda: ops.gemm_rows_packed(a2, pk2, b2)), t(lambda: ops.gemm_rows_pack(w2))))
