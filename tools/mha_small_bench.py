"""Development: the decoder's small attention (forward / backward) at the hybrid step's two shapes: self-attention over the labels
(Lq = Lk = 31, causal + padding mask) and source attention over the encoder memory (Lq = 31, Lk = 255, padding mask)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindaudio_amd.train import kernels as K  # noqa: E402

b, h, dk = 40, 4, 64
g = torch.Generator().manual_seed(0)
bf = lambda x: x.to(torch.bfloat16).cuda()  # noqa: E731


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


for name, lq, lk, mode in (("self", 31, 31, 2), ("source", 31, 255, 1)):
    q = bf(torch.randn(b * lq, 256, generator=g))
    k = bf(torch.randn(b * lk, 256, generator=g))
    v = bf(torch.randn(b * lk, 256, generator=g))
    mask = torch.ones((b, lq, lk) if mode == 2 else (b, lk)).cuda()
    if mode == 2:
        mask = torch.tril(mask)
    ctx, probs = K.mha_small_fwd(q, k, v, mask, mode, b, lq, lk, 0.125)
    dctx = bf(torch.randn(b * lq, 256, generator=g))
    dq, dkk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    tf = timeit(lambda: K.mha_small_fwd(q, k, v, mask, mode, b, lq, lk, 0.125))
    tb = timeit(lambda: K.mha_small_bwd(q, k, v, probs, ctx, dctx, b, lq, lk, 0.125, dq, dkk, dv))
    print("%-6s Lq %d Lk %3d: forward %6.1f us   backward %6.1f us" % (name, lq, lk, tf, tb))
