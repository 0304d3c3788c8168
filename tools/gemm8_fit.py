#!/usr/bin/env python3
"""Fixed cost per 256 x 256 tile of the 8-phase GEMM: one resident round (4096 x 4096 = 256 tiles) at K = 1024 .. 8192 -> T = a + b K."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd import ops
def timeit(fn, reps=30, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
pts = []
for rounds in (1, 2, 4):
    for k in (1024, 2048, 4096, 8192):
        m, n = 4096 * rounds, 4096
        a = torch.randn(m, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).bfloat16()
        o = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        us = timeit(lambda: ops.gemm(a, w, out=o))
        pts.append((rounds, k, us))
        print("rounds %d K %5d: %8.1f us  %7.1f TF/s  per round %.1f us" % (rounds, k, us, 2.0 * m * n * k / us / 1e6, us / rounds))
