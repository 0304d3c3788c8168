#!/usr/bin/env python3
"""Tile-shape scan of the general GEMM kernel on the shapes of the training step and of ECAPA (build variants with
tools/lib_variant.sh t3 "-DMA_GEMM_FORCE=3" gemm_bf16.hip ... and run with MINDAUDIO_AMD_LIB=...)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd import ops, _lib
def timeit(fn, reps=40, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
out = []
for (m, n, k) in [(10200, 2048, 256), (10200, 256, 2048), (10200, 768, 256), (10200, 256, 768), (10200, 256, 256), (10200, 512, 256),
                  (10200, 256, 512), (76800, 512, 512), (76800, 128, 1536), (76800, 1536, 1536), (15936, 2048, 256), (15936, 256, 2048), (15936, 256, 4864), (7968, 256, 4864), (7968, 256, 2048)]:
    a = torch.randn(m, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).bfloat16()
    bias = torch.randn(n, device="cuda")
    out.append("%.1f" % timeit(lambda: ops.gemm(a, w, bias=bias, act=_lib.ACT_RELU)))
print(" ".join(x.rjust(7) for x in out))
