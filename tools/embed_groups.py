#!/usr/bin/env python3
"""Front end of the encoder (conv1 -> conv2 per utterance group, then the 4864 -> 256 embed layer): the embed layer once over the
whole batch (production) against one embed launch per group right behind the group's conv2 (its operand still in the Infinity Cache)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd import ops
from mindaudio_amd.models import ConformerEncoder
B, T = 64, 1000
torch.manual_seed(0)
enc = ConformerEncoder(80, 256, 4, 2048, 12).eval().cuda().prepare()
P = enc._prepared
xs = torch.randn(B, T, 80, device="cuda")
sizes = [24, 20, 20]
t1, f1, t2, f2, c = 499, 39, 249, 19, 256
act1 = torch.empty((max(sizes), t1, f1, c), dtype=torch.bfloat16, device="cuda")
act2 = torch.empty((B, t2, f2, c), dtype=torch.bfloat16, device="cuda")
x = torch.empty((B * t2, 256), dtype=torch.float32, device="cuda")
def front(per_group):
    i = 0
    for n in sizes:
        ops.subsample_conv1(xs[i:i + n], P["conv1_w"], P["conv1_b"], enc.cmvn_mean, enc.cmvn_istd, out=act1[:n])
        ops.conv2d_3x3s2_packed(act1[:n], P["conv2_pk"], P["conv2_b"], relu=True, out=act2[i:i + n])
        if per_group:
            ops.gemm_rows_packed(act2[i:i + n].view(n * t2, f2 * c), P["out_pk"], P["out_b"], alpha=16.0, out=x[i * t2:(i + n) * t2])
        i += n
    if not per_group:
        ops.gemm_rows_packed(act2.view(B * t2, f2 * c), P["out_pk"], P["out_b"], alpha=16.0, out=x)
def timeit(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for r in range(3):
    print("embed once %.1f us   embed per group %.1f us" % (timeit(lambda: front(False)), timeit(lambda: front(True))))
