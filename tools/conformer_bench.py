#!/usr/bin/env python3
"""Time the Conformer-small encoder forward (eager launches vs HIP-graph replay)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd.models import ConformerEncoder
B = int(os.environ.get("B", 32)); T = 1000
torch.manual_seed(0)
enc = ConformerEncoder(80, 256, 4, 2048, 12).eval().cuda().prepare()
xs = torch.randn(B, T, 80, device="cuda"); masks = torch.ones(B, 1, 249, device="cuda")
def run(): return enc(xs, masks)[0]
for _ in range(3): run()
torch.cuda.synchronize()
def timeit(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timeit(run)
print("eager:  %.3f ms / batch of %d -> %.0f utt/s, %.1f TFLOP/s" % (ms, B, B / ms * 1e3, 23.12e9 * B / ms / 1e9))
if "--train" in sys.argv:  # cfg 3 in train mode: BatchNorm batch statistics + running-stat update, dropout 0.1 (forward only)
    enc.train()
    for _ in range(3): run()
    torch.cuda.synchronize()
    ms = timeit(run)
    print("train-mode forward:  %.3f ms / batch of %d -> %.0f utt/s" % (ms, B, B / ms * 1e3))
    enc.eval()
if "--graph" in sys.argv:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = run()
    ms = timeit(g.replay)
    print("graph:  %.3f ms / batch of %d -> %.0f utt/s, %.1f TFLOP/s" % (ms, B, B / ms * 1e3, 23.12e9 * B / ms / 1e9))
