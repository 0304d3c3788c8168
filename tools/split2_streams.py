#!/usr/bin/env python3
"""Experiment: the encoder forward of one batch as two half-batches on two HIP streams (inter-launch gaps of one chain filled by
the other) against the single chain.  B=64 T=1000."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd.models import ConformerEncoder
B = int(os.environ.get("B", 64)); T = 1000
NS = int(os.environ.get("NS", 2))
torch.manual_seed(0)
enc = ConformerEncoder(80, 256, 4, 2048, 12).eval().cuda().prepare()
xs = torch.randn(B, T, 80, device="cuda"); masks = torch.ones(B, 1, 249, device="cuda")
streams = [torch.cuda.Stream() for _ in range(NS)]
cuts = [B * i // NS for i in range(NS + 1)]
def run1(): return enc(xs, masks)[0]
def run2():
    cur = torch.cuda.current_stream()
    outs = []
    for s, lo, hi in zip(streams, cuts[:-1], cuts[1:]):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(enc(xs[lo:hi], masks[lo:hi])[0])
    for s in streams:
        cur.wait_stream(s)
    return outs
def timeit(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
a = run1(); b = torch.cat(run2(), 0); torch.cuda.synchronize()
print("equal:", torch.equal(a, b), float((a - b).abs().max()))
for _ in range(2):
    print("one chain: %.3f ms   %d chains: %.3f ms" % (timeit(run1), NS, timeit(run2)))
