"""Whole encoder forward (B = 64 x 1000 x 80) against the utterance grouping of the conv1 -> conv2 front end, interleaved rounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd.models import ConformerEncoder
B, T = 64, 1000
torch.manual_seed(0)
enc = ConformerEncoder(80, 256, 4, 2048, 12).eval().cuda().prepare()
xs = torch.randn(B, T, 80, device="cuda"); masks = torch.ones(B, 1, 249, device="cuda")
def timeit(reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): enc(xs, masks)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
variants = [None, [24, 20, 20], [20, 20, 20, 4], [4, 20, 20, 20], [16, 16, 16, 16], [20, 20, 12, 12], [20, 22, 22], [13, 13, 13, 13, 12], [10, 10, 11, 11, 11, 11]]
for rnd in range(3):
    for v in variants:
        enc.subsample_group = v
        for _ in range(3): enc(xs, masks)
        torch.cuda.synchronize()
        print("round %d  groups %-22s %.4f ms" % (rnd, v if v else "default", timeit()), flush=True)
