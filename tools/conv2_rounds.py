"""conv2_packed launch time against the number of utterances (= workgroups): the staircase of resident rounds (512 workgroups each)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd import ops
w = (torch.randn(256, 3, 3, 256, device="cuda") / 48).bfloat16(); b = torch.randn(256, device="cuda")
pk = ops.conv2d_3x3s2_pack(w)
full = torch.randn(28, 499, 39, 256, device="cuda").bfloat16()
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
for n in (8, 12, 13, 14, 16, 18, 20, 22, 24, 26, 27, 28):
    act = full[:n]
    us = t(lambda: ops.conv2d_3x3s2_packed(act, pk, b))
    rows = n * 249 * 19
    wgs = (rows + 127) // 128
    print("%2d utterances: %6d rows, %4d workgroups = %.2f rounds: %.1f us  (%.0f TFLOP/s)" % (n, rows, wgs, wgs / 512, us, 2.0 * rows * 256 * 2304 / us / 1e6))
