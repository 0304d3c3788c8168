import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd import ops
act = torch.randn(64, 499, 39, 256, device="cuda").bfloat16(); w = (torch.randn(256, 3, 3, 256, device="cuda") / 48).bfloat16(); b = torch.randn(256, device="cuda")
pk = ops.conv2d_3x3s2_pack(w)
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
fl = 2.0 * 64 * 249 * 19 * 256 * 2304
u0 = t(lambda: ops.conv2d_3x3s2_nhwc(act, w, b)); u1 = t(lambda: ops.conv2d_3x3s2_packed(act, pk, b))
print("conv2 general %.1f us (%.0f TF/s), packed %.1f us (%.0f TF/s)" % (u0, fl / u0 / 1e6, u1, fl / u1 / 1e6))
