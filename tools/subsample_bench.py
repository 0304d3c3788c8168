"""conv1 / conv2 of the subsampling front end at the north-star shape: full batch vs utterance chunks that reuse one
conv1-output buffer (does the 256 MB Infinity Cache keep the 637 MB intermediate off HBM?)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd import ops

B, T, F = 64, 1000, 80
feats = torch.randn(B, F, T + 1, device="cuda").transpose(1, 2)[:, :T]
w1 = (torch.randn(256, 9, device="cuda") * 0.3).contiguous()
b1 = torch.randn(256, device="cuda") * 0.1
w2 = (torch.randn(256, 3, 3, 256, device="cuda") / 48).bfloat16()
b2 = torch.randn(256, device="cuda")
pk = ops.conv2d_3x3s2_pack(w2)


def t(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


u1 = t(lambda: ops.subsample_conv1(feats, w1, b1))
act = ops.subsample_conv1(feats, w1, b1)
u2 = t(lambda: ops.conv2d_3x3s2_packed(act, pk, b2))
both = t(lambda: ops.conv2d_3x3s2_packed(ops.subsample_conv1(feats, w1, b1), pk, b2))
print("conv1 %.1f us (%.2f TB/s written)  conv2 %.1f us  conv1+conv2 %.1f us" % (u1, act.numel() * 2 / u1 / 1e6, u2, both))
spk = ops.subsample_fused_pack(w1, w2, F)
uf = t(lambda: ops.subsample_fused(feats, spk, b1, b2))
fl = 2.0 * B * 249 * 19 * 256 * 2304
print("fused conv1+conv2 (one launch, act1 in LDS): %.1f us  (conv2's %.0f GFLOP at %.0f TFLOP/s)" % (uf, fl / 1e9, fl / uf / 1e6))
if "--fused-only" in sys.argv:
    sys.exit(0)
for chunk in (32, 26, 24, 22, 20, 18, 16, 13, 11, 8):
    def run():
        outs = []
        for i in range(0, B, chunk):
            outs.append(ops.conv2d_3x3s2_packed(ops.subsample_conv1(feats[i:i + chunk], w1, b1), pk, b2))
        return outs
    print("chunks of %d utterances: %.1f us" % (chunk, t(run)))
