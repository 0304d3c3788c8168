"""Development: the relative-position attention backward (prep + keys-fixed + queries-fixed kernels) on the training step's shape,
under rocprofv3 or stand-alone (HIP events around the three launches)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindaudio_amd.train import kernels as K  # noqa: E402

b, t2, h, dk = 40, 255, 4, 64
g = torch.Generator().manual_seed(0)
bf = lambda x: x.to(torch.bfloat16).cuda()  # noqa: E731
qkv = bf(torch.randn(b * t2, 768, generator=g))
pos = bf(torch.randn(t2, 256, generator=g))
u, v = torch.randn(h, dk, generator=g).cuda() * 0.1, torch.randn(h, dk, generator=g).cuda() * 0.1
lens = torch.randint(180, t2 + 1, (b,), generator=g)
mask = (torch.arange(t2)[None, :] < lens[:, None]).float().cuda()
ctx, lse = K.attention_fwd(qkv, pos, u, v, mask, b, t2)
dctx = bf(torch.randn(b * t2, 256, generator=g))
dpos = torch.zeros(t2, 256, device="cuda")
du, dv = torch.zeros(h, dk, device="cuda"), torch.zeros(h, dk, device="cuda")
fn = lambda: K.attention_bwd(qkv, pos, u, v, mask, ctx, dctx, lse, b, t2, dpos, du, dv)  # noqa: E731
for _ in range(5):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    fn()
e1.record()
torch.cuda.synchronize()
print("attention backward (prep + dK'/dV + dQ' + two small reductions): %.1f us" % (e0.elapsed_time(e1) * 1000 / 50))
