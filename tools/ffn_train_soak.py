"""Development: soak test of the one-launch feed-forward training kernels - the same launch N times, every output compared bit for bit
with the first one's (a fragment consumed before it landed, a lost store or an exchange race would show as a changed bit).
    python tools/ffn_train_soak.py [--iters 3000] [--m 10200]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindaudio_amd import ops  # noqa: E402
from mindaudio_amd.train import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=3000)
ap.add_argument("--m", type=int, default=10200)
ap.add_argument("--second-stream", action="store_true", help="a second stream keeps the chip busy with unrelated GEMMs and fills")
a_ = ap.parse_args()
m, d, hid, p, seed = a_.m, 256, 2048, 0.1, 7
g = torch.Generator().manual_seed(1)
bf = lambda x: x.to(torch.bfloat16).cuda()  # noqa: E731
a = bf(torch.randn(m, d, generator=g))
w1 = bf(torch.randn(hid, d, generator=g) / 16)
w2 = bf(torch.randn(d, hid, generator=g) / 45)
b1, b2 = torch.randn(hid, generator=g).cuda(), torch.randn(d, generator=g).cuda()
x = torch.randn(m, d, generator=g).cuda()
g1, be1 = (1 + 0.1 * torch.randn(d, generator=g)).cuda(), (0.1 * torch.randn(d, generator=g)).cuda()
g2, be2 = (1 + 0.1 * torch.randn(d, generator=g)).cuda(), (0.1 * torch.randn(d, generator=g)).cuda()
dy = bf(torch.randn(m, d, generator=g))
g0 = torch.randn(m, d, generator=g).cuda()
x2 = torch.randn(m, d, generator=g).cuda() * 1.5
pk = ops.ffn_pack_weights(w1, w2)
pt = ops.ffn_pack_weights(w2.t().contiguous(), w1.t().contiguous())
parts0 = None


def fwd():
    return K.ffn_train(a, pk, hid, b1, p, seed, 3, b2, x, 0.5, p, 4, ln1=(g1, be1), ln2=(g2, be2), tape_derivative=True)


ref = [t.clone() for t in fwd()]
gk = ref[0]


def bwd():
    gg = g0.clone()
    parts = torch.zeros(K.ffn_train_parts(m) * 512, device="cuda")
    parts2 = torch.zeros(K.ffn_train_parts(m) * 512, device="cuda")
    # (with the chained second LayerNorm backward: the form eleven of the step's twelve macaron launches take)
    du, dn = K.ffn_train_bwd(dy, pt, hid, gk, x, g1, gg, parts, chain=(x2, g2, parts2, (0.5, p, seed, 9, None)))
    return du, dn, gg, parts, parts2


refb = [t.clone() for t in bwd()]
bad = 0
# other work in between, so that the launches do not always meet the same cache state
junk = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda")
side = torch.cuda.Stream() if a_.second_stream else None
sa = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
for it in range(a_.iters):
    if it % 7 == 0:
        junk.fill_(float(it))
    if side is not None:
        with torch.cuda.stream(side):  # (no dependency on the main stream: runs beside the launches under test)
            sb = sa @ sa
            junk[: 1 << 20].add_(1.0)
    out = fwd()
    for k, (o, r) in enumerate(zip(out, ref)):
        if not torch.equal(o, r):
            bad += 1
            print("forward output %d differs at iteration %d: %d elements" % (k, it, int((o != r).sum())), flush=True)
    outb = bwd()
    for k, (o, r) in enumerate(zip(outb, refb)):
        if not torch.equal(o, r):
            bad += 1
            print("backward output %d differs at iteration %d: %d elements" % (k, it, int((o != r).sum())), flush=True)
    if bad > 20:
        break
torch.cuda.synchronize()
print("%d iterations of forward + backward, %d mismatching outputs" % (a_.iters, bad))
sys.exit(1 if bad else 0)
