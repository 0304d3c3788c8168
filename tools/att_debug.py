import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from mindaudio_amd import ops
def _rand(*sh, seed, scale=1.0):
    return t.randn(*sh, generator=t.Generator().manual_seed(seed)) * scale
b, tt = 3, 249
h, dk = 4, 64
qkv = _rand(b * tt, 768, seed=40).bfloat16()
pos = _rand(tt, 256, seed=41).bfloat16()
u = _rand(h, dk, seed=42, scale=0.2)
v = _rand(h, dk, seed=43, scale=0.2)
lens = [tt, max(1, tt - 37), max(1, tt // 2)][:b]
mask = t.zeros(b, tt)
for i, n in enumerate(lens):
    mask[i, :n] = 1.0
q = qkv[:, :256].double().view(b, tt, h, dk)
k = qkv[:, 256:512].double().view(b, tt, h, dk).transpose(1, 2)
vv = qkv[:, 512:].double().view(b, tt, h, dk).transpose(1, 2)
p = pos.double().view(1, tt, h, dk).transpose(1, 2)
qu = (q.float() + u).bfloat16().double().transpose(1, 2)
qv = (q.float() + v).bfloat16().double().transpose(1, 2)
scores = (qu @ k.transpose(-1, -2) + qv @ p.transpose(-1, -2)) / 8.0
scores = scores + (mask[:, None, None, :] == 0).double() * (-10000.0)
ref = (t.softmax(scores, -1) @ vv).transpose(1, 2).reshape(b * tt, 256)
got = ops.relpos_attention(qkv.cuda(), pos.cuda(), u.cuda(), v.cuda(), mask.cuda(), b, tt).double().cpu()
err = (got - ref).abs().view(b, tt, h, dk)
for bi in range(b):
    for hi in range(h):
        e = err[bi, :, hi].max(-1).values
        bad = (e > 0.05).nonzero().flatten().tolist()
        print(bi, hi, "max %.3g" % float(e.max()), "bad rows:", bad[:12], len(bad))
print("nan:", int(t.isnan(got).sum()))
# which output columns (d) are wrong in the bad rows
bi, hi = 0, 0
e = err[bi, :, hi]
print("row 0 errs by d:", [round(float(x), 2) for x in e[0]][:16], "...")
print("row 4 errs by d:", [round(float(x), 3) for x in e[4]][:8])
print("got row0[:8]", [round(float(x), 3) for x in got.view(b, tt, h, dk)[0, 0, 0, :8]], "ref", [round(float(x), 3) for x in ref.view(b, tt, h, dk)[0, 0, 0, :8]])
