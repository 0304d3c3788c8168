import torch, math, sys, os
sys.path.insert(0, "/root/repo")
from mindaudio_amd import ops
t = torch
def _rand(*shape, seed=0, scale=1.0):
    g = t.Generator().manual_seed(seed); return t.randn(*shape, generator=g) * scale
m, hidden, d = 64, 256, 256
a = _rand(m, d, seed=90).bfloat16(); w1 = _rand(hidden, d, seed=91, scale=1/16).bfloat16(); b1 = _rand(hidden, seed=92, scale=0.3)
w2 = t.eye(256).bfloat16()
z = a.double() @ w1.double().T + b1.double()
lin = os.environ.get("MA_FFNPK_ABLATE") == "1"
h = z.bfloat16().double() if lin else (z * t.sigmoid(z)).bfloat16().double()
packed = ops.ffn_pack_weights(w1.cuda(), w2.cuda())
xg = t.zeros(m, d).cuda()
ops.ffn_packed(a.cuda(), packed, b1.cuda(), t.zeros(d).cuda(), xg, alpha=1.0)
got = xg.double().cpu()
err = (got - h).abs()
print("max err", float(err.max()))
bad = err > 0.02
print("bad count", int(bad.sum()), "of", bad.numel())
print("bad per hidden block of 32:", [int(bad[:, i*32:(i+1)*32].sum()) for i in range(8)])
print("bad per row tile of 16:", [int(bad[i*16:(i+1)*16].sum()) for i in range(4)])
hb = 0
for blk in range(8):
    sub = bad[:, blk*32:(blk+1)*32]
    print(blk, "bad by unit-in-block:", sub.sum(0).tolist())
