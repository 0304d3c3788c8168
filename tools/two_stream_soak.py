"""Development: the default encoder forward on two concurrent streams, N iterations, every launch traced on a mismatch.
    python tools/two_stream_soak.py [iterations] [batch per stream]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd.models import ConformerEncoder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.manual_seed(3)
enc = ConformerEncoder(80, 256, 4, 2048, 12).eval().cuda().prepare()
frames = 1000
t2 = ((frames - 3) // 2 + 1 - 3) // 2 + 1
xs = [torch.randn(b, frames, 80, device="cuda") for _ in range(2)]
m = torch.ones(b, 1, t2, device="cuda")
want = [enc(x, m)[0].clone() for x in xs]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
cur = torch.cuda.current_stream()
bad, worst = 0, 0.0
for it in range(n):
    outs = []
    for s_, x in zip(streams, xs):
        s_.wait_stream(cur)
        with torch.cuda.stream(s_):
            outs.append(enc(x, m)[0])
    for s_ in streams:
        cur.wait_stream(s_)
    torch.cuda.synchronize()
    d = max(float((o - w).abs().max()) for o, w in zip(outs, want))
    if d != 0.0:
        bad += 1
        worst = max(worst, d)
        print("iteration %d: max |diff| %.4g" % (it, d), flush=True)
print("two_stream_soak: %d iterations of 2 x %d utterances, %d differing, worst |diff| %.4g" % (n, b, bad, worst))
