"""Development: the headline encoder forward (64 x 1000 x 80 features) as ONE launch chain against the same batch as two (or four)
independent part-batch chains on separate HIP streams (kernel tails / launch ramps of one chain under the other's kernels).
    python tools/two_stream_fwd.py [parts ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd.models import ConformerEncoder

B, FRAMES = 64, 1000
t2 = ((FRAMES - 3) // 2 + 1 - 3) // 2 + 1
torch.manual_seed(777)
enc = ConformerEncoder(80, 256, 4, 2048, 12).eval().cuda().prepare()
xs = torch.randn(B, FRAMES, 80, device="cuda")
masks = torch.ones(B, 1, t2, device="cuda")
ref = enc(xs, masks)[0]
torch.cuda.synchronize()
for parts in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    n = B // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    xp = [xs[i * n:(i + 1) * n].contiguous() for i in range(parts)]
    mp = [masks[i * n:(i + 1) * n].contiguous() for i in range(parts)]

    def step():
        outs = []
        cur = torch.cuda.current_stream()
        for s, x_, m_ in zip(streams, xp, mp):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(enc(x_, m_)[0])
        for s in streams:
            cur.wait_stream(s)
        return outs

    for _ in range(5):
        outs = step()
    torch.cuda.synchronize()
    got = torch.cat(outs)
    err = float((got - ref).abs().max())
    t0 = time.perf_counter()
    k = 50
    for _ in range(k):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    print("parts=%d: %.3f ms per 64 utterances (%.0f utt/s), max |diff| vs one chain %.3g" % (parts, dt * 1e3, B / dt, err))
