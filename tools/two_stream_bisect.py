"""Development: which launch of the encoder forward gives different bits when two forwards run concurrently on two streams?
Runs the two-stream comparison of tests/test_conformer_encoder_gpu.py with parts of the pipeline switched to their other forms."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd.models import ConformerEncoder


def run(tag, blocks=12, **flags):
    torch.manual_seed(3)
    enc = ConformerEncoder(80, 256, 4, 2048, blocks).eval().cuda()
    for k, v in flags.items():
        setattr(enc, k, v)
    enc.prepare()
    if flags.get("_general"):
        enc._prepared["fused"] = False
    b, frames = 32, 1000
    t2 = ((frames - 3) // 2 + 1 - 3) // 2 + 1
    xs = [torch.randn(b, frames, 80, device="cuda") for _ in range(2)]
    m = torch.ones(b, 1, t2, device="cuda")
    want = [enc(x, m)[0].clone() for x in xs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    cur = torch.cuda.current_stream()
    bad, worst = 0, 0.0
    for _ in range(10):
        outs = []
        for s_, x in zip(streams, xs):
            s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                outs.append(enc(x, m)[0])
        for s_ in streams:
            cur.wait_stream(s_)
        torch.cuda.synchronize()
        d = max(float((o - w).abs().max()) for o, w in zip(outs, want))
        bad += d != 0.0
        worst = max(worst, d)
    print("%-40s differing runs %d / 10, worst |diff| %.4g" % (tag, bad, worst), flush=True)


run("default")
run("two-kernel front end", subsample_fused=False)
run("general blocks (one launch per cell)", _general=True)
run("1 block", blocks=1)
run("1 block, two-kernel front end", blocks=1, subsample_fused=False)
