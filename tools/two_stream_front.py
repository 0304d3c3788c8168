"""Development: the two-kernel front end (subsample_conv1 -> conv2d_3x3s2_packed) on two concurrent streams vs alone, per kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd import ops
from mindaudio_amd.models import ConformerEncoder

torch.manual_seed(3)
enc = ConformerEncoder(80, 256, 4, 2048, 1).eval().cuda()
enc.subsample_fused = False
enc.prepare()
P = enc._prepared
b, frames = 32, 1000
t1, f1 = (frames - 3) // 2 + 1, 39
t2, f2 = (t1 - 3) // 2 + 1, 19
xs = [torch.randn(b, frames, 80, device="cuda") for _ in range(2)]


def conv1(x):
    out = torch.empty((b, t1, f1, 256), dtype=torch.bfloat16, device="cuda")
    ops.subsample_conv1(x, P["conv1_w"], P["conv1_b"], enc.cmvn_mean, enc.cmvn_istd, out=out)
    return out


def conv2(a):
    out = torch.empty((b, t2, f2, 256), dtype=torch.bfloat16, device="cuda")
    ops.conv2d_3x3s2_packed(a, P["conv2_pk"], P["conv2_b"], relu=True, out=out)
    return out


a1 = [conv1(x) for x in xs]
a2 = [conv2(a) for a in a1]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
cur = torch.cuda.current_stream()
for name, fn, ins, want in (("conv1 || conv1", conv1, xs, a1), ("conv2 || conv2", conv2, a1, a2)):
    bad = 0
    for _ in range(20):
        outs = []
        for s_, i_ in zip(streams, ins):
            s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                outs.append(fn(i_))
        for s_ in streams:
            cur.wait_stream(s_)
        torch.cuda.synchronize()
        bad += not all(torch.equal(o, w) for o, w in zip(outs, want))
    print("%s: %d / 20 runs differ" % (name, bad), flush=True)
# conv1 on one stream beside conv2 on the other
bad = 0
for _ in range(20):
    for s_ in streams:
        s_.wait_stream(cur)
    with torch.cuda.stream(streams[0]):
        o1 = conv1(xs[0])
    with torch.cuda.stream(streams[1]):
        o2 = conv2(a1[1])
    for s_ in streams:
        cur.wait_stream(s_)
    torch.cuda.synchronize()
    bad += not (torch.equal(o1, a1[0]) and torch.equal(o2, a2[1]))
print("conv1 || conv2: %d / 20 runs differ (conv1 %s, conv2 %s)" % (bad, torch.equal(o1, a1[0]), torch.equal(o2, a2[1])))
# the model's own grouping: conv1 -> conv2 -> conv1 -> conv2 through ONE act1 buffer per forward
want = [enc._subsample(x, P).clone() for x in xs]
torch.cuda.synchronize()
bad = 0
for _ in range(20):
    outs = []
    for s_, x in zip(streams, xs):
        s_.wait_stream(cur)
        with torch.cuda.stream(s_):
            outs.append(enc._subsample(x, P))
    for s_ in streams:
        cur.wait_stream(s_)
    torch.cuda.synchronize()
    bad += not all(torch.equal(o, w) for o, w in zip(outs, want))
print("_subsample || _subsample (groups through one act1 buffer): %d / 20 runs differ" % bad)
