"""Same-process A/B of the headline step (fbank -> Conformer-small eval forward, 64 x 10 s) with the subsampling front end as one
launch (subsample_fused.hip) against the two kernels it replaces (conv1 -> conv2 over utterance groups): interleaved rounds."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import mindaudio_amd as ma
from mindaudio_amd.models import ConformerEncoder

dev = torch.device("cuda", 0)
encs = {}
for name, fused in (("fused", True), ("two-kernel", False)):
    torch.manual_seed(777)
    e = ConformerEncoder(80, 256, 4, 2048, 12).eval().to(dev)
    e.subsample_fused = fused
    encs[name] = e.prepare()
x = torch.from_numpy(bench.synth_batch(1234)).to(dev)
masks = torch.ones(bench.BATCH, 1, 249, device=dev)


def step(enc):
    feats = ma.fbank(x, **bench.FBANK_KW)
    return enc(feats.transpose(1, 2)[:, :bench.FRAMES], masks)[0]


outs = {k: step(e) for k, e in encs.items()}
d = (outs["fused"] - outs["two-kernel"]).abs().max().item()
print("max |fused - two-kernel| over the encoder output: %.3e (output scale %.2f)" % (d, outs["fused"].abs().max().item()))
res = {k: [] for k in encs}
for rnd in range(int(os.environ.get("ROUNDS", 5))):
    for k, e in encs.items():
        for _ in range(5):
            step(e)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            step(e)
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 50 * 1e3)
for k, v in res.items():
    v = sorted(v)
    print("%-11s ms per step: median %.4f  min %.4f  max %.4f  -> %.0f utt/s" % (k, v[len(v) // 2], v[0], v[-1], 64e3 / v[len(v) // 2]))
