#!/usr/bin/env python3
"""cfg 5 of SURVEY §8(d): EcapaTDNN forward on (256, 300, 80) synthetic features, C = 512 (class default) and 1024
(the example's size), eval-mode BatchNorm.  Prints one JSON line per configuration."""
import json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd.models import EcapaTDNN

flops = {512: 2.88e9, 1024: 10.78e9}
only = [int(a) for a in sys.argv[1:]] or [512, 1024]
for c in only:
    torch.manual_seed(0)
    m = EcapaTDNN(80, channels=(c, c, c, c, 3 * c)).eval().cuda().prepare()
    x = torch.randn(256, 300, 80, device="cuda")
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        out = m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"metric": "utterances/s, EcapaTDNN forward (256 x 3 s)", "channels": c, "value": round(256 / dt, 1),
                      "ms_per_batch": round(dt * 1e3, 3), "tflops": round(256 * flops[c] / dt / 1e12, 1),
                      "dtype": "bf16", "data": "synthetic"}))
