"""ffn_packed pair (+ qkv) launch time against the hidden size: slope = main-loop cost per 256 hidden units, intercept = the
fixed cost of the launch (prologues, two reductions / epilogues, stage hand-over, qkv tail, drain)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd import ops

m = 64 * 249
dev = "cuda"
r = lambda *sh: torch.randn(*sh, device=dev)
ln = (torch.ones(256, device=dev), torch.zeros(256, device=dev))
x = r(m, 256)
pq, bq = ops.ffn_qkv_pack((r(768, 256) / 16).bfloat16()), r(768)


def t(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


for hid in ((int(os.environ["HID"]),) if os.environ.get("HID") else (256, 512, 1024, 2048)):
    pa = ops.ffn_pack_weights((r(hid, 256) / 16).bfloat16(), (r(256, hid) / 45).bfloat16())
    pb = ops.ffn_pack_weights((r(hid, 256) / 16).bfloat16(), (r(256, hid) / 45).bfloat16())
    b1, b2 = r(hid), r(256)
    a = r(m, 256).bfloat16()
    single = t(lambda: ops.ffn_packed(a, pa, b1, b2, x, ln[0], ln[1]))
    pair = t(lambda: ops.ffn_packed_pair(pa, b1, b2, pb, b1, b2, x, ln, ln, ln, ln))
    pairq = t(lambda: ops.ffn_packed_pair(pa, b1, b2, pb, b1, b2, x, ln, ln, ln, ln, qkv=(pq, bq)))
    print("hidden %4d: single+LN %.1f us   pair %.1f us   pair+qkv %.1f us" % (hid, single, pair, pairq))
