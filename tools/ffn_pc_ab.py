"""The evaluation forward's FFN launch (pair + linear_q/k/v: 2 x [w1 -> Swish -> w2 + residual] + 4 LayerNorms + qkv, M = 64 x 249)
timed on the kernel the library selects.  Run twice for the A/B:  MINDAUDIO_AMD_FFN=pc python tools/ffn_pc_ab.py  (ffn_pc.hip)
and without the variable (ffn_packed.hip, the default)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd import ops

m, d, hid = 64 * 249, 256, 2048
r = lambda *sh: torch.randn(*sh, device="cuda")
packs = []
for i in range(12):  # 12 different weight sets, as in the encoder: the weights come from L2-cold memory
    packs.append((ops.ffn_pack_weights((r(hid, d) / 16).bfloat16(), (r(d, hid) / 45).bfloat16()), r(hid) * 0.3, r(d) * 0.3))
lns = [(1 + 0.1 * r(d), 0.1 * r(d)) for _ in range(4)]
pq, bq = ops.ffn_qkv_pack((r(768, d) / 16).bfloat16()), r(768) * 0.3
x = r(m, d)
k = [0]


def pair():
    a, b = packs[k[0] % 12], packs[(k[0] + 1) % 12]
    k[0] += 2
    return ops.ffn_packed_pair(a[0], a[1], a[2], b[0], b[1], b[2], x, lns[0], lns[1], lns[2], lns[3], qkv=(pq, bq))


def single():
    a = packs[k[0] % 12]
    k[0] += 1
    return ops.ffn_packed(None, a[0], a[1], a[2], x, lns[1][0], lns[1][1], lns[2][0], lns[2][1], out_dtype=torch.float32, ln_in=lns[0])


def t(fn, reps=60):
    for _ in range(6):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fl_pair = 2.0 * m * (2 * 2 * hid * d + 768 * d)
for rnd in range(3):
    up, us = t(pair), t(single)
    print("%s: pair + qkv %.1f us (%.0f TFLOP/s)   single + 2 LN %.1f us" % (os.environ.get("MINDAUDIO_AMD_FFN", "packed"), up, fl_pair / up / 1e6, us))
