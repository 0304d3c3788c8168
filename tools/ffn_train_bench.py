"""Development: the one-launch feed-forward training module (ma_ffn_train_bf16) against the two launches it replaces, M = 10 200.

    python tools/ffn_train_bench.py [--m 10200] [--iters 50]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindaudio_amd import _lib, ops  # noqa: E402
from mindaudio_amd.train import kernels as K  # noqa: E402
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=10200)
    ap.add_argument("--iters", type=int, default=50)
    a_ = ap.parse_args()
    lib = _lib.load()
    m, d, hid, p, seed = a_.m, 256, 2048, 0.1, 7
    g = torch.Generator().manual_seed(0)
    bf = lambda x: x.to(torch.bfloat16)  # noqa: E731
    a = bf(torch.randn(m, d, generator=g)).cuda()
    w1 = bf(torch.randn(hid, d, generator=g) / 16).cuda()
    w2 = bf(torch.randn(d, hid, generator=g) / 45).cuda()
    b1, b2 = torch.randn(hid, generator=g).cuda(), torch.randn(d, generator=g).cuda()
    x = torch.randn(m, d, generator=g).cuda()
    g1, be1 = torch.ones(d).cuda(), torch.zeros(d).cuda()
    st = torch.cuda.current_stream().cuda_stream

    def pack(w, kind):
        n, k = w.shape
        pieces = int(lib.ma_pack_item_pieces(kind, n, k))
        out = torch.empty(pieces * 16, dtype=torch.uint8, device="cuda")
        items = (_lib.PackItem * 1)(_lib.PackItem(w.data_ptr(), out.data_ptr(), w.stride(0), n, k, kind, 0))
        d_items = torch.from_numpy(np.frombuffer(bytes(items), dtype=np.uint8).copy()).cuda()
        d_map = torch.zeros((pieces + 255) // 256, dtype=torch.int32, device="cuda")
        _lib.check(lib.ma_pack_batch_bf16(d_items.data_ptr(), d_map.data_ptr(), d_map.numel(), st), "pack")
        return out

    pk = ops.ffn_pack_weights(w1, w2)
    pk1, pk2 = pack(w1, 0), pack(w2, 1)

    def fused():
        return K.ffn_train(a, pk, hid, b1, p, seed, 3, b2, x, 0.5, p, 4, ln1=(g1, be1))

    def two():
        u, h = K.dense_act_drop(a, pk1, hid, b1, p, seed, 3)
        return K.dense_join(h, pk2, hid, b2, x, 0.5, p, seed, 4, ln1=(g1, be1))

    dy = bf(torch.randn(m, d, generator=g)).cuda()
    g0 = torch.randn(m, d, generator=g).cuda()
    pt = ops.ffn_pack_weights(w2.t().contiguous(), w1.t().contiguous())
    pk2t, pk1t = pack(w2.t().contiguous(), 0), pack(w1.t().contiguous(), 1)
    parts = torch.zeros(max(K.ffn_train_parts(m), K.rows_train_parts(m)) * 512, device="cuda")
    gk = K.ffn_train(a, pk, hid, b1, p, seed, 3, b2, x, 0.5, p, 4, ln1=(g1, be1), tape_derivative=True)[0]
    u = K.dense_act_drop(a, pk1, hid, b1, p, seed, 3)[0]
    nxt = (0.5, p, seed, 9, None)

    def fused_bwd():
        return K.ffn_train_bwd(dy, pt, hid, gk, x, g1, g0, parts, nxt=nxt)

    def two_bwd():
        du = K.dense_act_drop_bwd(dy, pk2t, hid, u, p, seed, 3)
        return K.dense_lnbwd(du, pk1t, hid, x, g1, g0, parts, nxt=nxt)

    for name, fn in (("two launches", two), ("one launch", fused), ("two launches", two), ("one launch", fused),
                     ("bwd two", two_bwd), ("bwd one", fused_bwd), ("bwd two", two_bwd), ("bwd one", fused_bwd)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a_.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / a_.iters
        flops = 2 * 2 * m * d * hid
        print("%-13s %7.1f us   %6.1f TFLOP/s" % (name, us, flops / us / 1e6), flush=True)


main()
