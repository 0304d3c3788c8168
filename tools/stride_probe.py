"""Development probe of the FFN launches of the training step, alone and in sequence (hot vs. just-written activations).
python tools/stride_probe.py   (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd import _host, _lib
from mindaudio_amd.train import kernels as K


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    M = int(os.environ.get("M", 10200))
    pk_r = torch.zeros(int(lib.ma_pack_item_pieces(1, 256, 2048)) * 16, dtype=torch.uint8, device=dev)
    pk_k = torch.zeros(int(lib.ma_pack_item_pieces(0, 2048, 256)) * 16, dtype=torch.uint8, device=dev)
    x = torch.randn(M, 256, device=dev).to(torch.bfloat16)
    res = torch.randn(M, 256, device=dev)
    b1 = torch.zeros(2048, device=dev)
    b2 = torch.zeros(256, device=dev)
    ln = (torch.ones(256, device=dev), torch.zeros(256, device=dev))
    h = torch.randn(M, 2048, device=dev).to(torch.bfloat16)
    flush = torch.empty(1 << 28, dtype=torch.uint8, device=dev)  # 256 MiB: evicts MALL
    with _host.pinned_stream():
        print("M = %d" % M)
        print("rows<4> K=2048 hot: %.1f us" % timeit(lambda: K.dense_plain(h, pk_r, 256, 2048)))
        print("rows<3> K=2048 hot, p=0.1, LN: %.1f us" % timeit(lambda: K.dense_join(h, pk_r, 2048, b2, res, 0.5, 0.1, 1, 2, ln1=ln)))
        print("rows<3> K=2048 hot, p=0, no LN: %.1f us" % timeit(lambda: K.dense_join(h, pk_r, 2048, b2, res, 0.5, 0.0, 1, 2)))
        print("k256<1>: %.1f us" % timeit(lambda: K.dense_act_drop(x, pk_k, 2048, b1, 0.1, 1, 2)))

        def pair():
            u, hh = K.dense_act_drop(x, pk_k, 2048, b1, 0.1, 1, 2)
            K.dense_join(hh, pk_r, 2048, b2, res, 0.5, 0.1, 1, 2, ln1=ln)

        print("k256<1> + rows<3> in sequence: %.1f us" % timeit(pair))
        t_f = timeit(lambda: flush.zero_())

        def cold():
            flush.zero_()
            K.dense_join(h, pk_r, 2048, b2, res, 0.5, 0.1, 1, 2, ln1=ln)

        print("rows<3> after a 256 MiB fill: %.1f us (fill alone %.1f)" % (timeit(cold) - t_f, t_f))


main()


def cold_strides():
    """cold reads (256 MiB fill in between) of the K = 2048 activation at several row strides"""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    M = 10200
    pk_r = torch.zeros(int(lib.ma_pack_item_pieces(1, 256, 2048)) * 16, dtype=torch.uint8, device=dev)
    flush = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    dy = torch.randn(M, 256, device=dev).to(torch.bfloat16)
    splits = int(lib.ma_gemm_tn_splits(2048, 256, M))
    part = torch.empty(splits * 2048 * 257 * 4, dtype=torch.uint8, device=dev)
    with _host.pinned_stream():
        t_f = timeit(lambda: flush.zero_())
        for pad in (0, 64, 128, 192):
            buf = torch.randn(M, 2048 + pad, device=dev).to(torch.bfloat16)
            a = buf[:, :2048]

            def cold():
                flush.zero_()
                K.dense_plain(a, pk_r, 256, 2048)

            def cold_tn():
                flush.zero_()
                K.gemm_tn_partial(a, dy, part, with_colsum=False)

            print("cold, row stride %d B: rows<4> %.1f us, gemm_tn h^T dy %.1f us" % (2 * (2048 + pad), timeit(cold) - t_f, timeit(cold_tn) - t_f))


cold_strides()


def rotate():
    """cold reads without a fill in between: the launch walks 16 distinct 40 MB activations (640 MB > the 256 MB Infinity Cache)"""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    M = 10200
    pk_r = torch.zeros(int(lib.ma_pack_item_pieces(1, 256, 2048)) * 16, dtype=torch.uint8, device=dev)
    hs = [torch.randn(M, 2048, device=dev).to(torch.bfloat16) for _ in range(16)]
    dy = torch.randn(M, 256, device=dev).to(torch.bfloat16)
    splits = int(lib.ma_gemm_tn_splits(2048, 256, M))
    part = torch.empty(splits * 2048 * 257 * 4, dtype=torch.uint8, device=dev)
    with _host.pinned_stream():
        def walk():
            for h in hs:
                K.dense_plain(h, pk_r, 256, 2048)

        def walk_tn():
            for h in hs:
                K.gemm_tn_partial(h, dy, part, with_colsum=False)

        def walk_sum():
            for h in hs:
                h.sum()

        print("16 distinct activations: rows<4> %.1f us each, gemm_tn %.1f us each, torch sum %.1f us each" %
              (timeit(walk, 10) / 16, timeit(walk_tn, 10) / 16, timeit(walk_sum, 10) / 16))


rotate()

