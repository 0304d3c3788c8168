#!/usr/bin/env python3
"""The subsampling embed Dense (M = 64 x 249 rows, K = 4864 -> 256) on the launches that can run it: the row-owner packed kernel (the
evaluation forward's choice), the general GEMM, the split-K product.  HIP-event time of 50 back-to-back launches each.
    python tools/embed_gemm_bench.py [--rows 15936]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=15936)
    ap.add_argument("--k", type=int, default=4864)
    a = ap.parse_args()
    from mindaudio_amd import _host, ops
    from mindaudio_amd.train import kernels as K

    m, k = a.rows, a.k
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(m, k, device="cuda", generator=g).bfloat16()
    w = (torch.randn(256, k, device="cuda", generator=g) / 70).bfloat16()
    b = torch.randn(256, device="cuda", generator=g)
    pk = ops.gemm_rows_pack(w)
    out = torch.empty(m, 256, device="cuda")
    with _host.pinned_stream():
        print("rows_packed  %.1f us" % timed(lambda: ops.gemm_rows_packed(x, pk, b, alpha=16.0, out=out)))
        print("general gemm %.1f us" % timed(lambda: ops.gemm(x, w, bias=b, alpha=16.0, out_dtype=torch.float32, out=out)))
        print("split-K      %.1f us" % timed(lambda: K.gemm_splitk(x, w, out, accumulate=False)))
    flops = 2.0 * m * k * 256
    print("(%.1f GFLOP, %.0f MB of activations)" % (flops / 1e9, m * k * 2 / 1e6))


if __name__ == "__main__":
    main()
