"""Development: time of ma_gemm_bf16 at ECAPA's C = 512 shape (M = 80 896, N = K = 512, ReLU + BatchNorm + row-scale epilogue), rotating
over 8 activation buffers (the layer's input is the previous launch's output: cold).  python tools/gemm512_probe.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd import _host, _lib


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    m, n, k = int(os.environ.get("M", 80896)), 512, 512
    As = [torch.randn(m, k, device=dev).to(torch.bfloat16) for _ in range(8)]
    w = (torch.randn(n, k, device=dev) / 22).to(torch.bfloat16)
    bias, cs, ct = torch.randn(n, device=dev), torch.ones(n, device=dev), torch.zeros(n, device=dev)
    rs = torch.ones(m, device=dev)
    outs = [torch.empty(m, n, dtype=torch.bfloat16, device=dev) for _ in range(8)]
    e = _lib.GemmEpilogue()
    e.bias, e.row_scale, e.col_scale, e.col_shift = bias.data_ptr(), rs.data_ptr(), cs.data_ptr(), ct.data_ptr()
    e.alpha, e.act, e.act2, e.out_bf16 = 1.0, _lib.ACT_RELU, 0, 1

    def run():
        for a, o in zip(As, outs):
            lib.ma_gemm_bf16(_host.ptr(a), a.stride(0), _host.ptr(w), w.stride(0), _host.ptr(o), o.stride(0), m, n, k, ctypes.byref(e),
                             _host.current_stream_ptr())

    with _host.pinned_stream():
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 80 * 1e3
    print("M = %d: %.1f us per launch, %.0f TFLOP/s" % (m, us, 2.0 * m * n * k / us / 1e6))


main()
