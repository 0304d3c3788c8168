#!/usr/bin/env python3
"""The resampler's power-of-two FFT alone (ma_fft_pow2_c32): ms per transform batch and the HBM rate its passes reach.
    python tools/fft_pass_bench.py [rows] [log2L ...]"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    from mindaudio_amd import _host, _lib

    lib = _lib.load()
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 35
    for log2l in [int(v) for v in sys.argv[2:]] or [18, 19]:
        L = 1 << log2l
        d = torch.randn(rows, L, 2, device="cuda")
        tmp = torch.empty_like(d)
        res = ctypes.c_void_p()
        call = lambda: _lib.check(lib.ma_fft_pow2_c32(_host.ptr(d), _host.ptr(tmp), rows, L, 0, ctypes.byref(res), _host.current_stream_ptr()), "fft")  # noqa: E731
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        passes = (log2l + 6) // 7
        print(json.dumps({"rows": rows, "log2L": log2l, "ms_per_fft": round(ms, 4), "passes": passes,
                          "TBps_per_pass": round(passes * 2 * rows * L * 8 / (ms * 1e-3) / 1e12, 2)}))


if __name__ == "__main__":
    main()
