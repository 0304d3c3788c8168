"""Development: ma_asp_fused_bf16 alone at the cfg-5 sizes (B = 256, T = 300, att = 128, C = 1536 / 3072): us per launch.
    python tools/asp_bench.py [C ...]       (MINDAUDIO_AMD_LIB=<variant.so> for tools/ffn_variants.sh builds)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd import _host, _lib

lib = _lib.load()
B, T, H = 256, 300, 4
tp = T + 2 * H
for C in [int(a) for a in sys.argv[1:]] or [1536, 3072]:
    torch.manual_seed(0)
    a1 = torch.tanh(torch.randn(B * tp, 128, device="cuda")).to(torch.bfloat16)
    W = (torch.randn(C, 128, device="cuda") * 0.1).to(torch.bfloat16)
    x = torch.randn(B * tp, C, device="cuda").to(torch.bfloat16)
    sc = torch.ones(2 * C, device="cuda")
    sh = torch.zeros(2 * C, device="cuda")
    out = torch.empty(B, 2 * C, device="cuda", dtype=torch.bfloat16)
    s = _host.current_stream_ptr()

    def run():
        _lib.check(lib.ma_asp_fused_bf16(a1.data_ptr(), 128, W.data_ptr(), x.data_ptr(), C, B, T, H, C, 128, 1e-12, sc.data_ptr(),
                                         sh.data_ptr(), out.data_ptr(), s), "asp_fused")

    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print("C=%d: %.1f us per launch, x stream %.2f TB/s" % (C, us, B * T * C * 2 / us / 1e6))
