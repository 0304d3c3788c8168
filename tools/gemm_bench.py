#!/usr/bin/env python3
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd import ops, _lib
def timeit(fn, reps=30, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for (m, n, k, nm) in [(7968, 2048, 256, "ffn w1"), (7968, 256, 2048, "ffn w2"), (7968, 768, 256, "qkv"), (7968, 256, 256, "out proj"),
                      (7968, 512, 256, "pw conv1"), (7968, 256, 4864, "subsample out"), (15936, 2048, 256, "ffn w1 B64"), (76800, 1024, 1024, "ecapa 1x1"), (76800, 3072, 3072, "ecapa mfa"), (10240, 2048, 256, "train w1"), (10240, 256, 2048, "train w2"), (4096, 4096, 4096, "4096^3"), (8192, 8192, 8192, "8192^3")]:
    a = torch.randn(m, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).bfloat16()
    bias = torch.randn(n, device="cuda")
    s = timeit(lambda: ops.gemm(a, w, bias=bias, act=_lib.ACT_RELU if m > 20000 else _lib.ACT_SWISH))
    sref = timeit(lambda: torch.nn.functional.linear(a, w))
    print("%-16s M=%6d N=%5d K=%5d  %8.1f us  %7.1f TF/s   (hipBLASLt via torch: %8.1f us %7.1f TF/s)" % (nm, m, n, k, s * 1e6, 2.0 * m * n * k / s / 1e12, sref * 1e6, 2.0 * m * n * k / sref / 1e12))
x = torch.randn(32, 499, 39, 256, device="cuda").bfloat16(); w = (torch.randn(256, 3, 3, 256, device="cuda") / 48).bfloat16(); bias = torch.randn(256, device="cuda")
s = timeit(lambda: ops.conv2d_3x3s2_nhwc(x, w, bias=bias))
fl = 2.0 * 32 * 249 * 19 * 256 * 2304
print("conv2 implicit gemm B=32: %.1f us %.1f TF/s" % (s * 1e6, fl / s / 1e12))
m, n, k = 15936, 256, 4864
a = torch.randn(m, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).bfloat16(); bias = torch.randn(n, device="cuda")
pk = ops.gemm_rows_pack(w)
s = timeit(lambda: ops.gemm_rows_packed(a, pk, bias, alpha=16.0))
s0 = timeit(lambda: ops.gemm(a, w, bias=bias, alpha=16.0, out_dtype=torch.float32))
print("embed layer M=%d K=%d: rows_packed %.1f us (%.0f TF/s)   general kernel %.1f us" % (m, k, s * 1e6, 2.0 * m * n * k / s / 1e12, s0 * 1e6))
