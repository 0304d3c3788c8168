import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd import ops, _lib
B = int(os.environ.get("B", 64)); m = B * 249
a = torch.randn(m, 256, device="cuda").bfloat16(); w1 = (torch.randn(2048, 256, device="cuda") / 16).bfloat16(); b1 = torch.randn(2048, device="cuda")
w2 = (torch.randn(256, 2048, device="cuda") / 45).bfloat16(); b2 = torch.randn(256, device="cuda"); x = torch.randn(m, 256, device="cuda")
def t(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
fl = 2.0 * m * 2048 * 256 * 2
def unf():
    h = ops.gemm(a, w1, bias=b1, act=_lib.ACT_SWISH); ops.gemm(h, w2, bias=b2, residual=x, alpha=0.5, out_dtype=torch.float32, out=x)
us = t(unf); print("2-gemm ffn M=%d: %.1f us  %.0f TF/s" % (m, us, fl / us / 1e6))
pk = ops.ffn_pack_weights(w1, w2)
g = torch.ones(256, device="cuda"); be = torch.zeros(256, device="cuda")
us = t(lambda: ops.ffn_packed(a, pk, b1, b2, x)); print("packed ffn M=%d: %.1f us  %.0f TF/s" % (m, us, fl / us / 1e6))
us = t(lambda: ops.ffn_packed(a, pk, b1, b2, x, g, be)); print("packed+LN  M=%d: %.1f us  %.0f TF/s" % (m, us, fl / us / 1e6))
us = t(lambda: ops.ffn_packed(a[:64], pk, b1, b2, x[:64])); print("packed M=64 (launch floor + one workgroup): %.1f us" % us)
us = t(lambda: ops.ffn_packed(a, pk, b1, b2, x, g, be, out_dtype=torch.float32)); print("packed+LN f32 out: %.1f us" % us)
us = t(lambda: ops.ffn_packed(a, pk, b1, b2, x, g, be, g, be)); print("packed+LN2 bf16 out: %.1f us" % us)
