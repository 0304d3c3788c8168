import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd import ops
m = 64 * 249
def t(fn, reps=50):
    # the launches are replayed from a captured graph: a Python launch loop costs ~9 us per call, more than some of these kernels
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(4): g.replay()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / (4 * reps) * 1e3
a = torch.randn(m, 256, device="cuda").bfloat16(); x = torch.randn(m, 256, device="cuda")
for n, kind in ((768, "qkv bf16"), (512, "pw1 bf16"), (256, "out f32+res")):
    w = (torch.randn(n, 256, device="cuda") / 16).bfloat16(); b = torch.randn(n, device="cuda"); pk = ops.gemm_k256_pack(w)
    if n == 256:
        u0 = t(lambda: ops.gemm(a, w, bias=b, residual=x, out_dtype=torch.float32, out=x)); u1 = t(lambda: ops.gemm_packed(a, pk, bias=b, residual=x, out_dtype=torch.float32, out=x))
    else:
        o = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        u0 = t(lambda: ops.gemm(a, w, bias=b, out=o)); u1 = t(lambda: ops.gemm_packed(a, pk, bias=b, out=o))
    print("N=%d %s: general %.1f us, packed %.1f us" % (n, kind, u0, u1))
