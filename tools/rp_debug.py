import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd import ops
m, n, k = int(sys.argv[1]), 256, int(sys.argv[2])
a = torch.randn(m, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).bfloat16(); bias = torch.randn(n, device="cuda")
pk = ops.gemm_rows_pack(w); torch.cuda.synchronize(); print("pack ok", flush=True)
out = ops.gemm_rows_packed(a, pk, bias, alpha=2.0); torch.cuda.synchronize(); print("gemm ok", flush=True)
ref = (a.double() @ w.double().T + bias.double()) * 2.0
print("max err", float((out.double() - ref).abs().max()), "scale", float(ref.abs().max()))
