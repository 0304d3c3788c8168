"""rocprofv3 --kernel-trace target: 20 launches of ffn_packed in each form, in a fixed order (mode 0, LN bf16, LN f32, LN2 bf16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindaudio_amd import ops
m = 64 * 249
a = torch.randn(m, 256, device="cuda").bfloat16(); w1 = (torch.randn(2048, 256, device="cuda") / 16).bfloat16(); b1 = torch.randn(2048, device="cuda")
w2 = (torch.randn(256, 2048, device="cuda") / 45).bfloat16(); b2 = torch.randn(256, device="cuda"); x = torch.randn(m, 256, device="cuda")
pk = ops.ffn_pack_weights(w1, w2); g = torch.ones(256, device="cuda"); be = torch.zeros(256, device="cuda")
ob = torch.empty(m, 256, device="cuda", dtype=torch.bfloat16)
for rep in range(2):
    for _ in range(20): ops.ffn_packed(a, pk, b1, b2, x)
    for _ in range(20): ops.ffn_packed(a, pk, b1, b2, x, g, be)
    for _ in range(20): ops.ffn_packed(a, pk, b1, b2, x, g, be, out_dtype=torch.float32)
    for _ in range(20): ops.ffn_packed(a, pk, b1, b2, x, g, be, g, be)
torch.cuda.synchronize()
