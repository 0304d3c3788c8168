#!/usr/bin/env python3
"""Small fixed workloads for rocprofv3 passes: `fbank` (cfg-2 fbank kernel), `gemm` (FFN w_1 shape), `step` (bench step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mindaudio_amd as ma
from mindaudio_amd import _host, _lib, ops
from mindaudio_amd.models import ConformerEncoder
what = sys.argv[1] if len(sys.argv) > 1 else "step"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, N = 64, 160000
x = torch.from_numpy((0.1 * np.random.RandomState(1234).randn(B, N)).astype(np.float32)).cuda()
if what == "fbank":
    for _ in range(n): ma.fbank(x, n_mels=80, n_fft=512, hop_length=160)
elif what == "gemm8k":  # the 256 x 256 8-phase GEMM kernel on a large cube
    m = k2 = 8192
    a = torch.randn(m, k2, device="cuda").bfloat16(); w = (torch.randn(m, k2, device="cuda") / 90).bfloat16()
    bias = torch.randn(m, device="cuda"); out = torch.empty(m, m, device="cuda", dtype=torch.bfloat16)
    for _ in range(n): ops.gemm(a, w, bias=bias, act=_lib.ACT_RELU, out=out)
elif what == "gemm":
    m, nn, k = B * 249, 2048, 256
    a = torch.randn(m, k, device="cuda").bfloat16(); w = (torch.randn(nn, k, device="cuda") / 16).bfloat16(); bias = torch.randn(nn, device="cuda")
    out = torch.empty(m, nn, dtype=torch.bfloat16, device="cuda")
    for _ in range(n): ops.gemm(a, w, bias=bias, act=_lib.ACT_SWISH, out=out)
elif what == "ffnpk":
    m = B * 249
    a = torch.randn(m, 256, device="cuda").bfloat16(); w1 = (torch.randn(2048, 256, device="cuda") / 16).bfloat16(); b1 = torch.randn(2048, device="cuda")
    w2 = (torch.randn(256, 2048, device="cuda") / 45).bfloat16(); b2 = torch.randn(256, device="cuda"); xx = torch.randn(m, 256, device="cuda")
    pk = ops.ffn_pack_weights(w1, w2)
    for _ in range(n): ops.ffn_packed(a, pk, b1, b2, xx)
elif what == "ffnpkln":  # the form the encoder launches: FFN + residual + LayerNorm (bf16 out)
    m = B * 249
    a = torch.randn(m, 256, device="cuda").bfloat16(); w1 = (torch.randn(2048, 256, device="cuda") / 16).bfloat16(); b1 = torch.randn(2048, device="cuda")
    w2 = (torch.randn(256, 2048, device="cuda") / 45).bfloat16(); b2 = torch.randn(256, device="cuda"); xx = torch.randn(m, 256, device="cuda")
    pk = ops.ffn_pack_weights(w1, w2); g = torch.ones(256, device="cuda"); be = torch.zeros(256, device="cuda")
    for _ in range(n): ops.ffn_packed(a, pk, b1, b2, xx, g, be)
elif what == "ffnpair":  # the form the encoder launches 11 x per forward: FFN + FFN' + LayerNorms + linear_q/k/v
    m = B * 249
    r = lambda *sh: torch.randn(*sh, device="cuda")
    pa = ops.ffn_pack_weights((r(2048, 256) / 16).bfloat16(), (r(256, 2048) / 45).bfloat16())
    pb = ops.ffn_pack_weights((r(2048, 256) / 16).bfloat16(), (r(256, 2048) / 45).bfloat16())
    pq = ops.ffn_qkv_pack((r(768, 256) / 16).bfloat16()); bq = r(768); b1 = r(2048); b2 = r(256); xx = r(m, 256)
    ln = (torch.ones(256, device="cuda"), torch.zeros(256, device="cuda"))
    for _ in range(n): ops.ffn_packed_pair(pa, b1, b2, pb, b1, b2, xx, ln, ln, ln, ln, qkv=(pq, bq))
elif what == "attn":
    T = 249
    qkv = (torch.randn(B * T, 768, device="cuda") * 0.5).bfloat16(); pos = (torch.randn(T, 256, device="cuda") * 0.5).bfloat16()
    u = torch.randn(4, 64, device="cuda") * 0.1; v = torch.randn(4, 64, device="cuda") * 0.1; mask = torch.ones(B, T, device="cuda")
    for _ in range(n): ops.relpos_attention(qkv, pos, u, v, mask, B, T)
else:
    torch.manual_seed(777)
    enc = ConformerEncoder(80, 256, 4, 2048, 12).eval().cuda().prepare()
    masks = torch.ones(B, 1, 249, device="cuda")
    for _ in range(n):
        feats = ma.fbank(x, n_mels=80, n_fft=512, hop_length=160)
        enc(feats.transpose(1, 2)[:, :1000].contiguous(), masks)
torch.cuda.synchronize()
