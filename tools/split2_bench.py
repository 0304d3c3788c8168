"""Experiment: the 12 Conformer blocks on the full 64-utterance batch vs on two 32-utterance halves issued to two streams
(kernels of the two halves may overlap: one half's store drain / latency phases under the other half's MFMA phases)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd import ops
from mindaudio_amd.models import ConformerEncoder

torch.manual_seed(777)
dev = torch.device("cuda")
enc = ConformerEncoder(80, 256, 4, 2048, 12).eval().to(dev).prepare()
P = enc._prepared
B, t2 = 64, 249
pos_all = enc._pos_projection(t2)
x0 = torch.randn(B * t2, 256, device=dev)
mask = torch.ones(B, t2, device=dev)


def blocks(x, mask2d, b):
    mask_rows = mask2d.reshape(-1)
    L = P["layers"]
    l0 = enc.encoders[0]
    a = ops.layernorm(x, l0.norm_ff_macaron.gamma, l0.norm_ff_macaron.beta)
    qkv = ops.ffn_packed_qkv(a, L[0]["ffm_pk"], L[0]["ffm_b1"], L[0]["ffm_b2"], x, l0.norm_mha.gamma, l0.norm_mha.beta,
                             L[0]["qkv_fpk"], L[0]["qkv_b"])
    n = len(L)
    for li, (l, W) in enumerate(zip(enc.encoders, L)):
        ctx = ops.relpos_attention(qkv, pos_all[:, li * 256:(li + 1) * 256], W["u"], W["v"], mask2d, b, t2, 4, 64)
        _, a = ops.gemm_packed_ln(ctx, W["o_pk"], l.norm_conv.gamma, l.norm_conv.beta, ln_row_scale=mask_rows, bias=W["o_b"],
                                  residual=x, out=x)
        ops.convmodule(a, W["pw1_pk"], W["pw1_b"], W["dw_w"], W["bn_scale"], W["bn_shift"], W["pw2_pk"], W["pw2_b"], mask_rows, x,
                       b, t2)
        if li + 1 < n:
            ln, Wn = enc.encoders[li + 1], L[li + 1]
            qkv = ops.ffn_packed_pair(W["ff_pk"], W["ff_b1"], W["ff_b2"], Wn["ffm_pk"], Wn["ffm_b1"], Wn["ffm_b2"], x,
                                      (l.norm_ff.gamma, l.norm_ff.beta), (l.norm_final.gamma, l.norm_final.beta),
                                      (ln.norm_ff_macaron.gamma, ln.norm_ff_macaron.beta), (ln.norm_mha.gamma, ln.norm_mha.beta),
                                      qkv=(Wn["qkv_fpk"], Wn["qkv_b"]))
        else:
            x = ops.ffn_packed(None, W["ff_pk"], W["ff_b1"], W["ff_b2"], x, l.norm_final.gamma, l.norm_final.beta,
                               enc.after_norm.gamma, enc.after_norm.beta, out_dtype=torch.float32,
                               ln_in=(l.norm_ff.gamma, l.norm_ff.beta))
    return x


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


full = lambda: blocks(x0.clone(), mask, B)
print("full batch, one stream: %.1f us" % timeit(full))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h = B // 2


def split():
    cur = torch.cuda.current_stream()
    xa, xb = x0[:h * t2].clone(), x0[h * t2:].clone()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    # generator-style interleaving of the two halves' launches so that both queues are fed
    ga, gb = gen_blocks(xa, mask[:h], h, s1), gen_blocks(xb, mask[h:], h, s2)
    done_a = done_b = False
    while not (done_a and done_b):
        if not done_a:
            done_a = next(ga, "end") == "end"
        if not done_b:
            done_b = next(gb, "end") == "end"
    cur.wait_stream(s1)
    cur.wait_stream(s2)


def gen_blocks(x, mask2d, b, stream):
    mask_rows = mask2d.reshape(-1)
    L = P["layers"]
    l0 = enc.encoders[0]
    with torch.cuda.stream(stream):
        a = ops.layernorm(x, l0.norm_ff_macaron.gamma, l0.norm_ff_macaron.beta)
        qkv = ops.ffn_packed_qkv(a, L[0]["ffm_pk"], L[0]["ffm_b1"], L[0]["ffm_b2"], x, l0.norm_mha.gamma, l0.norm_mha.beta,
                                 L[0]["qkv_fpk"], L[0]["qkv_b"])
    yield 1
    n = len(L)
    for li, (l, W) in enumerate(zip(enc.encoders, L)):
        with torch.cuda.stream(stream):
            ctx = ops.relpos_attention(qkv, pos_all[:, li * 256:(li + 1) * 256], W["u"], W["v"], mask2d, b, t2, 4, 64)
        yield 1
        with torch.cuda.stream(stream):
            _, a = ops.gemm_packed_ln(ctx, W["o_pk"], l.norm_conv.gamma, l.norm_conv.beta, ln_row_scale=mask_rows, bias=W["o_b"],
                                      residual=x, out=x)
        yield 1
        with torch.cuda.stream(stream):
            ops.convmodule(a, W["pw1_pk"], W["pw1_b"], W["dw_w"], W["bn_scale"], W["bn_shift"], W["pw2_pk"], W["pw2_b"],
                           mask_rows, x, b, t2)
        yield 1
        with torch.cuda.stream(stream):
            if li + 1 < n:
                ln, Wn = enc.encoders[li + 1], L[li + 1]
                qkv = ops.ffn_packed_pair(W["ff_pk"], W["ff_b1"], W["ff_b2"], Wn["ffm_pk"], Wn["ffm_b1"], Wn["ffm_b2"], x,
                                          (l.norm_ff.gamma, l.norm_ff.beta), (l.norm_final.gamma, l.norm_final.beta),
                                          (ln.norm_ff_macaron.gamma, ln.norm_ff_macaron.beta),
                                          (ln.norm_mha.gamma, ln.norm_mha.beta), qkv=(Wn["qkv_fpk"], Wn["qkv_b"]))
            else:
                ops.ffn_packed(None, W["ff_pk"], W["ff_b1"], W["ff_b2"], x, l.norm_final.gamma, l.norm_final.beta,
                               enc.after_norm.gamma, enc.after_norm.beta, out_dtype=torch.float32,
                               ln_in=(l.norm_ff.gamma, l.norm_ff.beta))
        yield 1


print("two halves, two streams: %.1f us" % timeit(split))
# offset variant: the second half starts two launches later
def split_offset():
    cur = torch.cuda.current_stream()
    xa, xb = x0[:h * t2].clone(), x0[h * t2:].clone()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    ga, gb = gen_blocks(xa, mask[:h], h, s1), gen_blocks(xb, mask[h:], h, s2)
    next(ga), next(ga)
    done_a = done_b = False
    while not (done_a and done_b):
        if not done_b:
            done_b = next(gb, "end") == "end"
        if not done_a:
            done_a = next(ga, "end") == "end"
    cur.wait_stream(s1)
    cur.wait_stream(s2)


print("two halves, two streams, offset by two launches: %.1f us" % timeit(split_offset))
halfonly = lambda: blocks(x0[:h * t2].clone(), mask[:h], h)
print("one half alone: %.1f us" % timeit(halfonly))
