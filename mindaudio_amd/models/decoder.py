"""TransformerDecoder — mirror of mindaudio.models.conformer.TransformerDecoder (models/conformer.py:500-639), the attention
branch of the hybrid CTC / attention loss.  `forward` is the evaluation forward (dropout off) on HIP kernels through the C-ABI;
the training-mode forward + backward run inside mindaudio_amd.train.engine on the same kernels.  The PyTorch layers only hold the
float32 masters and their reference initialisation (their own forward() is never called)."""
import math

import torch
import torch.nn as nn

from .. import _lib, ops
from .conformer import _LN, _sinusoid_table


class _MHA(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.linear_q = nn.Linear(d, d)
        self.linear_k = nn.Linear(d, d)
        self.linear_v = nn.Linear(d, d)
        self.linear_out = nn.Linear(d, d)


class _FF(nn.Module):
    def __init__(self, d, hidden):
        super().__init__()
        self.w_1 = nn.Linear(d, hidden)
        self.w_2 = nn.Linear(hidden, d)


class _DecoderLayer(nn.Module):
    def __init__(self, d, hidden):
        super().__init__()
        self.self_attn = _MHA(d)
        self.src_attn = _MHA(d)
        self.feed_forward = _FF(d, hidden)
        self.norm1, self.norm2, self.norm3 = _LN(d), _LN(d), _LN(d)


class TransformerDecoder(nn.Module):
    """Constructor arguments of the reference (models/conformer.py:523-538); input_layer "embed", pre-norm, no concat."""

    def __init__(self, vocab_size, encoder_output_size, attention_heads=4, linear_units=2048, num_blocks=6,
                 dropout_rate=0.1, positional_dropout_rate=0.1, self_attention_dropout_rate=0.0,
                 src_attention_dropout_rate=0.0, input_layer="embed", use_output_layer=True, normalize_before=True,
                 concat_after=False, compute_type=None, max_len=5000):
        super().__init__()
        if input_layer != "embed" or not use_output_layer or not normalize_before or concat_after:
            raise NotImplementedError("only the shipped decoder configuration is built")
        if self_attention_dropout_rate or src_attention_dropout_rate:
            raise NotImplementedError("attention-weight dropout is 0.0 in the shipped configuration")
        if encoder_output_size not in (256, 512, 768, 1024) or encoder_output_size != attention_heads * 64:
            raise NotImplementedError("kernels are built for 64-wide heads with d_model 256, 512, 768 or 1024")
        d = encoder_output_size
        self.d, self.heads, self.vocab_size = d, attention_heads, vocab_size
        self.dropout_rate, self.positional_dropout_rate = dropout_rate, positional_dropout_rate
        self.embed = nn.Embedding(vocab_size, d)
        self.xscale = math.sqrt(d)
        self.register_buffer("pe", _sinusoid_table(max_len, d), persistent=False)
        self.decoders = nn.ModuleList([_DecoderLayer(d, linear_units) for _ in range(num_blocks)])
        self.after_norm = _LN(d)
        self.output_layer = nn.Linear(d, vocab_size)
        self._prepared = None
        self.register_load_state_dict_post_hook(lambda module, _keys: setattr(module, "_prepared", None))

    @torch.no_grad()
    def prepare(self):
        """bf16 copies of the matmul weights (q/k/v of the self-attention and k/v of the source attention fused)."""
        bf = torch.bfloat16
        f = lambda p: p.detach().float().contiguous()  # noqa: E731
        layers = []
        for l in self.decoders:
            sa, ca, ff = l.self_attn, l.src_attn, l.feed_forward
            layers.append({
                "sa_qkv_w": torch.cat([sa.linear_q.weight, sa.linear_k.weight, sa.linear_v.weight], 0).detach().to(bf).contiguous(),
                "sa_qkv_b": f(torch.cat([sa.linear_q.bias, sa.linear_k.bias, sa.linear_v.bias], 0)),
                "sa_o_w": sa.linear_out.weight.detach().to(bf).contiguous(), "sa_o_b": f(sa.linear_out.bias),
                "ca_q_w": ca.linear_q.weight.detach().to(bf).contiguous(), "ca_q_b": f(ca.linear_q.bias),
                "ca_kv_w": torch.cat([ca.linear_k.weight, ca.linear_v.weight], 0).detach().to(bf).contiguous(),
                "ca_kv_b": f(torch.cat([ca.linear_k.bias, ca.linear_v.bias], 0)),
                "ca_o_w": ca.linear_out.weight.detach().to(bf).contiguous(), "ca_o_b": f(ca.linear_out.bias),
                "ff_w1": ff.w_1.weight.detach().to(bf).contiguous(), "ff_b1": f(ff.w_1.bias),
                "ff_w2": ff.w_2.weight.detach().to(bf).contiguous(), "ff_b2": f(ff.w_2.bias),
            })
        vp = (self.vocab_size + 63) // 64 * 64
        out_b = torch.zeros(vp, dtype=torch.float32, device=self.output_layer.bias.device)
        out_b[:self.vocab_size] = self.output_layer.bias.detach().float()
        self._prepared = {"layers": layers, "embed": f(self.embed.weight),
                          "out_w": self.output_layer.weight.detach().to(bf).contiguous(), "out_b": out_b}
        return self

    @torch.no_grad()
    def forward(self, memory, memory_mask, ys_in_pad, ys_masks, r_ys_in_pad=None):
        """TransformerDecoder.construct (models/conformer.py:604-639), evaluation mode: memory (B, T', 256) float32, memory_mask
        (B, 1, T'), ys_in_pad (B, L) token ids, ys_masks (B, L, L) -> (scores before softmax (B, L, V) float32, tensor0).
        Pre-norm layers with eps 1e-12 (models/conformer.py:417-419, 548); both attentions scale q AND k by 1/sqrt(d_k)
        (layers/attention.py:150-152), i.e. scores / d_k."""
        from ..train import kernels as K

        if self.training:
            raise NotImplementedError("the training-mode decoder (dropout) runs inside mindaudio_amd.train.engine")
        if self._prepared is None:
            self.prepare()
        P = self._prepared
        f32 = torch.float32
        b, t2, d = memory.shape
        L1 = ys_in_pad.shape[1]
        dk = d // self.heads
        scale, eps = 1.0 / dk, 1e-12
        mem_bf = ops.cast_bf16(memory.reshape(b * t2, d).to(f32).contiguous())
        emask = memory_mask.reshape(b, t2).to(f32).contiguous()
        sub = ys_masks.to(f32).contiguous()
        toks = ys_in_pad.to(torch.int32).contiguous().reshape(-1)
        pe = self.pe[:L1].to(f32).contiguous()
        x = K.embed_posenc(toks, P["embed"], pe, L1, self.xscale, 0.0, 0, 0)            # (B*L, d) float32
        for l, W in zip(self.decoders, P["layers"]):
            a = ops.layernorm(x, l.norm1.gamma, l.norm1.beta, eps=eps)
            qkv = ops.gemm(a, W["sa_qkv_w"], bias=W["sa_qkv_b"])
            ctx, _ = K.mha_small_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], sub, 2, b, L1, L1, scale, self.heads, dk)
            ops.gemm(ctx, W["sa_o_w"], bias=W["sa_o_b"], residual=x, out_dtype=f32, out=x)
            a = ops.layernorm(x, l.norm2.gamma, l.norm2.beta, eps=eps)
            q = ops.gemm(a, W["ca_q_w"], bias=W["ca_q_b"])
            kv = ops.gemm(mem_bf, W["ca_kv_w"], bias=W["ca_kv_b"])
            ctx, _ = K.mha_small_fwd(q, kv[:, :d], kv[:, d:], emask, 1, b, L1, t2, scale, self.heads, dk)
            ops.gemm(ctx, W["ca_o_w"], bias=W["ca_o_b"], residual=x, out_dtype=f32, out=x)
            a = ops.layernorm(x, l.norm3.gamma, l.norm3.beta, eps=eps)
            h = ops.gemm(a, W["ff_w1"], bias=W["ff_b1"], act=_lib.ACT_RELU)
            ops.gemm(h, W["ff_w2"], bias=W["ff_b2"], residual=x, out_dtype=f32, out=x)
        y = ops.layernorm(x, self.after_norm.gamma, self.after_norm.beta, eps=eps)
        vp = P["out_b"].numel()
        logits = torch.empty((b * L1, vp), dtype=f32, device=memory.device)
        ops.gemm(y, P["out_w"], bias=P["out_b"], out_dtype=f32, out=logits[:, :self.vocab_size])
        return logits[:, :self.vocab_size].view(b, L1, self.vocab_size), torch.zeros(1, device=memory.device)
