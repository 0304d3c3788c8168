"""CPU ORACLE — test infrastructure, NOT a product path.

PyTorch-CPU float32 eager restatement of the Conformer encoder (+ CTC head) of
mindspore-lab/mindaudio, written from the reference's layer definitions.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

**Parity unpinned.**  The reference model is 100 % mindspore.nn / mindspore.ops
(mindaudio/models/conformer.py, mindaudio/models/layers/*.py, mindaudio/loss/ctc_loss.py); MindSpore 2.3.0
is not installable here and the reference has no test or golden vector for any model code
(SURVEY §4, §8c).  What pins this file is the source it restates, line by line, including the quirks:

  * rel-pos attention has NO relative shift and uses rows 0..T-1 of the absolute sinusoid table as
    "relative" embeddings (layers/attention.py:226-235, layers/embedding.py:86-88);
  * the mask is additive -10000 (not -inf) and masked rows are not zeroed after softmax
    (layers/attention.py:100-109);
  * BatchNorm1d in the conv module runs over (B*T, C) rows, padded frames included
    (layers/convolution.py:113-121);
  * Swish is hard-wired whatever `activation_type` says (models/conformer.py:327);
  * LayerNorm: biased variance, eps inside the sqrt (layers/layernorm.py:53-60);
  * Dense/Conv init = Kaiming-uniform(a=sqrt 5) + U(+-1/sqrt(fan_in)) bias, i.e. PyTorch's defaults
    (layers/dense.py:40-50, layers/conv1d.py:54-72); pos_bias_u/v Xavier-uniform (attention.py:173-178).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def sinusoid_table(max_len, d_model):
    """layers/embedding.py:36-44 (float32 arithmetic as in the reference's NumPy code)."""
    pe = np.zeros((max_len, d_model))
    position = np.expand_dims(np.arange(0, max_len, dtype=np.float32), 1)
    div_term = np.exp(np.arange(0, d_model, 2, dtype=np.float32) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = np.sin(position * div_term)
    pe[:, 1::2] = np.cos(position * div_term)
    return torch.from_numpy(pe.astype(np.float32))


class LayerNorm(nn.Module):
    """layers/layernorm.py:10-60."""

    def __init__(self, size, eps=1e-5):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(size))
        self.beta = nn.Parameter(torch.zeros(size))
        self.eps = eps

    def forward(self, x):
        mean = x.mean(-1, keepdim=True)
        diff = x - mean
        var = (diff * diff).mean(-1, keepdim=True)
        return diff / torch.sqrt(var + self.eps) * self.gamma + self.beta


class PositionwiseFeedForward(nn.Module):
    """layers/positionwise_feed_forward.py:9-46 with Swish (layers/swish.py:14-16)."""

    def __init__(self, idim, hidden, dropout):
        super().__init__()
        self.w_1 = nn.Linear(idim, hidden)
        self.w_2 = nn.Linear(hidden, idim)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        h = self.w_1(x)
        return self.w_2(self.dropout(h * torch.sigmoid(h)))


class RelPositionMultiHeadedAttention(nn.Module):
    """layers/attention.py:17-237."""

    def __init__(self, n_head, n_feat, dropout):
        super().__init__()
        self.h, self.d_k = n_head, n_feat // n_head
        self.linear_q = nn.Linear(n_feat, n_feat)
        self.linear_k = nn.Linear(n_feat, n_feat)
        self.linear_v = nn.Linear(n_feat, n_feat)
        self.linear_out = nn.Linear(n_feat, n_feat)
        self.linear_pos = nn.Linear(n_feat, n_feat, bias=False)
        self.pos_bias_u = nn.Parameter(torch.empty(self.h, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.empty(self.h, self.d_k))
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, mask, pos_emb):
        b, t, _ = x.shape
        q = self.linear_q(x).view(b, t, self.h, self.d_k)
        k = self.linear_k(x).view(b, t, self.h, self.d_k).transpose(1, 2)
        v = self.linear_v(x).view(b, t, self.h, self.d_k).transpose(1, 2)
        p = self.linear_pos(pos_emb).view(pos_emb.shape[0], -1, self.h, self.d_k).transpose(1, 2)
        q_u = (q + self.pos_bias_u).transpose(1, 2)
        q_v = (q + self.pos_bias_v).transpose(1, 2)
        ac = q_u @ k.transpose(-1, -2)
        bd = q_v @ p.transpose(-1, -2)  # no rel_shift (attention.py:232-234)
        scores = (ac + bd) * (1.0 / math.sqrt(self.d_k))
        if mask is not None:  # (B, 1, T) -> additive -10000 where mask == 0 (attention.py:100-107)
            scores = scores + (mask.unsqueeze(1) == 0).to(scores.dtype) * (-10000.0)
        attn = self.dropout(torch.softmax(scores, dim=-1))
        ctx = (attn @ v).transpose(1, 2).reshape(b, t, self.h * self.d_k)
        return self.linear_out(ctx)


class ConvolutionModule(nn.Module):
    """layers/convolution.py:14-129 with batch_norm (the default, models/conformer.py:310)."""

    def __init__(self, channels, kernel_size):
        super().__init__()
        self.pointwise_conv1 = nn.Conv1d(channels, 2 * channels, 1)
        self.depthwise_conv = nn.Conv1d(channels, channels, kernel_size, padding=(kernel_size - 1) // 2,
                                        groups=channels)
        # MindSpore nn.BatchNorm1d defaults: eps 1e-5, momentum 0.9 (= torch momentum 0.1)
        self.norm = nn.BatchNorm1d(channels, eps=1e-5, momentum=0.1)
        self.pointwise_conv2 = nn.Conv1d(channels, channels, 1)

    def forward(self, x, mask_pad):
        x = x.transpose(1, 2)  # (B, C, T)
        if mask_pad is not None:
            x = x * mask_pad
        x = self.pointwise_conv1(x)
        out, gate = x.chunk(2, dim=1)  # layers/glu.py:24-28
        x = out * torch.sigmoid(gate)
        x = self.depthwise_conv(x)
        b, c, t = x.shape
        y = self.norm(x.transpose(1, 2).reshape(b * t, c))  # rows = B*T, padded frames included
        y = y * torch.sigmoid(y)
        x = y.reshape(b, t, c).transpose(1, 2)
        x = self.pointwise_conv2(x)
        if mask_pad is not None:
            x = x * mask_pad
        return x.transpose(1, 2)


class ConformerEncoderLayer(nn.Module):
    """models/conformer.py:25-161 (normalize_before=True, concat_after=False: the shipped config)."""

    def __init__(self, size, heads, linear_units, kernel, dropout, att_dropout):
        super().__init__()
        self.self_attn = RelPositionMultiHeadedAttention(heads, size, att_dropout)
        self.feed_forward = PositionwiseFeedForward(size, linear_units, dropout)
        self.feed_forward_macaron = PositionwiseFeedForward(size, linear_units, dropout)
        self.conv_module = ConvolutionModule(size, kernel)
        self.norm_ff = LayerNorm(size)
        self.norm_mha = LayerNorm(size)
        self.norm_ff_macaron = LayerNorm(size)
        self.norm_conv = LayerNorm(size)
        self.norm_final = LayerNorm(size)
        self.dropout = nn.Dropout(dropout)
        self.ff_scale = 0.5

    def forward(self, x, mask, pos_emb, mask_pad):
        x = x + self.ff_scale * self.dropout(self.feed_forward_macaron(self.norm_ff_macaron(x)))
        x = x + self.dropout(self.self_attn(self.norm_mha(x), mask, pos_emb))
        x = x + self.dropout(self.conv_module(self.norm_conv(x), mask_pad))
        x = x + self.ff_scale * self.dropout(self.feed_forward(self.norm_ff(x)))
        return self.norm_final(x)


class Conv2dSubsampling4(nn.Module):
    """layers/subsampling.py:21-78 + RelPositionalEncoding (layers/embedding.py:65-88)."""

    def __init__(self, idim, odim, pos_dropout, max_len=5000):
        super().__init__()
        self.conv1 = nn.Conv2d(1, odim, 3, 2)
        self.conv2 = nn.Conv2d(odim, odim, 3, 2)
        self.out = nn.Linear(odim * (((idim - 1) // 2 - 1) // 2), odim)
        self.xscale = math.sqrt(odim)
        self.register_buffer("pe", sinusoid_table(max_len, odim), persistent=False)
        self.dropout = nn.Dropout(pos_dropout)

    def forward(self, x):
        x = x.unsqueeze(1)
        x = F.relu(self.conv2(F.relu(self.conv1(x))))
        b, c, t, f = x.shape
        x = self.out(x.transpose(1, 2).reshape(b, t, c * f))
        x = x * self.xscale
        pos_emb = self.pe[:t].unsqueeze(0)
        return self.dropout(x), self.dropout(pos_emb)


class ConformerEncoder(nn.Module):
    """models/conformer.py:164-379 (input_layer='conv2d', pos_enc_layer_type='rel_pos')."""

    def __init__(self, input_size, output_size=256, attention_heads=4, linear_units=2048, num_blocks=6,
                 dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.0,
                 cnn_module_kernel=15, cmvn_mean=None, cmvn_istd=None):
        super().__init__()
        self.embed = Conv2dSubsampling4(input_size, output_size, positional_dropout_rate)
        self.encoders = nn.ModuleList([
            ConformerEncoderLayer(output_size, attention_heads, linear_units, cnn_module_kernel, dropout_rate,
                                  attention_dropout_rate) for _ in range(num_blocks)])
        self.after_norm = LayerNorm(output_size)
        if cmvn_mean is not None:
            self.register_buffer("cmvn_mean", torch.as_tensor(cmvn_mean, dtype=torch.float32))
            self.register_buffer("cmvn_istd", torch.as_tensor(cmvn_istd, dtype=torch.float32))
        else:
            self.cmvn_mean = None

    def forward(self, xs, masks, xs_chunk_masks=None):
        """xs (B, T, D); masks (B, 1, T') float/bool, already subsampled ([:, :, :-2:2][:, :, :-2:2],
        examples/conformer/dataset.py:620-632). Returns (B, T', 256), masks."""
        if xs_chunk_masks is None:
            xs_chunk_masks = masks
        if self.cmvn_mean is not None:  # layers/cmvn.py:33-35
            xs = (xs - self.cmvn_mean) * self.cmvn_istd
        xs, pos_emb = self.embed(xs)
        for layer in self.encoders:
            xs = layer(xs, xs_chunk_masks, pos_emb, masks)
        return self.after_norm(xs), masks


class CTC(nn.Module):
    """mindaudio/loss/ctc_loss.py:10-64: Dense -> fp32 log_softmax -> CTCLossV2(blank 0, zero_infinity) -> sum / B."""

    def __init__(self, odim, eprojs):
        super().__init__()
        self.ctc_lo = nn.Linear(eprojs, odim)

    def log_probs(self, hs):
        return torch.log_softmax(self.ctc_lo(hs).float(), dim=2)

    def forward(self, hs, hlens, ys_pad, ys_lens):
        lp = self.log_probs(hs).transpose(0, 1)
        loss = F.ctc_loss(lp, ys_pad, hlens, ys_lens, blank=0, reduction="none", zero_infinity=True)
        return loss.sum() / hs.shape[0]


def subsample_mask(mask):
    """examples/conformer/dataset.py:620-632: (B, 1, T) -> (B, 1, T') for two stride-2 valid 3x3 convs."""
    return mask[:, :, :-2:2][:, :, :-2:2]


def forward_flops_per_utt(t_in=1000, idim=80, d=256, heads=4, ff=2048, layers=12, kernel=15):
    """Algorithmic forward FLOPs (2 * MACs) of the encoder for one utterance — SURVEY §3.5."""
    t1, f1 = (t_in - 3) // 2 + 1, (idim - 3) // 2 + 1
    t2, f2 = (t1 - 3) // 2 + 1, (f1 - 3) // 2 + 1
    sub = 2 * (t1 * f1 * d * 9 + t2 * f2 * d * d * 9 + t2 * d * f2 * d)
    per_layer = 2 * t2 * (2 * 2 * d * ff + 4 * d * d) + 2 * t2 * d * d  # FFN x2, q k v out; pos proj once (1, T, d)
    per_layer += 2 * 3 * heads * t2 * t2 * (d // heads)  # ac, bd, attn @ v
    per_layer += 2 * t2 * (d * 2 * d + d * kernel + d * d)  # conv module
    return sub + layers * per_layer


# ---- attention decoder + label smoothing: the hybrid loss of examples/conformer/asr_model.py:75-209 -----------------
class MultiHeadedAttention(nn.Module):
    """layers/attention.py:17-157.  Quirk: q and k are BOTH multiplied by 1/sqrt(d_k) before the product, i.e. the
    scores are divided by d_k (attention.py:150-152); additive -10000 mask of shape (B, 1, T2) or (B, T1, T2)."""

    def __init__(self, n_head, n_feat):
        super().__init__()
        self.h, self.d_k = n_head, n_feat // n_head
        self.linear_q = nn.Linear(n_feat, n_feat)
        self.linear_k = nn.Linear(n_feat, n_feat)
        self.linear_v = nn.Linear(n_feat, n_feat)
        self.linear_out = nn.Linear(n_feat, n_feat)

    def forward(self, query, key, value, mask):
        b = query.shape[0]
        q = self.linear_q(query).view(b, -1, self.h, self.d_k).transpose(1, 2)
        k = self.linear_k(key).view(b, -1, self.h, self.d_k).transpose(1, 2)
        v = self.linear_v(value).view(b, -1, self.h, self.d_k).transpose(1, 2)
        s = 1.0 / math.sqrt(self.d_k)
        scores = (q * s) @ (k * s).transpose(-1, -2)
        if mask is not None:
            scores = scores + (mask.unsqueeze(1) == 0).to(scores.dtype) * (-10000.0)
        ctx = (torch.softmax(scores, dim=-1) @ v).transpose(1, 2).reshape(b, -1, self.h * self.d_k)
        return self.linear_out(ctx)


class DecoderFeedForward(nn.Module):
    """PositionwiseFeedForward with the decoder's ReLU (models/conformer.py:521)."""

    def __init__(self, idim, hidden, dropout):
        super().__init__()
        self.w_1 = nn.Linear(idim, hidden)
        self.w_2 = nn.Linear(hidden, idim)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        return self.w_2(self.dropout(torch.relu(self.w_1(x))))


class DecoderLayer(nn.Module):
    """models/conformer.py:382-497 (normalize_before=True, concat_after=False); LayerNorm eps 1e-12 (:417-419)."""

    def __init__(self, size, heads, linear_units, dropout):
        super().__init__()
        self.self_attn = MultiHeadedAttention(heads, size)
        self.src_attn = MultiHeadedAttention(heads, size)
        self.feed_forward = DecoderFeedForward(size, linear_units, dropout)
        self.norm1, self.norm2, self.norm3 = LayerNorm(size, 1e-12), LayerNorm(size, 1e-12), LayerNorm(size, 1e-12)
        self.dropout = nn.Dropout(dropout)

    def forward(self, tgt, tgt_mask, memory, memory_mask):
        t = self.norm1(tgt)
        x = tgt + self.dropout(self.self_attn(t, t, t, tgt_mask))
        x = x + self.dropout(self.src_attn(self.norm2(x), memory, memory, memory_mask))
        return x + self.dropout(self.feed_forward(self.norm3(x)))


class TransformerDecoder(nn.Module):
    """models/conformer.py:500-639: Embedding -> x*sqrt(d) + pe -> dropout, N DecoderLayers, after_norm, output layer."""

    def __init__(self, vocab_size, encoder_output_size, attention_heads=4, linear_units=2048, num_blocks=6,
                 dropout_rate=0.1, positional_dropout_rate=0.1, max_len=5000):
        super().__init__()
        d = encoder_output_size
        self.embed = nn.Embedding(vocab_size, d)
        self.xscale = math.sqrt(d)
        self.register_buffer("pe", sinusoid_table(max_len, d), persistent=False)
        self.pos_dropout = nn.Dropout(positional_dropout_rate)
        self.decoders = nn.ModuleList([DecoderLayer(d, attention_heads, linear_units, dropout_rate)
                                       for _ in range(num_blocks)])
        self.after_norm = LayerNorm(d, 1e-12)
        self.output_layer = nn.Linear(d, vocab_size)

    def forward(self, memory, memory_mask, ys_in_pad, ys_masks):
        x = self.pos_dropout(self.embed(ys_in_pad) * self.xscale + self.pe[:ys_in_pad.shape[1]].unsqueeze(0))
        for layer in self.decoders:
            x = layer(x, ys_masks, memory, memory_mask)
        return self.output_layer(self.after_norm(x))


def label_smoothing_loss(x, target, target_masks, smoothing, normalize_length=False):
    """loss/label_smoothing_loss.py:84-117: x (B, L, V) logits, target (B, L) with -1 padding, target_masks (B, 1, L)."""
    b, _, v = x.shape
    x = x.reshape(-1, v)
    tm = target_masks.reshape(-1).to(x.dtype)
    tgt = (target.reshape(-1).to(x.dtype) * tm).long()
    true = torch.full_like(x, smoothing / (v - 1))
    true.scatter_(1, tgt.unsqueeze(1), 1.0 - smoothing)
    kl = true * (torch.log(true) - torch.log_softmax(x, dim=1))
    kl = kl * tm.unsqueeze(1)
    return kl.sum() / (tm.sum() if normalize_length else b)


def th_accuracy(pad_outputs, pad_targets, ys_masks):
    """asr_model.py:188-209."""
    pred = pad_outputs.argmax(2)
    m = ys_masks.squeeze(1).to(torch.float32)
    return ((pred == pad_targets).to(torch.float32) * m).sum() / m.sum()


def hybrid_loss(encoder, ctc, decoder, batch, ctc_weight=0.3, lsm_weight=0.1, length_normalized_loss=False):
    """ASRModelWithAcc.construct (asr_model.py:75-153) with reverse_weight 0: returns (loss, acc_att, loss_ctc, loss_att).
    `batch` = the 11 collate columns; length_normalized_loss = LabelSmoothingLoss(normalize_length=...) (asr_model.py:57-62)."""
    xs_pad, ys_pad, ys_in_pad, ys_out_pad, _, _, xs_masks, ys_sub_masks, ys_masks, ys_lengths, xs_chunk_masks = batch
    enc, enc_mask = encoder(xs_pad, xs_masks, xs_chunk_masks)
    hlens = enc_mask.reshape(enc_mask.shape[0], -1).sum(1).to(torch.int32)
    loss_ctc = ctc(enc, hlens, ys_pad.clamp(min=0).long(), ys_lengths.long())
    dec_out = decoder(enc, enc_mask, ys_in_pad.long(), ys_sub_masks)
    loss_att = label_smoothing_loss(dec_out, ys_out_pad, ys_masks, lsm_weight, length_normalized_loss)
    acc = th_accuracy(dec_out, ys_out_pad, ys_masks)
    return ctc_weight * loss_ctc + (1 - ctc_weight) * loss_att, acc, loss_ctc, loss_att
