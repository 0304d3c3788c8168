"""TransformerDecoder parameter container — mirror of mindaudio.models.conformer.TransformerDecoder
(models/conformer.py:500-639) for the attention branch of the hybrid loss.  Forward and backward run inside
mindaudio_amd.train.engine (HIP kernels through the C-ABI); the PyTorch layers here only hold the float32 masters and
their reference initialisation."""
import math

import torch.nn as nn

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
        if encoder_output_size != 256 or encoder_output_size // attention_heads != 64:
            raise NotImplementedError("kernels are built for d_model 256 with 64-wide heads")
        d = encoder_output_size
        self.d, self.heads, self.vocab_size = d, attention_heads, vocab_size
        self.dropout_rate, self.positional_dropout_rate = dropout_rate, positional_dropout_rate
        self.embed = nn.Embedding(vocab_size, d)
        self.xscale = math.sqrt(d)
        self.register_buffer("pe", _sinusoid_table(max_len, d), persistent=False)
        self.decoders = nn.ModuleList([_DecoderLayer(d, linear_units) for _ in range(num_blocks)])
        self.after_norm = _LN(d)
        self.output_layer = nn.Linear(d, vocab_size)
