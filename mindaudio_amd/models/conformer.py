"""ConformerEncoder on MI355X — mirror of mindaudio.models.conformer.ConformerEncoder
(mindaudio/models/conformer.py:261-379 over BaseEncoder :164-258), forward pass in hand-written HIP kernels.

The module keeps float32 master parameters under the same structure as the reference cells (PyTorch layers are
used as parameter containers only — their forward() is never called); `forward` runs bf16 MFMA GEMMs with fused
epilogues, a fused rel-pos attention kernel, a fused GLU/depthwise/BatchNorm/Swish kernel and LayerNorm kernels
through the C-ABI (include/mindaudio_amd.h).  The residual stream, LayerNorm statistics and softmax are float32.

forward() in eval mode (dropout off, BatchNorm running statistics) is the path the `utterances/s fbanks+Conformer fwd` metric
measures; in train mode it applies dropout and BatchNorm batch statistics (the forward half of the training step, whose backward
and optimizer live in mindaudio_amd.train.engine).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import _lib, ops


def _sinusoid_table(max_len, d_model):
    """layers/embedding.py:36-44."""
    pe = np.zeros((max_len, d_model))
    position = np.expand_dims(np.arange(0, max_len, dtype=np.float32), 1)
    div_term = np.exp(np.arange(0, d_model, 2, dtype=np.float32) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = np.sin(position * div_term)
    pe[:, 1::2] = np.cos(position * div_term)
    return torch.from_numpy(pe.astype(np.float32))


class _LN(nn.Module):
    def __init__(self, size):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(size))
        self.beta = nn.Parameter(torch.zeros(size))


class _FFN(nn.Module):
    def __init__(self, d, hidden):
        super().__init__()
        self.w_1 = nn.Linear(d, hidden)
        self.w_2 = nn.Linear(hidden, d)


class _Attn(nn.Module):
    def __init__(self, heads, d):
        super().__init__()
        self.linear_q = nn.Linear(d, d)
        self.linear_k = nn.Linear(d, d)
        self.linear_v = nn.Linear(d, d)
        self.linear_out = nn.Linear(d, d)
        self.linear_pos = nn.Linear(d, d, bias=False)
        self.pos_bias_u = nn.Parameter(torch.empty(heads, d // heads))
        self.pos_bias_v = nn.Parameter(torch.empty(heads, d // heads))
        nn.init.xavier_uniform_(self.pos_bias_u)  # layers/attention.py:173-178
        nn.init.xavier_uniform_(self.pos_bias_v)


class _ConvModule(nn.Module):
    def __init__(self, d, kernel):
        super().__init__()
        self.pointwise_conv1 = nn.Conv1d(d, 2 * d, 1)
        self.depthwise_conv = nn.Conv1d(d, d, kernel, padding=(kernel - 1) // 2, groups=d)
        self.norm = nn.BatchNorm1d(d, eps=1e-5, momentum=0.1)
        self.pointwise_conv2 = nn.Conv1d(d, d, 1)


class _Layer(nn.Module):
    def __init__(self, d, heads, hidden, kernel):
        super().__init__()
        self.self_attn = _Attn(heads, d)
        self.feed_forward = _FFN(d, hidden)
        self.feed_forward_macaron = _FFN(d, hidden)
        self.conv_module = _ConvModule(d, kernel)
        self.norm_ff = _LN(d)
        self.norm_mha = _LN(d)
        self.norm_ff_macaron = _LN(d)
        self.norm_conv = _LN(d)
        self.norm_final = _LN(d)


class _Embed(nn.Module):
    def __init__(self, idim, odim):
        super().__init__()
        self.conv1 = nn.Conv2d(1, odim, 3, 2)
        self.conv2 = nn.Conv2d(odim, odim, 3, 2)
        self.out = nn.Linear(odim * (((idim - 1) // 2 - 1) // 2), odim)


class ConformerEncoder(nn.Module):
    """Same constructor arguments as the reference (models/conformer.py:293-313).  `global_cmvn` is a
    (mean, istd) pair of arrays (layers/cmvn.py); `compute_type` is accepted for signature parity — matmul
    inputs are bf16, accumulation float32 (the float32 validation mode is a switch of the training engine)."""

    def __init__(self, input_size, output_size=256, attention_heads=4, linear_units=2048, num_blocks=6,
                 dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="conv2d",
                 pos_enc_layer_type="rel_pos", normalize_before=True, feature_norm=True, concat_after=False,
                 activation_type="relu", cnn_module_kernel=15, cnn_module_norm="batch_norm", global_cmvn=None,
                 compute_type=None, max_len=5000):
        super().__init__()
        if input_layer != "conv2d" or pos_enc_layer_type != "rel_pos" or not normalize_before or concat_after \
                or cnn_module_norm != "batch_norm":
            raise NotImplementedError("only the shipped conformer.yaml configuration is on the hot path: conv2d "
                                      "input, rel_pos, pre-norm, no concat, batch_norm conv module")
        if output_size not in (256, 512, 768, 1024) or output_size != attention_heads * 64:
            raise NotImplementedError("kernels are built for 64-wide heads with d_model 256 (Conformer-small: the fused, "
                                      "packed-weight launches), 512, 768 or 1024 (one launch per reference cell)")
        self.d, self.heads, self.idim = output_size, attention_heads, input_size
        self.kernel = cnn_module_kernel
        self.dropout_rate, self.positional_dropout_rate = float(dropout_rate), float(positional_dropout_rate)
        self._bn_dirty = False
        self.seed, self._train_calls = 777, 0  # dropout stream of the training-mode forward (examples/conformer/train.py:56)
        self.embed = _Embed(input_size, output_size)
        self.encoders = nn.ModuleList([_Layer(output_size, attention_heads, linear_units, cnn_module_kernel)
                                       for _ in range(num_blocks)])
        self.after_norm = _LN(output_size)
        self.register_buffer("pe", _sinusoid_table(max_len, output_size), persistent=False)
        if global_cmvn is not None:
            mean, istd = global_cmvn
            # constructor data, like the reference's GlobalCMVN tensors (layers/cmvn.py:19-22): moves with .to(device) but is
            # not part of state_dict()/checkpoints
            self.register_buffer("cmvn_mean", torch.as_tensor(np.asarray(mean), dtype=torch.float32), persistent=False)
            self.register_buffer("cmvn_istd", torch.as_tensor(np.asarray(istd), dtype=torch.float32), persistent=False)
        else:
            self.cmvn_mean = self.cmvn_istd = None
        self._prepared = None
        self._pos_cache = {}
        self.fuse_min_rows = 1        # rows (B * T') below which forward() uses the general one-launch-per-cell form
        self.subsample_group = None   # utterances per conv1 -> conv2 group (None: sized for the Infinity Cache)
        self.subsample_fused = True   # conv1 inside conv2's launch where the shape is covered (set False before prepare() for the two-kernel path)
        # any load_state_dict (torch's own or utils.ckpt.load_mindspore_checkpoint) invalidates the bf16 / packed copies
        self.register_load_state_dict_post_hook(lambda module, _keys: setattr(module, "_prepared", None))

    def output_size(self):
        return self.d

    # ---- weight preparation: bf16 copies in the layouts the kernels read ------------------------------------
    @torch.no_grad()
    def prepare(self):
        """(Re)build the bf16 / folded tensors from the float32 masters. Call after loading or updating weights."""
        bf = torch.bfloat16
        e = self.embed
        c = self.d
        f2 = e.out.in_features // c
        prep = {
            "conv1_w": e.conv1.weight.detach().reshape(c, 9).contiguous().float(),
            "conv1_b": e.conv1.bias.detach().float().contiguous(),
            # (Cout, Cin, 3, 3) -> (Cout, kh, kw, Cin): k = (kh, kw, c) of the implicit GEMM
            "conv2_w": e.conv2.weight.detach().permute(0, 2, 3, 1).contiguous().to(bf),
            "conv2_b": e.conv2.bias.detach().float().contiguous(),
            # reference flattens (c, f) (subsampling.py:76); our activation is (f, c): permute the columns once
            "out_w": e.out.weight.detach().view(c, c, f2).permute(0, 2, 1).reshape(c, f2 * c).contiguous().to(bf),
            "out_b": e.out.bias.detach().float().contiguous(),
            "pos_w": torch.cat([l.self_attn.linear_pos.weight.detach() for l in self.encoders], 0).contiguous().to(bf),
            "layers": [],
        }
        for l in self.encoders:
            a, cm = l.self_attn, l.conv_module
            bn = cm.norm
            scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
            shift = bn.bias.detach() + (cm.depthwise_conv.bias.detach() - bn.running_mean) * scale
            prep["layers"].append({
                "ffm_w1": l.feed_forward_macaron.w_1.weight.detach().to(bf).contiguous(),
                "ffm_b1": l.feed_forward_macaron.w_1.bias.detach().float().contiguous(),
                "ffm_w2": l.feed_forward_macaron.w_2.weight.detach().to(bf).contiguous(),
                "ffm_b2": l.feed_forward_macaron.w_2.bias.detach().float().contiguous(),
                "ff_w1": l.feed_forward.w_1.weight.detach().to(bf).contiguous(),
                "ff_b1": l.feed_forward.w_1.bias.detach().float().contiguous(),
                "ff_w2": l.feed_forward.w_2.weight.detach().to(bf).contiguous(),
                "ff_b2": l.feed_forward.w_2.bias.detach().float().contiguous(),
                "qkv_w": torch.cat([a.linear_q.weight, a.linear_k.weight, a.linear_v.weight], 0).detach().to(bf).contiguous(),
                "qkv_b": torch.cat([a.linear_q.bias, a.linear_k.bias, a.linear_v.bias], 0).detach().float().contiguous(),
                "o_w": a.linear_out.weight.detach().to(bf).contiguous(),
                "o_b": a.linear_out.bias.detach().float().contiguous(),
                "u": a.pos_bias_u.detach().float().contiguous(),
                "v": a.pos_bias_v.detach().float().contiguous(),
                "pw1_w": cm.pointwise_conv1.weight.detach().squeeze(-1).to(bf).contiguous(),
                "pw1_b": cm.pointwise_conv1.bias.detach().float().contiguous(),
                "dw_w": cm.depthwise_conv.weight.detach().squeeze(1).float().contiguous(),
                "bn_scale": scale.float().contiguous(),
                "bn_shift": shift.float().contiguous(),
                "pw2_w": cm.pointwise_conv2.weight.detach().squeeze(-1).to(bf).contiguous(),
                "pw2_b": cm.pointwise_conv2.bias.detach().float().contiguous(),
            })
        prep["conv2_pk"] = ops.conv2d_3x3s2_pack(prep["conv2_w"])
        # conv1 + conv2 in one launch (subsample_fused.hip: conv1's output never leaves the LDS); None unless idim 80, d_model 256
        prep["sub_pk"] = ops.subsample_fused_pack(prep["conv1_w"], prep["conv2_w"], self.idim) if self.subsample_fused else None
        prep["out_pk"] = ops.gemm_rows_pack(prep["out_w"])  # None when d_model != 256
        # fragment-ordered packed copies: FFN weights for the hidden-slice-owner kernel (ops.ffn_packed), the K = 256 dense
        # layers for ops.gemm_packed
        for W in prep["layers"]:
            for key in ("qkv", "o", "pw1", "pw2"):
                W[key + "_pk"] = ops.gemm_k256_pack(W[key + "_w"]) if W[key + "_w"].shape[1] == 256 else None
            W["qkv_fpk"] = ops.ffn_qkv_pack(W["qkv_w"])  # for the linear_q/k/v tail of the FFN launch in front of the attention
            for key in ("ffm", "ff"):
                w1 = W[key + "_w1"]
                W[key + "_pk"] = ops.ffn_pack_weights(w1, W[key + "_w2"]) if w1.shape[0] % 256 == 0 and w1.shape[1] == 256 else None
        prep["fused"] = prep["conv2_pk"] is not None and self.kernel <= 15 and all(
            W[k] is not None for W in prep["layers"] for k in ("ffm_pk", "ff_pk", "qkv_fpk", "o_pk", "pw1_pk", "pw2_pk"))
        self._prepared = prep
        self._pos_cache = {}
        return self

    def _pos_projection(self, t2):
        """linear_pos(pos_emb) of every layer in one GEMM, cached per length: (t2, num_blocks * 256) bf16.
        pos_emb = rows 0..t2-1 of the absolute table (embedding.py:86-88); no dropout in eval."""
        if t2 not in self._pos_cache:
            pe = self.pe[:t2].to(torch.bfloat16).contiguous()
            self._pos_cache[t2] = ops.gemm(pe, self._prepared["pos_w"])
        return self._pos_cache[t2]

    # ---- subsampling front end --------------------------------------------------------------------------------
    def _subsample(self, xs, P):
        """CMVN + Conv2dSubsampling4's two convolutions (layers/subsampling.py:40-45): (B, T, idim) f32 -> NHWC bf16
        (B, T2, F2, C)."""
        b, t, idim = xs.shape
        if P.get("sub_pk") is not None and t >= 7:
            return ops.subsample_fused(xs, P["sub_pk"], P["conv1_b"], P["conv2_b"], self.cmvn_mean, self.cmvn_istd)
        if P["conv2_pk"] is None:
            act1 = ops.subsample_conv1(xs, P["conv1_w"], P["conv1_b"], self.cmvn_mean, self.cmvn_istd)
            return ops.conv2d_3x3s2_nhwc(act1, P["conv2_w"], P["conv2_b"], relu=True)
        # conv1's output (B x 10 MB at T = 1000) is written once and read once: run conv1 -> conv2 over groups of
        # utterances through ONE reused buffer so the intermediate lives in the 256 MB Infinity Cache instead of HBM
        # (inside the bench step: 2.465 ms grouped vs 2.496 ms ungrouped).  Group size: at most what fits ~220 MB, and at most
        # 1.5 resident rounds of conv2_packed (2 workgroups of 128 rows per CU): up to there the workgroups behind the first
        # round run one per CU, at the solo rate (tools/conv2_rounds.py: 20 utterances = 740 workgroups 101 us, 22 = 814
        # workgroups 123 us).  A short remainder joins the first group if that still fits the cache (64 -> 24 + 20 + 20:
        # 2.188 vs 2.217 ms for 22 + 22 + 20, tools/group_scan.py).  `self.subsample_group` (an int or a list) overrides it.
        t1, f1 = (t - 3) // 2 + 1, (idim - 3) // 2 + 1
        t2, f2 = (t1 - 3) // 2 + 1, (f1 - 3) // 2 + 1
        c = P["conv2_b"].numel()
        per_utt = t1 * f1 * c * 2
        group = self.subsample_group
        if group is None:
            sizes = [b]
            if b * per_utt > 240e6:
                cus = torch.cuda.get_device_properties(xs.device).multi_processor_count
                cap = max(1, min(b, int(220e6 // per_utt), int(3 * cus * 128 // (t2 * f2))))
                sizes = [cap] * (b // cap)
                rem = b - cap * (b // cap)
                if rem and sizes and rem < cap // 2 and (cap + rem) * per_utt <= 245e6:
                    sizes[0] += rem
                elif rem:
                    sizes.append(rem)
        elif isinstance(group, (list, tuple)):  # explicit group sizes (tools/group_scan.py)
            sizes = [int(g) for g in group]
            assert sum(sizes) == b and min(sizes) > 0
        else:
            sizes = [min(group, b - i) for i in range(0, b, group)]
        act2 = torch.empty((b, t2, f2, c), dtype=torch.bfloat16, device=xs.device)
        act1 = torch.empty((max(sizes), t1, f1, c), dtype=torch.bfloat16, device=xs.device)
        i = 0
        for n in sizes:
            ops.subsample_conv1(xs[i:i + n], P["conv1_w"], P["conv1_b"], self.cmvn_mean, self.cmvn_istd, out=act1[:n])
            ops.conv2d_3x3s2_packed(act1[:n], P["conv2_pk"], P["conv2_b"], relu=True, out=act2[i:i + n])
            i += n
        return act2

    # ---- the 12 blocks, fused form: 3 launches per block ------------------------------------------------------
    def _blocks_fused(self, x, P, pos_all, att_mask, mask_rows, b, t2):
        """[FFN (+ previous block's last FFN) + LayerNorms + linear_q/k/v] -> attention -> [linear_out + residual + norm_conv
        + ConvolutionModule] (models/conformer.py:100-161); needs the packed weights of prepare()."""
        n_layers = len(self.encoders)
        l0, W0 = self.encoders[0], P["layers"][0]
        x_alt = torch.empty_like(x)
        # x += 0.5 FFN_macaron(norm_ff_macaron(x)); qkv = linear_q/k/v(norm_mha(x))              :109-119
        qkv = ops.ffn_packed_qkv(None, W0["ffm_pk"], W0["ffm_b1"], W0["ffm_b2"], x, l0.norm_mha.gamma, l0.norm_mha.beta,
                                 W0["qkv_fpk"], W0["qkv_b"], ln_in=(l0.norm_ff_macaron.gamma, l0.norm_ff_macaron.beta))
        for li, (l, W) in enumerate(zip(self.encoders, P["layers"])):
            ctx = ops.relpos_attention(qkv, pos_all[:, li * self.d:(li + 1) * self.d], W["u"], W["v"], att_mask, b, t2,
                                       self.heads, self.d // self.heads)
            # x += linear_out(ctx); x += ConvModule(norm_conv(x), mask_pad)                      :121-143, convolution.py:83-129
            # (out of place, the two residual buffers alternate: a tile reads its neighbours' residual rows)
            x, x_alt = ops.attn_out_convmodule(ctx, W["o_pk"], W["o_b"], l.norm_conv.gamma, l.norm_conv.beta, W["pw1_pk"], W["pw1_b"],
                                               W["dw_w"], W["bn_scale"], W["bn_shift"], W["pw2_pk"], W["pw2_b"], mask_rows, x, b, t2,
                                               out=x_alt), x
            # x = norm_final(x + 0.5 FFN(norm_ff(x)))                                             :147-156
            if li + 1 < n_layers:  # ... and the next block's macaron FFN + norm_mha + linear_q/k/v on the same rows
                ln, Wn = self.encoders[li + 1], P["layers"][li + 1]
                ev = self.__dict__.get("_pair_events")  # (bench.py: the launch timed IN the step, one event pair per launch)
                if ev is not None:
                    ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                    ev[-1][0].record()
                qkv = ops.ffn_packed_pair(W["ff_pk"], W["ff_b1"], W["ff_b2"], Wn["ffm_pk"], Wn["ffm_b1"], Wn["ffm_b2"], x,
                                          (l.norm_ff.gamma, l.norm_ff.beta), (l.norm_final.gamma, l.norm_final.beta),
                                          (ln.norm_ff_macaron.gamma, ln.norm_ff_macaron.beta),
                                          (ln.norm_mha.gamma, ln.norm_mha.beta), qkv=(Wn["qkv_fpk"], Wn["qkv_b"]))
                if ev is not None:
                    ev[-1][1].record()
            else:                  # ... and after_norm (:253)
                x = ops.ffn_packed(None, W["ff_pk"], W["ff_b1"], W["ff_b2"], x, l.norm_final.gamma, l.norm_final.beta,
                                   self.after_norm.gamma, self.after_norm.beta, out_dtype=torch.float32,
                                   ln_in=(l.norm_ff.gamma, l.norm_ff.beta))
        return x

    # ---- the blocks, general form: one launch per reference cell ----------------------------------------------
    def _blocks_general(self, x, P, pos_all, att_mask, mask_rows, b, t2):
        f32 = torch.float32
        n_layers = len(self.encoders)

        def ffn(a, W, key):  # x += 0.5 * (w_2 swish(w_1 a))   positionwise_feed_forward.py:33-46
            h = ops.gemm(a, W[key + "_w1"], bias=W[key + "_b1"], act=_lib.ACT_SWISH)
            ops.gemm(h, W[key + "_w2"], bias=W[key + "_b2"], residual=x, alpha=0.5, out_dtype=f32, out=x)

        a = ops.layernorm(x, self.encoders[0].norm_ff_macaron.gamma, self.encoders[0].norm_ff_macaron.beta)
        for li, (l, W) in enumerate(zip(self.encoders, P["layers"])):
            ffn(a, W, "ffm")                                                                      # :109-112
            a = ops.layernorm(x, l.norm_mha.gamma, l.norm_mha.beta)                               # :117-135
            qkv = ops.gemm(a, W["qkv_w"], bias=W["qkv_b"])
            ctx = ops.relpos_attention(qkv, pos_all[:, li * self.d:(li + 1) * self.d], W["u"], W["v"], att_mask, b, t2,
                                       self.heads, self.d // self.heads)
            ops.gemm(ctx, W["o_w"], bias=W["o_b"], residual=x, out_dtype=f32, out=x)
            a = ops.layernorm(x, l.norm_conv.gamma, l.norm_conv.beta, row_scale=mask_rows)        # :139-143
            y = ops.gemm(a, W["pw1_w"], bias=W["pw1_b"])
            z = ops.convmodule_mid(y, W["dw_w"], W["bn_scale"], W["bn_shift"], b, t2)
            ops.gemm(z, W["pw2_w"], bias=W["pw2_b"], row_scale=mask_rows, residual=x, out_dtype=f32, out=x)
            ffn(ops.layernorm(x, l.norm_ff.gamma, l.norm_ff.beta), W, "ff")                       # :147-151
            # x = norm_final(x), chained with the LayerNorm that consumes it next                  :153-156, 253
            nxt = self.encoders[li + 1].norm_ff_macaron if li + 1 < n_layers else self.after_norm
            if self.d == 256:
                y = ops.layernorm2(x, l.norm_final.gamma, l.norm_final.beta, nxt.gamma, nxt.beta,
                                   out2_dtype=None if li + 1 < n_layers else f32)
            else:
                ops.layernorm(x, l.norm_final.gamma, l.norm_final.beta, out_dtype=f32, out=x)
                y = ops.layernorm(x, nxt.gamma, nxt.beta, out_dtype=None if li + 1 < n_layers else f32)
            if li + 1 < n_layers:
                a = y
            else:
                x = y
        return x

    # ---- training-mode forward: dropout + BatchNorm batch statistics (models/conformer.py:100-161 with self.training) ---------
    def _train_engine(self):
        """The forward half of mindaudio_amd.train.engine.ConformerCTCTrainStep on this module's weights: ONE implementation of the
        training-mode forward (fused feed-forward modules, dense layers with dropout + residual + LayerNorm epilogues, the BatchNorm
        statistics kernels).  Built on first use around a throw-away CTC head; the engine's flat copies of the parameters are refreshed
        whenever a parameter of this module has changed (torch's version counters), its BatchNorm running statistics ARE this
        module's buffers."""
        params = list(self.parameters())
        stamp = tuple(p._version for p in params) + tuple(p.data_ptr() for p in params)
        eng = self.__dict__.get("_train_eng")
        if eng is None:
            from ..conformer.asr_model import CTC, ASRModel
            from ..train.engine import ConformerCTCTrainStep

            holder = ASRModel(4, self, CTC(4, self.d).to(params[0].device), 1.0)
            eng = ConformerCTCTrainStep(holder, dropout_rate=self.dropout_rate, positional_dropout_rate=self.positional_dropout_rate,
                                        seed=self.seed, bn_momentum=self.encoders[0].conv_module.norm.momentum)
            eng.block_tables = False
            object.__setattr__(self, "_train_eng", eng)  # (not a sub-module: the holder refers back to this encoder)
            self.__dict__["_train_eng_stamp"] = None
        if self.__dict__.get("_train_eng_stamp") != stamp:
            eng._copy_params(to_flat=True)
            eng.refresh_weights()
            self.__dict__["_train_eng_stamp"] = stamp
        eng.bn_mean = [l.conv_module.norm.running_mean for l in self.encoders]
        eng.bn_var = [l.conv_module.norm.running_var for l in self.encoders]
        eng.p_drop, eng.p_pos = float(self.dropout_rate), float(self.positional_dropout_rate)
        return eng

    def _forward_train(self, xs, P, masks, xs_chunk_masks):
        """`module.train()(xs, masks)`: the training step's forward half (ConformerCTCTrainStep.encoder_forward_train) with this
        module's dropout seed rule; updates the BatchNorm running statistics in place like nn.BatchNorm1d.  Nothing is kept for a
        backward pass."""
        seed = (self.seed + self._train_calls) & 0x7fffffff
        self._train_calls += 1
        x = self._train_engine().encoder_forward_train(xs, masks, xs_chunk_masks, seed=seed)
        self._bn_dirty = True  # the running statistics folded into the eval path's bn_scale / bn_shift have moved
        return x, masks

    @torch.no_grad()
    def _refresh_bn(self, P):
        """Re-fold BatchNorm (running statistics) + depthwise bias into the affine form the eval kernels read."""
        for l, W in zip(self.encoders, P["layers"]):
            cm, bn = l.conv_module, l.conv_module.norm
            scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
            W["bn_scale"].copy_(scale.float())
            W["bn_shift"].copy_((bn.bias.detach() + (cm.depthwise_conv.bias.detach() - bn.running_mean) * scale).float())
        self._bn_dirty = False

    @staticmethod
    def _attention_mask(mask2d, xs_chunk_masks, b, t2, f32):
        """The mask the attention sees (models/conformer.py:251-252: `mask=xs_chunk_masks`): the (B, T') padding mask, or the
        (B, T', T') chunk mask of the streaming configuration (utils/mask.py:201-271; the padding mask is already folded in)."""
        if xs_chunk_masks is None:
            return mask2d
        cm = xs_chunk_masks
        if cm.dim() == 3 and cm.shape[1] == t2 and t2 > 1:
            if tuple(cm.shape) != (b, t2, t2):
                raise ValueError("xs_chunk_masks must be (B, 1, %d) or (B, %d, %d), got %s" % (t2, t2, t2, tuple(cm.shape)))
            return cm.to(f32).contiguous()
        return cm.reshape(b, t2).to(f32).contiguous()

    @torch.no_grad()
    def forward(self, xs, masks, xs_chunk_masks=None):
        """xs (B, T, idim) float32 on the HIP device; masks (B, 1, T') — the subsampled pad mask the collate
        function builds (dataset.py:620-632). Returns (xs (B, T', 256) float32, masks) like BaseEncoder.construct
        (models/conformer.py:229-258)."""
        if self._prepared is None:
            self.prepare()
        P = self._prepared
        f32 = torch.float32
        if not self.training and self._bn_dirty:
            self._refresh_bn(P)
        if self.training:
            if self.kernel not in (3, 7, 15, 31):
                raise NotImplementedError("the training-mode depthwise convolution is built for kernel sizes 3, 7, 15 and 31")
            return self._forward_train(xs.to(f32), P, masks, xs_chunk_masks)
        b = xs.shape[0]
        act2 = self._subsample(xs.to(f32), P)  # (any strides: conv1 reads the view as it is)
        _, t2, f2, c = act2.shape
        m = b * t2
        if masks.shape[-1] != t2:
            raise ValueError("masks must be the subsampled pad mask (B, 1, %d), got %s" % (t2, tuple(masks.shape)))
        mask2d = masks.reshape(b, t2).to(f32).contiguous()
        mask_rows = mask2d.reshape(m)
        att_mask = self._attention_mask(mask2d, xs_chunk_masks, b, t2, f32)
        # Dense(4864 -> 256) then x * sqrt(d) (subsampling.py:76, embedding.py:84)
        if P.get("out_pk") is not None:
            x = ops.gemm_rows_packed(act2.view(m, f2 * c), P["out_pk"], P["out_b"], alpha=math.sqrt(self.d))
        else:
            x = ops.gemm(act2.view(m, f2 * c), P["out_w"], bias=P["out_b"], alpha=math.sqrt(self.d), out_dtype=f32)
        pos_all = self._pos_projection(t2)
        # The fused launches win at every batch size measured (B = 1 .. 64 at T = 1000: 1.44 vs 2.25 ms at B = 1); the
        # general form covers shapes the packed kernels do not (and `fuse_min_rows` lets the tests run it on any shape).
        blocks = self._blocks_fused if P["fused"] and m >= self.fuse_min_rows else self._blocks_general
        x = blocks(x, P, pos_all, att_mask, mask_rows, b, t2)
        return x.view(b, t2, self.d), masks
