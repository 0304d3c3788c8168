"""EcapaTDNN on MI355X — mirror of mindaudio.models.ecapatdnn.EcapaTDNN (ecapatdnn.py:310-432), forward (eval mode:
BatchNorm running statistics) in hand-written HIP kernels through the C-ABI.

Every convolution is an implicit bf16 MFMA GEMM over (B, T + 2*HALO, C) activations with zero halo frames
(ma_conv1d_taps_bf16 / ma_gemm_bf16) whose epilogue applies bias -> ReLU -> BatchNorm (affine) [-> tanh] and re-zeroes
the halo; the SE squeeze/excite, Res2Net adds and attentive statistics pooling are small bandwidth kernels.
PyTorch layers are parameter containers only (their forward() is never called)."""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from .. import _host, _lib, ops

HALO = 4  # >= max dilation * (kernel - 1) / 2 of the shipped configuration (dilations up to 4, kernel 3; kernel 5 x 1)


class _TDNN(nn.Module):
    def __init__(self, cin, cout, k, d):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, k, dilation=d, padding=d * (k - 1) // 2)
        self.norm = nn.BatchNorm1d(cout, eps=1e-5)


class _Res2Net(nn.Module):
    def __init__(self, c, scale, k, d):
        super().__init__()
        self.blocks = nn.ModuleList([_TDNN(c // scale, c // scale, k, d) for _ in range(scale - 1)])


class _SE(nn.Module):
    def __init__(self, c, se):
        super().__init__()
        self.conv1 = nn.Conv1d(c, se, 1)
        self.conv2 = nn.Conv1d(se, c, 1)


class _SERes2Net(nn.Module):
    def __init__(self, c, scale, se, k, d):
        super().__init__()
        self.tdnn1 = _TDNN(c, c, 1, 1)
        self.res2net_block = _Res2Net(c, scale, k, d)
        self.tdnn2 = _TDNN(c, c, 1, 1)
        self.se_block = _SE(c, se)


class _ASP(nn.Module):
    def __init__(self, c, att):
        super().__init__()
        self.tdnn = _TDNN(c, att, 1, 1)
        self.conv = nn.Conv1d(att, c, 1)


def _conv(x, lda, rows, cin, w, taps, dil, out, ldo, n, bias, act, bn=None, act2=0, row_scale=None, out_f32=False):
    e = _lib.GemmEpilogue()
    e.bias = bias.data_ptr() if bias is not None else None
    e.row_scale = row_scale.data_ptr() if row_scale is not None else None
    e.alpha, e.act, e.act2 = 1.0, act, act2
    e.out_bf16 = 0 if out_f32 else 1
    if bn is not None:
        e.col_scale, e.col_shift = bn[0].data_ptr(), bn[1].data_ptr()
    lib = _lib.load()
    if taps == 1:
        rc = lib.ma_gemm_bf16(x, lda, w.data_ptr(), w.stride(0), out, ldo, rows, n, cin, ctypes.byref(e),
                              _host.current_stream_ptr())
    else:
        rc = lib.ma_conv1d_taps_bf16(x, lda, rows, cin, taps, dil, w.data_ptr(), out, ldo, n, ctypes.byref(e),
                                     _host.current_stream_ptr())
    _lib.check(rc, "ecapa conv")


class EcapaTDNN(nn.Module):
    """Constructor arguments of the reference (ecapatdnn.py:336-349).  `activation` must be ReLU, `groups` all 1 and
    `global_context` False — what both shipped configurations use (train_speaker_embeddings.py:468-472)."""

    def __init__(self, input_size, lin_neurons=192, activation=None, channels=(512, 512, 512, 512, 1536),
                 kernel_sizes=(5, 3, 3, 3, 1), dilations=(1, 2, 3, 4, 1), attention_channels=128, res2net_scale=8,
                 se_channels=128, global_context=False, groups=(1, 1, 1, 1, 1)):
        super().__init__()
        c = channels[0]
        if global_context or any(g != 1 for g in groups) or activation not in (None, nn.ReLU):
            raise NotImplementedError("global_context / grouped convolutions / non-ReLU activations are not built")
        if any(ch != c for ch in channels[:-1]) or channels[-1] != 3 * c or len(channels) != 5:
            raise NotImplementedError("built for channels (C, C, C, C, 3C)")
        if c % (64 * res2net_scale) or attention_channels % 64 or se_channels % 64 or kernel_sizes[-1] != 1:
            raise NotImplementedError("channel counts must keep every GEMM K a multiple of 64")
        if max(d * (k - 1) // 2 for k, d in zip(kernel_sizes, dilations)) > HALO:
            raise NotImplementedError("dilation * (kernel - 1) / 2 must not exceed the %d-frame halo" % HALO)
        self.c, self.scale, self.input_size, self.lin = c, res2net_scale, input_size, lin_neurons
        self.att, self.se = attention_channels, se_channels
        self.kernel_sizes, self.dilations = tuple(kernel_sizes), tuple(dilations)
        self.blocks = nn.ModuleList([_TDNN(input_size, c, kernel_sizes[0], dilations[0])])
        for i in range(1, 4):
            self.blocks.append(_SERes2Net(c, res2net_scale, se_channels, kernel_sizes[i], dilations[i]))
        self.mfa = _TDNN(3 * c, 3 * c, 1, 1)
        self.asp = _ASP(3 * c, attention_channels)
        self.asp_bn = nn.BatchNorm1d(6 * c, eps=1e-5)
        self.fc = nn.Conv1d(6 * c, lin_neurons, 1)
        self._prepared = None
        self._ws = {}
        # any load_state_dict invalidates the bf16 / packed copies of prepare() (as the Conformer encoder, decoder and CTC head do)
        self.register_load_state_dict_post_hook(lambda module, _keys: setattr(module, "_prepared", None))
        self.fuse_res2net = True  # False: one launch per convolution / add of the Res2Net chain (the tests run both)
        self.fuse_se_block = True  # False: squeeze, excitation and scale + residual as three launches (the tests run both)
        self.fuse_se = True       # False: the SE excitation and the embedding Linear as ma_gemm_bf16 launches (the tests run both)
        self.fuse_asp = True      # False: the ASP logits as a GEMM launch + the pooling launch (the tests run both)

    @torch.no_grad()
    def prepare(self):
        bf = torch.bfloat16
        self._ws = {}  # activation buffers of earlier forwards are dropped with the old weights

        def tdnn(m, cin_pad=None):
            w = m.conv.weight.detach()  # (Cout, Cin, k) -> (Cout, k, Cin[pad]): K index = tap * Cin + c
            cout, cin, k = w.shape
            wp = w.permute(0, 2, 1)
            if cin_pad and cin_pad != cin:
                wp = torch.cat([wp, wp.new_zeros(cout, k, cin_pad - cin)], 2)
            scale = m.norm.weight.detach() / torch.sqrt(m.norm.running_var + m.norm.eps)
            shift = m.norm.bias.detach() - m.norm.running_mean * scale
            return dict(w=wp.contiguous().to(bf).view(cout, -1), b=m.conv.bias.detach().float().contiguous(),
                        bn=(scale.float().contiguous(), shift.float().contiguous()))

        def lin(m):
            return dict(w=m.weight.detach().squeeze(-1).to(bf).contiguous(), b=m.bias.detach().float().contiguous())

        self.fpad = (self.input_size + 63) // 64 * 64
        P = {"l0": tdnn(self.blocks[0], self.fpad), "blocks": []}
        for blk in self.blocks[1:]:
            r = [tdnn(b) for b in blk.res2net_block.blocks]
            P["blocks"].append(dict(t1=tdnn(blk.tdnn1), r=r, t2=tdnn(blk.tdnn2), se1=lin(blk.se_block.conv1),
                                    se2=lin(blk.se_block.conv2),
                                    # the whole chain for the one-launch form (ma_res2net_fused_bf16)
                                    r_w=torch.stack([q["w"] for q in r]).contiguous(), r_b=torch.stack([q["b"] for q in r]).contiguous(),
                                    r_s=torch.stack([q["bn"][0] for q in r]).contiguous(),
                                    r_t=torch.stack([q["bn"][1] for q in r]).contiguous()))
        P["mfa"] = tdnn(self.mfa)
        P["asp_t"] = tdnn(self.asp.tdnn)
        P["asp_c"] = lin(self.asp.conv)
        scale = self.asp_bn.weight.detach() / torch.sqrt(self.asp_bn.running_var + self.asp_bn.eps)
        P["asp_bn"] = (scale.float().contiguous(), (self.asp_bn.bias.detach() - self.asp_bn.running_mean * scale).float().contiguous())
        P["fc"] = lin(self.fc)
        self._prepared = P
        return self

    @torch.no_grad()
    def forward(self, x, lengths=None):
        """x (B, T, input_size) float32 on the HIP device -> (B, lin_neurons) float32 (`lengths` is accepted and unused,
        as in the reference, ecapatdnn.py:152-156, 411)."""
        if self.training:
            raise NotImplementedError("ECAPA training (BatchNorm batch statistics, backward) is not built; call .eval()")
        if self._prepared is None:
            self.prepare()
        P, lib = self._prepared, _lib.load()
        t = torch
        b, T, f = x.shape
        dev, c, H = x.device, self.c, HALO
        tp = T + 2 * H
        rows = b * tp
        bf = t.bfloat16
        s = _host.current_stream_ptr()

        # Activation buffers and the row mask are kept per (batch, frames, device): the margins above the first and below the last
        # utterance are read by the taps and never written, so they are zeroed once (26 fill launches per forward otherwise:
        # 2.52 -> 2.47 ms for the C = 512 forward, same box).  Calls on one stream reuse them in order; the returned embedding is a fresh tensor.
        # (the stream is part of the key: buffers shared by forwards on two streams would race)
        key = (b, T, str(dev), bool(self.fuse_res2net), int(t.cuda.current_stream().cuda_stream))
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) >= 4:
                self._ws.clear()
            rs0 = t.zeros((b, tp), dtype=t.float32, device=dev)
            rs0[:, H:H + T] = 1.0
            ws = self._ws[key] = {"rs": rs0.view(-1), "bufs": []}
        nbuf = [0]

        def buf(cols):  # (rows + 2H margin, cols) bf16; the view starts at row H (first utterance's first halo row)
            i = nbuf[0]
            nbuf[0] += 1
            if i == len(ws["bufs"]):
                full = t.empty((rows + 2 * H, cols), dtype=bf, device=dev)
                full[:H].zero_()
                full[-H:].zero_()
                ws["bufs"].append(full)
            full = ws["bufs"][i]
            assert full.shape[1] == cols
            return full, full[H:H + rows]

        rs = ws["rs"]
        xin_full, xin = buf(self.fpad)
        _lib.check(lib.ma_ecapa_pack_input_bf16(x.float().contiguous().data_ptr(), b, T, f, H, self.fpad, xin.data_ptr(), s),
                   "pack_input")
        RELU = _lib.ACT_RELU
        # blocks[0]: TDNN(input -> C, k=5)
        _, x0 = buf(c)
        L0 = P["l0"]
        _conv(xin.data_ptr(), self.fpad, rows, self.fpad, L0["w"], self.kernel_sizes[0], self.dilations[0], x0.data_ptr(),
              c, c, L0["b"], RELU, L0["bn"], row_scale=rs)
        _, cat = buf(3 * c)
        cur, cur_ld = x0, c
        cc = c // self.scale
        for bi, B_ in enumerate(P["blocks"]):
            k, d = self.kernel_sizes[bi + 1], self.dilations[bi + 1]
            _, t1 = buf(c)
            _conv(cur.data_ptr(), cur_ld, rows, c, B_["t1"]["w"], 1, 1, t1.data_ptr(), c, c, B_["t1"]["b"], RELU,
                  B_["t1"]["bn"], row_scale=rs)
            _, y = buf(c)
            fused = (self.fuse_res2net and k == 3 and self.scale == 8 and cc in (64, 128) and tp <= 384 and d <= H
                     and 0 < lib.ma_res2net_fused_lds_bytes(cc, tp, d) <= 160 * 1024)
            if fused:  # the 7 dilated convolutions + adds of the block in one launch, one utterance per workgroup
                _lib.check(lib.ma_res2net_fused_bf16(t1.data_ptr(), c, y.data_ptr(), c, b, T, H, cc, self.scale, d,
                                                     B_["r_w"].data_ptr(), B_["r_b"].data_ptr(), B_["r_s"].data_ptr(),
                                                     B_["r_t"].data_ptr(), s), "res2net_fused")
            else:
                _, tmp = buf(cc)
                _lib.check(lib.ma_add_bf16(t1.data_ptr(), c, None, 0, y.data_ptr(), c, rows, cc, s), "res2net copy")
            for i in range(1, self.scale if not fused else 1):
                R = B_["r"][i - 1]
                if i == 1:
                    src, ld = t1[:, cc:2 * cc], c
                else:
                    _lib.check(lib.ma_add_bf16(t1[:, i * cc:].data_ptr(), c, y[:, (i - 1) * cc:].data_ptr(), c,
                                               tmp.data_ptr(), cc, rows, cc, s), "res2net add")
                    src, ld = tmp, cc
                _conv(src.data_ptr(), ld, rows, cc, R["w"], k, d, y[:, i * cc:].data_ptr(), c, cc, R["b"], RELU, R["bn"],
                      row_scale=rs)
            _, t2 = buf(c)
            _conv(y.data_ptr(), c, rows, c, B_["t2"]["w"], 1, 1, t2.data_ptr(), c, c, B_["t2"]["b"], RELU, B_["t2"]["bn"],
                  row_scale=rs)
            out = cat[:, bi * c:]
            rc = _lib.MA_ERR_UNSUPPORTED
            if self.fuse_se and self.fuse_se_block:  # squeeze + excitation + scale + residual in one launch (C = 512 / 1024)
                rc = lib.ma_se_block_bf16(t2.data_ptr(), c, B_["se1"]["w"].data_ptr(), B_["se1"]["b"].data_ptr(), B_["se2"]["w"].data_ptr(),
                                          B_["se2"]["b"].data_ptr(), cur.data_ptr(), cur_ld, out.data_ptr(), 3 * c, b, T, H, c,
                                          B_["se1"]["w"].shape[0], s)
                if rc != _lib.MA_ERR_UNSUPPORTED:
                    _lib.check(rc, "se_block")
            if rc == _lib.MA_ERR_UNSUPPORTED:
                mean = t.empty((b, c), dtype=bf, device=dev)
                _lib.check(lib.ma_time_mean_bf16(t2.data_ptr(), c, b, T, H, c, mean.data_ptr(), s), "time_mean")
                g2 = t.empty((b, c), dtype=bf, device=dev)
                if self.fuse_se:  # both 1 x 1 convolutions of the excitation in one launch
                    rc = lib.ma_se_gate_bf16(mean.data_ptr(), B_["se1"]["w"].data_ptr(), B_["se1"]["b"].data_ptr(),
                                             B_["se2"]["w"].data_ptr(), B_["se2"]["b"].data_ptr(), g2.data_ptr(), b, c,
                                             B_["se1"]["w"].shape[0], s)
                    if rc != _lib.MA_ERR_UNSUPPORTED:
                        _lib.check(rc, "se_gate")
                if rc == _lib.MA_ERR_UNSUPPORTED:
                    g1 = ops.gemm(mean, B_["se1"]["w"], bias=B_["se1"]["b"], act=RELU)
                    ops.gemm(g1, B_["se2"]["w"], bias=B_["se2"]["b"], act=_lib.ACT_SIGMOID, out=g2)
                _lib.check(lib.ma_se_apply_bf16(t2.data_ptr(), c, g2.data_ptr(), cur.data_ptr(), cur_ld, out.data_ptr(), 3 * c,
                                                b, T, H, c, s), "se_apply")
            cur, cur_ld = out, 3 * c
        _, xm = buf(3 * c)
        M = P["mfa"]
        _conv(cat.data_ptr(), 3 * c, rows, 3 * c, M["w"], 1, 1, xm.data_ptr(), 3 * c, 3 * c, M["b"], RELU, M["bn"],
              row_scale=rs)
        A = P["asp_t"]
        a1 = t.empty((rows, self.att), dtype=bf, device=dev)
        _conv(xm.data_ptr(), 3 * c, rows, 3 * c, A["w"], 1, 1, a1.data_ptr(), self.att, self.att, A["b"], RELU, A["bn"],
              act2=_lib.ACT_TANH)
        pooled = t.empty((b, 6 * c), dtype=bf, device=dev)
        rc = _lib.MA_ERR_UNSUPPORTED
        if self.fuse_asp:  # logits GEMM + softmax pooling in one launch (att == 128, 3C % 256 == 0), else the two launches
            rc = lib.ma_asp_fused_bf16(a1.data_ptr(), self.att, P["asp_c"]["w"].data_ptr(), xm.data_ptr(),
                                       3 * c, b, T, H, 3 * c, self.att, 1e-12, P["asp_bn"][0].data_ptr(), P["asp_bn"][1].data_ptr(),
                                       pooled.data_ptr(), s)
            if rc != _lib.MA_ERR_UNSUPPORTED:
                _lib.check(rc, "asp_fused")
        if rc == _lib.MA_ERR_UNSUPPORTED:
            logits = ops.gemm(a1, P["asp_c"]["w"], bias=P["asp_c"]["b"])
            _lib.check(lib.ma_asp_pool_bf16(logits.data_ptr(), 3 * c, xm.data_ptr(), 3 * c, b, T, H, 3 * c, 1e-12,
                                            P["asp_bn"][0].data_ptr(), P["asp_bn"][1].data_ptr(), pooled.data_ptr(), s), "asp_pool")
        emb = t.empty((b, P["fc"]["w"].shape[0]), dtype=t.float32, device=dev)
        rc = lib.ma_linear_small_bf16(pooled.data_ptr(), 6 * c, P["fc"]["w"].data_ptr(), P["fc"]["w"].stride(0), P["fc"]["b"].data_ptr(),
                                      emb.data_ptr(), emb.stride(0), b, emb.shape[1], 6 * c, s) if self.fuse_se else _lib.MA_ERR_UNSUPPORTED
        if rc == _lib.MA_ERR_UNSUPPORTED:  # (batch not a multiple of 16, lin_neurons not of 64)
            return ops.gemm(pooled, P["fc"]["w"], bias=P["fc"]["b"], out_dtype=t.float32, out=emb)
        _lib.check(rc, "linear_small")
        return emb
