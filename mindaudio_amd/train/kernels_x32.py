"""float32 validation mode of the training step: the same function names as `train/kernels.py` and the handful of `ops`
the engine calls, over the `_x32` C-ABI entry points (include/mindaudio_amd.h, "float32 validation mode").  Every activation
tensor is float32; `compute_type=float32` is the reference's own default (mindaudio/models/conformer.py:61,
examples/conformer/asr_model.py:307-310).  Validation shapes only: plain FMA kernels, explicit im2col."""
import ctypes

from .. import _host, _lib
from .. import ops as _ops
from . import kernels as _K

# pieces that are float32 on both paths
pad64 = _K.pad64
layernorm_bwd = _K.layernorm_bwd        # takes float32 or bf16 dy
dropout_add = _K.dropout_add            # takes float32 or bf16 y
grad_overflow = _K.grad_overflow
adam = _K.adam
_reduce_ws = _K._reduce_ws
embed_posenc = _K.embed_posenc          # float32 in, float32 out
embed_bwd = _K.embed_bwd
transpose = _K.transpose


def _t():
    return _host.torch()


def _s():
    return _host.current_stream_ptr()


def _p(x):
    return _host.ptr(x) if x is not None else None


def _f32(*xs):
    t = _t()
    for x in xs:
        assert x is None or x.dtype == t.float32, "float32 validation mode takes float32 tensors"


def _gemm(a, a_rs, a_cs, b, b_ns, b_ks, out, m, n, k, bias=None, residual=None, row_scale=None, alpha=1.0, act=_lib.ACT_NONE):
    e = _lib.GemmEpilogue()
    e.bias = bias.data_ptr() if bias is not None else None
    e.residual = residual.data_ptr() if residual is not None else None
    e.row_scale = row_scale.data_ptr() if row_scale is not None else None
    e.ldr = residual.stride(0) if residual is not None else 0
    e.alpha, e.act, e.out_bf16 = float(alpha), int(act), 0
    _lib.check(_lib.load().ma_gemm_x32(_p(a), a_rs, a_cs, _p(b), b_ns, b_ks, _p(out), out.stride(0), m, n, k, ctypes.byref(e),
                                       _s()), "gemm_x32")
    return out


# ---- the `ops` the engine uses ----------------------------------------------------------------------------------------
def gemm(a, w, bias=None, residual=None, row_scale=None, alpha=1.0, act=_lib.ACT_NONE, out_dtype=None, out=None):
    """out (M, N) float32 = act(a (M, K) @ w (N, K)^T + bias) * alpha * row_scale[:, None] (+ residual)."""
    t = _t()
    _f32(a, w, bias, residual, row_scale)
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k and a.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = t.empty((m, n), dtype=t.float32, device=a.device)
    assert out.dtype == t.float32 and tuple(out.shape) == (m, n) and out.stride(1) == 1
    return _gemm(a, a.stride(0), 1, w, w.stride(0), 1, out, m, n, k, bias, residual, row_scale, alpha, act)


def gemm_nn(a, w, residual=None, out=None):
    """out (M, K) float32 = a (M, N) @ w (N, K) (+ residual): dX = dY . W with W in the forward's (out, in) layout."""
    t = _t()
    _f32(a, w, residual)
    m, n = a.shape
    k = w.shape[1]
    assert w.shape[0] == n and a.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = t.empty((m, k), dtype=t.float32, device=a.device)
    return _gemm(a, a.stride(0), 1, w, 1, w.stride(0), out, m, k, n, residual=residual)


def cast_bf16(x):
    return x  # nothing is rounded in this mode


def layernorm(x, gamma, beta, eps=1e-5, row_scale=None, out_dtype=None, out=None):
    return _ops.layernorm(x, gamma, beta, eps=eps, row_scale=row_scale, out_dtype=_t().float32, out=out)


def subsample_conv1(x, w, bias, cmvn_mean=None, cmvn_istd=None, out=None):
    """x (B, T, idim) float32 -> NHWC float32 (B, T1, F1, C) after CMVN, conv(1 -> C, 3x3, s2), ReLU (subsampling.py:40-45)."""
    t = _t()
    b, tt, idim = x.shape
    c = w.shape[0]
    out = t.empty((b, (tt - 3) // 2 + 1, (idim - 3) // 2 + 1, c), dtype=t.float32, device=x.device)
    _lib.check(_lib.load().ma_subsample_conv1_nhwc_x32(_p(x), x.stride(0), x.stride(1), x.stride(2), b, tt, idim,
                                                       _p(cmvn_mean), _p(cmvn_istd), _p(w), _p(bias), c, _p(out), _s()),
               "subsample_conv1_x32")
    return out


def _im2col(act):
    t = _t()
    b, h, wd, c = act.shape
    m = b * ((h - 3) // 2 + 1) * ((wd - 3) // 2 + 1)
    col = t.empty((m, 9 * c), dtype=t.float32, device=act.device)
    _lib.check(_lib.load().ma_im2col_3x3s2_nhwc_x32(_p(act), b, h, wd, c, _p(col), _s()), "im2col_x32")
    return col


def conv2d_3x3s2_nhwc(act, w, bias=None, relu=True, out_dtype=None):
    """act (B, H, W, C) float32 NHWC, w (Cout, 3, 3, C) float32 -> (B, Ho, Wo, Cout) float32."""
    b, h, wd, c = act.shape
    cout = w.shape[0]
    out = gemm(_im2col(act), w.reshape(cout, 9 * c), bias=bias, act=_lib.ACT_RELU if relu else _lib.ACT_NONE)
    return out.view(b, (h - 3) // 2 + 1, (wd - 3) // 2 + 1, cout)


# ---- the training kernels ---------------------------------------------------------------------------------------------
def gemm_tn(a, b, out, colsum=None, rows_store=None, alpha=1.0, accumulate=True):
    """out (rows_store, No) float32 (+)= alpha * a^T @ b for row-major a (Kc, Mo), b (Kc, No); colsum (+)= column sums of a."""
    _f32(a, b, out, colsum)
    kc, mo = a.shape
    no = b.shape[1]
    rows_store = mo if rows_store is None else rows_store
    assert b.shape[0] == kc and tuple(out.shape) == (rows_store, no)
    _gemm(a, 1, a.stride(0), b, 1, b.stride(0), out, rows_store, no, kc, residual=out if accumulate else None, alpha=alpha)
    if colsum is not None:
        _lib.check(_lib.load().ma_colsum_x32(_p(a), a.stride(0), kc, min(mo, colsum.numel()), _p(colsum), 1, _s()), "colsum_x32")
    return out


def conv2d_dw(dy, act, dw, dbias):
    """dw (Cout, 9C) += dy^T @ im2col(act); dbias += column sums of dy."""
    gemm_tn(dy, _im2col(act), dw, colsum=dbias)


def act_dropout_fwd(u, p, seed, salt, act=_lib.ACT_SWISH):
    _f32(u)
    h = _t().empty_like(u)
    _lib.check(_lib.load().ma_act_dropout_fwd_x32(_p(u), _p(h), u.numel(), act, float(p), seed, salt, _s()), "act_dropout_x32")
    return h


def act_dropout_bwd(u, dh, p, seed, salt, out=None, act=_lib.ACT_SWISH):
    _f32(u, dh)
    du = out if out is not None else _t().empty_like(u)
    _lib.check(_lib.load().ma_act_dropout_bwd_x32(_p(u), _p(dh), _p(du), u.numel(), act, float(p), seed, salt, _s()),
               "act_dropout_bwd_x32")
    return du


def dropout_bwd(g, alpha, p, seed, salt, row_scale=None):
    t = _t()
    dy = t.empty(g.shape, dtype=t.float32, device=g.device)
    _lib.check(_lib.load().ma_dropout_bwd_x32(_p(g), g.stride(0), _p(dy), dy.stride(0), g.shape[0], g.shape[1], float(alpha),
                                              _p(row_scale), float(p), seed, salt, _s()), "dropout_bwd_x32")
    return dy


def convmid_fwd_train(y, batch, T, dw_w, dw_b, gamma, beta, run_mean, run_var, eps=1e-5, momentum=0.1):
    t = _t()
    lib = _lib.load()
    _f32(y)
    c, ks = dw_w.shape
    rows = batch * T
    z = t.empty((rows, c), dtype=t.float32, device=y.device)
    nparts = int(lib.ma_convmid_fwd_train_parts(batch, T, c))
    _lib.check(min(nparts, 0), "convmid_fwd_train")
    sums = t.empty(nparts * 2 * c, dtype=t.float32, device=y.device)  # per-workgroup partial (sum | sum of squares) vectors
    stats = t.empty(2 * c, dtype=t.float32, device=y.device)
    out = t.empty((rows, c), dtype=t.float32, device=y.device)
    _lib.check(lib.ma_convmid_fwd_train_x32(_p(y), y.stride(0), batch, T, c, _p(dw_w), ks, _p(dw_b), _p(z), _p(sums), _s()),
               "convmid_fwd_train_x32")
    _lib.check(lib.ma_bn_finalize_f32(_p(sums), nparts, c, rows, float(eps), float(momentum), _p(run_mean), _p(run_var),
                                      _p(stats), _s()), "bn_finalize")
    _lib.check(lib.ma_bn_swish_fwd_x32(_p(z), _p(stats), _p(gamma), _p(beta), _p(out), rows, c, _s()), "bn_swish_fwd_x32")
    return out, z, stats


def convmid_bwd(dout, y, z, stats, batch, T, dw_w, gamma, beta, d_dw_w, d_dw_b, d_gamma, d_beta):
    t = _t()
    lib = _lib.load()
    _f32(dout, y)
    c, ks = dw_w.shape
    rows = batch * T
    dz = t.empty((rows, c), dtype=t.float32, device=y.device)
    dsum = t.empty(2 * c, dtype=t.float32, device=y.device)
    rw = _K._reduce_ws(y.device)
    _lib.check(lib.ma_bn_swish_bwd_x32(_p(dout), _p(z), _p(stats), _p(gamma), _p(beta), _p(dz), rows, c, _p(dsum), _p(d_gamma), _p(d_beta),
               _p(rw), rw.numel(), _s()), "bn_swish_bwd_x32")
    dy = t.empty((rows, 2 * c), dtype=t.float32, device=y.device)
    rw = _reduce_ws(y.device)
    _lib.check(lib.ma_convmid_bwd_x32(_p(dz), _p(y), y.stride(0), batch, T, c, _p(dw_w), ks, _p(dy), dy.stride(0), _p(d_dw_w),
                                      _p(d_dw_b), _p(rw), rw.numel(), _s()), "convmid_bwd_x32")
    return dy


def relu_bwd(dy, y):
    _f32(dy, y)
    _lib.check(_lib.load().ma_relu_bwd_x32(_p(dy), _p(y), dy.numel(), _s()), "relu_bwd_x32")
    return dy


def col2im_relu(dcol, act):
    _f32(dcol, act)
    dact = _t().empty_like(act)
    b, h, w, c = act.shape
    _lib.check(_lib.load().ma_col2im_3x3s2_relu_x32(_p(dcol), _p(act), b, h, w, c, _p(dact), _s()), "col2im_x32")
    return dact


def conv1_dw(dact, x, cmvn_mean, cmvn_istd, dw, db):
    _f32(dact)
    b, tt, idim = x.shape
    rw = _reduce_ws(x.device)
    _lib.check(_lib.load().ma_subsample_conv1_dw_x32(_p(dact), _p(x), b, tt, idim, _p(cmvn_mean), _p(cmvn_istd),
                                                     dact.shape[-1], _p(dw), _p(db), _p(rw), rw.numel(), _s()), "conv1_dw_x32")


def attention_fwd(qkv, pos, bias_u, bias_v, mask, batch, T, heads=4, d_k=64):
    """(ctx (B*T, H*64) float32, lse (B, H, T) float32)."""
    t = _t()
    _f32(qkv, pos)
    ctx = t.empty((batch * T, heads * d_k), dtype=t.float32, device=qkv.device)
    lse = t.empty((batch, heads, T), dtype=t.float32, device=qkv.device)
    lib = _lib.load()
    fn = lib.ma_relpos_attention_fwd_x32
    if mask is not None and mask.dim() == 3:  # (B, T, T) chunk masks
        assert tuple(mask.shape) == (batch, T, T) and mask.dtype == t.float32 and mask.is_contiguous()
        fn = lib.ma_relpos_attention_fwd_qmask_x32
    _lib.check(fn(_p(qkv), qkv.stride(0), _p(pos), pos.stride(0), _p(bias_u), _p(bias_v), _p(mask), batch, T, heads, d_k,
                  _p(ctx), ctx.stride(0), _p(lse), _s()), "attention_fwd_x32")
    return ctx, lse


def attention_bwd(qkv, pos, bias_u, bias_v, mask, ctx, dctx, lse, batch, T, dpos, dbias_u, dbias_v, heads=4, d_k=64):
    t = _t()
    lib = _lib.load()
    _f32(qkv, pos, ctx, dctx)
    dqkv = t.empty((batch * T, 3 * heads * d_k), dtype=t.float32, device=qkv.device)
    ws_bytes = lib.ma_relpos_attention_bwd_x32_workspace_bytes(batch, T, heads)
    ws = t.empty(ws_bytes, dtype=t.uint8, device=qkv.device)
    fn = lib.ma_relpos_attention_bwd_x32
    if mask is not None and mask.dim() == 3:
        assert tuple(mask.shape) == (batch, T, T) and mask.dtype == t.float32 and mask.is_contiguous()
        fn = lib.ma_relpos_attention_bwd_qmask_x32
    _lib.check(fn(_p(qkv), qkv.stride(0), _p(pos), pos.stride(0), _p(bias_u), _p(bias_v), _p(mask), _p(ctx), ctx.stride(0),
                  _p(dctx), dctx.stride(0), _p(lse), batch, T, heads, d_k, _p(dqkv), dqkv.stride(0), _p(dpos), dpos.stride(0),
                  _p(dbias_u), _p(dbias_v), _p(ws), ws_bytes, _s()), "attention_bwd_x32")
    return dqkv


def ctc_loss_grad(logits, V, batch, T, ys_pad, hlens, ys_lens, grad_scale, blank=0):
    """logits (B*T, ld >= V) float32 -> (loss, per-utterance nll, dlogits (B*T, ld) float32 scaled by grad_scale)."""
    t = _t()
    lib = _lib.load()
    dev = logits.device
    ys_pad = ys_pad.to(t.int32).contiguous()
    hlens = hlens.to(t.int32).contiguous()
    ys_lens = ys_lens.to(t.int32).contiguous()
    per = t.empty(batch, dtype=t.float32, device=dev)
    lse = t.empty(batch * T, dtype=t.float32, device=dev)
    out = t.empty(1, dtype=t.float32, device=dev)
    dlog = t.empty((batch * T, logits.stride(0)), dtype=t.float32, device=dev)
    ws_bytes = lib.ma_ctc_grad_workspace_bytes(batch, T, ys_pad.shape[1])
    ws = t.empty(ws_bytes, dtype=t.uint8, device=dev)
    _lib.check(lib.ma_ctc_loss_grad_x32(_p(logits), logits.stride(0), batch, T, V, _p(ys_pad), ys_pad.shape[1], _p(hlens),
                                        _p(ys_lens), blank, 1, float(grad_scale), _p(per), _p(lse), _p(out), _p(dlog),
                                        dlog.stride(0), _p(ws), ws_bytes, _s()), "ctc_loss_grad_x32")
    return out[0], per, dlog


# ---- attention-decoder branch of the hybrid loss on float32 activations ----------------------------------------------------------
def mha_small_fwd(q, k, v, mask, mask_mode, batch, lq, lk, scale, heads=4, d_k=64):
    t = _t()
    _f32(q, k, v)
    ctx = t.empty((batch * lq, heads * d_k), dtype=t.float32, device=q.device)
    probs = t.empty((batch, heads, lq, lk), dtype=t.float32, device=q.device)
    _lib.check(_lib.load().ma_mha_small_fwd_x32(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(mask), mask_mode, batch,
                                                lq, lk, heads, d_k, float(scale), _p(ctx), ctx.stride(0), _p(probs), _s()),
               "mha_small_fwd_x32")
    return ctx, probs


def mha_small_bwd(q, k, v, probs, ctx, dctx, batch, lq, lk, scale, dq, dk, dv, heads=4, d_k=64):
    _f32(q, k, v, ctx, dctx, dq, dk, dv)
    _lib.check(_lib.load().ma_mha_small_bwd_x32(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(probs), _p(ctx),
                                                ctx.stride(0), _p(dctx), dctx.stride(0), batch, lq, lk, heads, d_k, float(scale),
                                                _p(dq), dq.stride(0), _p(dk), dk.stride(0), _p(dv), dv.stride(0), _s()),
               "mha_small_bwd_x32")


def label_smoothing_loss_grad(logits, V, target, mask, smoothing, grad_scale, normalize_length=False):
    return _K._label_smoothing(logits, V, target, mask, smoothing, grad_scale, normalize_length, True)
