"""Python wrappers over the training C-ABI entry points (include/mindaudio_amd.h, "training step" section).
Tensors are torch HIP tensors used as device buffers; all arithmetic happens in the HIP kernels."""
from .. import _host, _lib


def _t():
    return _host.torch()  # (a block table being filled gets a proxy that keeps what the wrappers allocate: train/block_table.py)


def _s():
    return _host.current_stream_ptr()


def _p(x):
    return _host.ptr(x) if x is not None else None


_red_ws = {}


def _reduce_ws(device):
    """Scratch of the two-stage parameter-gradient reductions (own buffer: it must not alias the GEMM workspace)."""
    key = str(device)
    if key not in _red_ws:
        _red_ws[key] = _t().empty(_lib.load().ma_train_reduce_workspace_bytes(), dtype=_t().uint8, device=device)
    return _red_ws[key]


def pad64(n):
    return (n + 63) // 64 * 64


_tr_pool = {}


def transpose(x, out=None, colsum=None, slot=0):
    """x (rows, cols) bf16 -> (cols, pad64(rows)) bf16 with zero padding (a K-padded GEMM operand).
    Without `out` the result lives in a pooled scratch buffer keyed by (shape, slot): it is valid until the next
    transpose of the same shape and slot (the pad columns are zeroed once, when the buffer is created)."""
    t = _t()
    rows, cols = x.shape
    if out is None:
        key = (rows, cols, slot, str(x.device))
        out = _tr_pool.get(key)
        if out is None:
            out = _tr_pool[key] = t.zeros((cols, pad64(rows)), dtype=t.bfloat16, device=x.device)
    _lib.check(_lib.load().ma_transpose_bf16(_p(x), x.stride(0), rows, cols, _p(out), out.stride(0), _p(colsum), _s()),
               "transpose")
    return out


def gemm_splitk(a, w, out, alpha=1.0, accumulate=True):
    """out (M, N) f32 (+)= alpha * a (M, K) @ w (N, K)^T."""
    t = _t()
    lib = _lib.load()
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k and tuple(out.shape) == (m, n) and out.dtype == t.float32
    ws = _host.workspace(lib.ma_gemm_splitk_workspace_bytes(m, n, k), a.device)
    _lib.check(lib.ma_gemm_bf16_splitk_f32(_p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), m, n, k,
                                           float(alpha), 1 if accumulate else 0, _p(ws), ws.numel(), _s()), "gemm_splitk")
    return out


def gemm_tn(a, b, out, colsum=None, rows_store=None, alpha=1.0, accumulate=True):
    """out (Mo_store, No) f32 (+)= alpha * a^T @ b for row-major a (Kc, Mo), b (Kc, No) bf16; colsum (+)= column sums
    of a."""
    t = _t()
    lib = _lib.load()
    kc, mo = a.shape
    no = b.shape[1]
    rows_store = mo if rows_store is None else rows_store
    assert b.shape[0] == kc and tuple(out.shape) == (rows_store, no) and out.dtype == t.float32
    ws = _host.workspace(lib.ma_gemm_tn_workspace_bytes(mo, no, kc), a.device)
    _lib.check(lib.ma_gemm_tn_bf16_f32(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), mo, no, kc,
                                       rows_store, float(alpha), 1 if accumulate else 0, _p(colsum), _p(ws), ws.numel(),
                                       _s()), "gemm_tn")
    return out


def gemm_tn_partial(a, b, partial, with_colsum=False, rows_store=None):
    """The split-K partial products of gemm_tn into `partial` (a flat uint8 view of >= ma_gemm_tn_workspace_bytes bytes) and, behind
    them, one partial column-sum vector of `a` per split; the sums are taken later by ma_reduce_splits_batch_f32."""
    lib = _lib.load()
    kc, mo = a.shape
    no = b.shape[1]
    _lib.check(lib.ma_gemm_tn_partial_bf16(_p(a), a.stride(0), _p(b), b.stride(0), mo, no, kc, mo if rows_store is None else rows_store,
                                           1 if with_colsum else 0, _p(partial), partial.numel(), _s()), "gemm_tn_partial")


def gemm_tn_partial_group(triples, with_colsum=True):
    """gemm_tn_partial for a list of (a, b, partial) in one launch per eight products (same partials, bit for bit)."""
    n = len(triples)
    items = (_lib.TnItem * n)()
    for it, (a, b, part) in zip(items, triples):
        kc, mo = a.shape
        it.A, it.B, it.partial = a.data_ptr(), b.data_ptr(), part.data_ptr()
        it.lda, it.ldb, it.Mo, it.No, it.Kc, it.Mo_store = a.stride(0), b.stride(0), mo, b.shape[1], kc, mo
        it.partial_bytes, it.with_colsum = part.numel(), 1 if with_colsum else 0
    _lib.check(_lib.load().ma_gemm_tn_partial_group_bf16(items, n, _s()), "gemm_tn_partial_group")


def gemm_tn_direct_ok(a, b, out):
    """True if out = a^T b can run on the no-split-K kernel (ma_gemm_tn_direct_group_bf16: 256 x 256 tiles)."""
    mo, no = a.shape[1], b.shape[1]
    return (mo % 256 == 0 and no % 256 == 0 and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and out.stride(0) % 4 == 0 and
            a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0)


def gemm_tn_direct_group(quads):
    """out (Mo, No) f32 = a^T b and colsum (Mo) = column sums of a (or None) for a list of (a, b, out, colsum): one launch per
    ma_gemm_tn_direct_max_items() products, every 256 x 256 tile with the full contraction (stored, not accumulated)."""
    lib = _lib.load()
    cap = int(lib.ma_gemm_tn_direct_max_items())
    for lo in range(0, len(quads), cap):
        part = quads[lo:lo + cap]
        items = (_lib.TnDirectItem * len(part))()
        for it, (a, b, out, colsum) in zip(items, part):
            it.A, it.B, it.out, it.colsum = a.data_ptr(), b.data_ptr(), out.data_ptr(), (colsum.data_ptr() if colsum is not None else None)
            it.lda, it.ldb, it.ldo = a.stride(0), b.stride(0), out.stride(0)
            it.Mo, it.No, it.Kc = a.shape[1], b.shape[1], a.shape[0]
        _lib.check(lib.ma_gemm_tn_direct_group_bf16(items, len(part), _s()), "gemm_tn_direct_group")


class DirectGroup:
    """A growing host table of weight-gradient products for ma_gemm_tn_direct_group_bf16: add() fills the next slot when the product
    is queued (so that the launch itself is one C call - the table of 48 products took 0.1 ms of Python to build at launch time, with
    the GPU idle behind it), launch() issues what has been added and keeps the operands referenced until clear()."""

    def __init__(self):
        self.cap = int(_lib.load().ma_gemm_tn_direct_max_items())
        self.items = (_lib.TnDirectItem * self.cap)()
        self.n = 0
        self.keep = []

    def add(self, a, b, out, colsum):
        if self.n == self.cap:
            # a full table leaves as its own grid: the kernel STORES its products (one per weight), so cutting a group in two changes
            # nothing but the launch count (a 7-layer decoder queues 49 products into one group)
            self.launch()
            self.clear()
        it = self.items[self.n]
        it.A, it.B, it.out, it.colsum = a.data_ptr(), b.data_ptr(), out.data_ptr(), (colsum.data_ptr() if colsum is not None else None)
        it.lda, it.ldb, it.ldo = a.stride(0), b.stride(0), out.stride(0)
        it.Mo, it.No, it.Kc = a.shape[1], b.shape[1], a.shape[0]
        self.n += 1
        self.keep.append((a, b))

    def launch(self):
        if self.n:
            _lib.check(_lib.load().ma_gemm_tn_direct_group_bf16(self.items, self.n, _s()), "gemm_tn_direct_group")

    def clear(self):
        self.n = 0
        self.keep = []


def conv2d_dw_workspace_bytes(rows, c, cout):
    return int(_lib.load().ma_conv2d_3x3s2_dw_workspace_bytes(rows, c, cout))


def conv2d_dw(dy, act, dw, dbias, ws=None):
    """dw (Cout, 9C) f32 += dy^T @ im2col(act); dbias (Cout) += column sums.  dy (B*Ho*Wo, Cout) bf16, act NHWC bf16.
    ws: a private split-K workspace (uint8, >= conv2d_dw_workspace_bytes) when the call runs beside other products on another stream."""
    lib = _lib.load()
    b, h, w, c = act.shape
    cout = dy.shape[1]
    if ws is None:
        ws = _host.workspace(lib.ma_conv2d_3x3s2_dw_workspace_bytes(dy.shape[0], c, cout), dy.device)
    _lib.check(lib.ma_conv2d_3x3s2_dw_bf16(_p(dy), dy.stride(0), _p(act), b, h, w, c, cout, _p(dw), _p(dbias), _p(ws),
                                           ws.numel(), _s()), "conv2d_dw")


def layernorm_bwd(x, gamma, dy, g, dgamma, dbeta, row_scale=None, accumulate=True, eps=1e-5, partials=None):
    """g (+)= dL/dx; dgamma / dbeta += the parameter gradients - or, with `partials` (a float32 buffer of >=
    ma_layernorm_bwd_parts(rows) * 512 elements), the per-workgroup partial (dgamma | dbeta) vectors are left there for the caller's
    batched reduction and dgamma / dbeta are not touched."""
    t = _t()
    ws = partials if partials is not None else _reduce_ws(x.device)
    _lib.check(_lib.load().ma_layernorm_bwd_f32(_p(x), x.stride(0), x.shape[0], x.shape[1], _p(gamma), float(eps),
                                                _p(row_scale), _p(dy), dy.stride(0), 1 if dy.dtype == t.bfloat16 else 0,
                                                _p(g), g.stride(0), 1 if accumulate else 0,
                                                None if partials is not None else _p(dgamma),
                                                None if partials is not None else _p(dbeta), _p(ws),
                                                ws.numel() * ws.element_size(), _s()),
               "layernorm_bwd")
    return g


def act_dropout_fwd(u, p, seed, salt, act=_lib.ACT_SWISH):
    h = _t().empty_like(u)
    _lib.check(_lib.load().ma_act_dropout_fwd_bf16(_p(u), _p(h), u.numel(), act, float(p), seed, salt, _s()),
               "act_dropout")
    return h


def act_dropout_bwd(u, dh, p, seed, salt, out=None, act=_lib.ACT_SWISH):
    du = out if out is not None else _t().empty_like(u)
    _lib.check(_lib.load().ma_act_dropout_bwd_bf16(_p(u), _p(dh), _p(du), u.numel(), act, float(p), seed, salt, _s()),
               "act_dropout_bwd")
    return du


def embed_posenc(tokens, table, pe, L, xscale, p, seed, salt):
    t = _t()
    rows, (v, d) = tokens.numel(), table.shape
    out = t.empty((rows, d), dtype=t.float32, device=table.device)
    _lib.check(_lib.load().ma_embed_posenc_f32(_p(tokens), _p(table), _p(pe), rows, L, d, v, float(xscale), float(p), seed,
                                               salt, _p(out), _s()), "embed_posenc")
    return out


def embed_bwd(tokens, g, dtable, xscale, p, seed, salt, row_keep=None):
    """dtable (V, D) += the embedding's gradient; row_keep (rows,) float32: rows with 0 are skipped (their g is known to be zero)."""
    v, d = dtable.shape
    if row_keep is None:
        _lib.check(_lib.load().ma_embed_bwd_f32(_p(tokens), _p(g), tokens.numel(), d, v, float(xscale), float(p), seed, salt,
                                                _p(dtable), _s()), "embed_bwd")
    else:
        assert row_keep.dtype == _t().float32 and row_keep.numel() == tokens.numel()
        _lib.check(_lib.load().ma_embed_bwd_rows_f32(_p(tokens), _p(g), _p(row_keep), tokens.numel(), d, v, float(xscale), float(p),
                                                     seed, salt, _p(dtable), _s()), "embed_bwd_rows")


def mha_small_fwd(q, k, v, mask, mask_mode, batch, lq, lk, scale, heads=4, d_k=64):
    """q (B*Lq, >=H*64) / k, v (B*Lk, ...) bf16 views -> (ctx (B*Lq, H*64) bf16, probs (B, H, Lq, Lk) f32)."""
    t = _t()
    ctx = t.empty((batch * lq, heads * d_k), dtype=t.bfloat16, device=q.device)
    probs = t.empty((batch, heads, lq, lk), dtype=t.float32, device=q.device)
    _lib.check(_lib.load().ma_mha_small_fwd_bf16(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(mask),
                                                 mask_mode, batch, lq, lk, heads, d_k, float(scale), _p(ctx),
                                                 ctx.stride(0), _p(probs), _s()), "mha_small_fwd")
    return ctx, probs


def mha_small_bwd(q, k, v, probs, ctx, dctx, batch, lq, lk, scale, dq, dk, dv, heads=4, d_k=64):
    _lib.check(_lib.load().ma_mha_small_bwd_bf16(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(probs),
                                                 _p(ctx), ctx.stride(0), _p(dctx), dctx.stride(0), batch, lq, lk, heads,
                                                 d_k, float(scale), _p(dq), dq.stride(0), _p(dk), dk.stride(0), _p(dv),
                                                 dv.stride(0), _s()), "mha_small_bwd")


def label_smoothing_loss_grad(logits, V, target, mask, smoothing, grad_scale, normalize_length=False, bufs=None):
    """-> (stats (3,) f32 = [sum kl, correct, tokens], dlogits (rows, ld) bf16).  normalize_length: the gradient is divided by the
    number of unmasked tokens (a device-side sum of `mask`) instead of what the caller folded into grad_scale.
    bufs: a dict that receives the output / scratch tensors of the first call and hands them back to the later ones (the launch table's
    steps: the outputs stay at the addresses the recorded neighbours of this call read)."""
    return _label_smoothing(logits, V, target, mask, smoothing, grad_scale, normalize_length, False, bufs)


def _label_smoothing(logits, V, target, mask, smoothing, grad_scale, normalize_length, out_f32, bufs=None):
    t = _t()
    rows = logits.shape[0]
    if bufs:
        stats, row_stats, dlog = bufs["stats"], bufs["row_stats"], bufs["dlog"]
    else:
        stats = t.empty(3, dtype=t.float32, device=logits.device)  # (the kernel stores all three)
        row_stats = t.empty(rows * 3, dtype=t.float32, device=logits.device)
        dlog = t.empty((rows, logits.stride(0)), dtype=t.float32 if out_f32 else t.bfloat16, device=logits.device)
        if bufs is not None:
            bufs.update(stats=stats, row_stats=row_stats, dlog=dlog)
    denom = mask.sum().reshape(1).to(t.float32) if normalize_length else None
    fn = _lib.load().ma_label_smoothing_loss_grad_len_x32 if out_f32 else _lib.load().ma_label_smoothing_loss_grad_len_f32
    _lib.check(fn(_p(logits), logits.stride(0), rows, V, _p(target), _p(mask), float(smoothing), float(grad_scale), _p(denom),
                  _p(dlog), dlog.stride(0), _p(stats), _p(row_stats), _s()), "label_smoothing")
    return stats, dlog


def dropout_add(x, y, alpha, p, seed, salt, out=None):
    """out (default: a new tensor) = x + alpha * dropout(y); pass out=x for the in-place form; x = None: no residual term."""
    t = _t()
    if out is None:
        out = t.empty(y.shape, dtype=t.float32, device=y.device)
    _lib.check(_lib.load().ma_dropout_add_f32(_p(out), out.stride(0), _p(x), x.stride(0) if x is not None else 0, _p(y),
                                              y.stride(0), 1 if y.dtype == t.bfloat16 else 0, y.shape[0], y.shape[1],
                                              float(alpha), float(p), seed, salt, _s()), "dropout_add")
    return out


def dropout_bwd(g, alpha, p, seed, salt, row_scale=None):
    t = _t()
    dy = t.empty(g.shape, dtype=t.bfloat16, device=g.device)
    _lib.check(_lib.load().ma_dropout_bwd_bf16(_p(g), g.stride(0), _p(dy), dy.stride(0), g.shape[0], g.shape[1],
                                               float(alpha), _p(row_scale), float(p), seed, salt, _s()), "dropout_bwd")
    return dy


def convmid_fwd_train(y, batch, T, dw_w, dw_b, gamma, beta, run_mean, run_var, eps=1e-5, momentum=0.1):
    """y (B*T, 2C) bf16 -> (out bf16 (B*T, C), z f32, stats f32 (2C)); updates the running statistics in place."""
    t = _t()
    lib = _lib.load()
    c, ks = dw_w.shape
    rows = batch * T
    z = t.empty((rows, c), dtype=t.float32, device=y.device)
    nparts = int(lib.ma_convmid_fwd_train_parts(batch, T, c))
    _lib.check(min(nparts, 0), "convmid_fwd_train")
    sums = t.empty(nparts * 2 * c, dtype=t.float32, device=y.device)  # per-workgroup partial (sum | sum of squares) vectors
    stats = t.empty(2 * c, dtype=t.float32, device=y.device)
    out = t.empty((rows, c), dtype=t.bfloat16, device=y.device)
    _lib.check(lib.ma_convmid_fwd_train(_p(y), y.stride(0), batch, T, c, _p(dw_w), ks, _p(dw_b), _p(z), _p(sums), _s()),
               "convmid_fwd_train")
    _lib.check(lib.ma_bn_finalize_f32(_p(sums), nparts, c, rows, float(eps), float(momentum), _p(run_mean), _p(run_var),
                                      _p(stats), _s()), "bn_finalize")
    _lib.check(lib.ma_bn_swish_fwd_bf16(_p(z), _p(stats), _p(gamma), _p(beta), _p(out), rows, c, _s()), "bn_swish_fwd")
    return out, z, stats


def convmid_bwd(dout, y, z, stats, batch, T, dw_w, gamma, beta, d_dw_w, d_dw_b, d_gamma, d_beta, partials=None):
    """dout (B*T, C) bf16 -> dy (B*T, 2C) bf16; parameter gradients accumulate into the given float32 buffers.  partials (float32,
    >= ma_convmid_bwd_parts * C * (k + 1)): the depthwise kernel's / bias' per-workgroup partial sums are left there (d_dw_w / d_dw_b are
    not touched) for the caller's batched sum."""
    t = _t()
    lib = _lib.load()
    c, ks = dw_w.shape
    rows = batch * T
    dz = t.empty((rows, c), dtype=t.float32, device=y.device)
    dsum = t.empty(2 * c, dtype=t.float32, device=y.device)
    rw = _reduce_ws(y.device)
    dy = t.empty((rows, 2 * c), dtype=t.bfloat16, device=y.device)
    if partials is not None:
        # (the fused step: the BatchNorm backward's second stage rides in the depthwise backward's loads)
        _lib.check(lib.ma_bn_swish_bwd_stage1_f32(_p(dout), _p(z), _p(stats), _p(gamma), _p(beta), _p(dz), rows, c, _p(dsum), _p(d_gamma),
                                                  _p(d_beta), _p(rw), rw.numel(), _s()), "bn_swish_bwd_stage1")
        _lib.check(lib.ma_convmid_bwd_bn_bf16(_p(dz), _p(z), _p(stats), _p(gamma), _p(dsum), _p(y), y.stride(0), batch, T, c, _p(dw_w), ks,
                                              _p(dy), dy.stride(0), None, None, _p(partials),
                                              partials.numel() * partials.element_size(), _s()), "convmid_bwd_bn")
        return dy
    _lib.check(lib.ma_bn_swish_bwd_f32(_p(dout), _p(z), _p(stats), _p(gamma), _p(beta), _p(dz), rows, c, _p(dsum), _p(d_gamma), _p(d_beta),
               _p(rw), rw.numel(), _s()), "bn_swish_bwd")
    rw = _reduce_ws(y.device)
    _lib.check(lib.ma_convmid_bwd_bf16(_p(dz), _p(y), y.stride(0), batch, T, c, _p(dw_w), ks, _p(dy), dy.stride(0),
                                       _p(d_dw_w), _p(d_dw_b), _p(rw), rw.numel(), _s()), "convmid_bwd")
    return dy


def relu_bwd(dy, y):
    _lib.check(_lib.load().ma_relu_bwd_bf16(_p(dy), _p(y), dy.numel(), _s()), "relu_bwd")
    return dy


def im2col_t(act):
    """act (B, H, W, C) bf16 NHWC -> (9C, pad64(B*Ho*Wo)) bf16."""
    t = _t()
    b, h, w, c = act.shape
    m = b * ((h - 3) // 2 + 1) * ((w - 3) // 2 + 1)
    key = ("im2col", 9 * c, m, str(act.device))
    out = _tr_pool.get(key)
    if out is None:  # pooled like transpose(): the pad columns are zeroed once
        out = _tr_pool[key] = t.zeros((9 * c, pad64(m)), dtype=t.bfloat16, device=act.device)
    _lib.check(_lib.load().ma_im2col_t_3x3s2_nhwc_bf16(_p(act), b, h, w, c, _p(out), out.stride(0), _s()), "im2col_t")
    return out


def col2im_relu(dcol, act):
    dact = _t().empty_like(act)
    b, h, w, c = act.shape
    _lib.check(_lib.load().ma_col2im_3x3s2_relu_bf16(_p(dcol), _p(act), b, h, w, c, _p(dact), _s()), "col2im")
    return dact


_ZERO_ROWS = {}


def conv2_dinput(dy, wt, act):
    """dact (B, H, W, C) bf16 = relu'(act) * conv_transpose(dy (B*Ho*Wo, C), W): one implicit-GEMM launch (no dcol intermediate).
    wt = the transposed weight ((kh, kw, c), co)."""
    t = _t()
    b, h, w, c = act.shape
    z = _ZERO_ROWS.get(act.device)
    if z is None:
        z = _ZERO_ROWS[act.device] = t.zeros(128, dtype=t.uint8, device=act.device)
    dact = t.empty_like(act)
    _lib.check(_lib.load().ma_conv2d_3x3s2_dinput_bf16(_p(dy), b, h, w, c, _p(wt), _p(act), _p(z), _p(dact), _s()), "conv2_dinput")
    return dact


def conv1_dw(dact, x, cmvn_mean, cmvn_istd, dw, db):
    b, tt, idim = x.shape
    rw = _reduce_ws(x.device)
    _lib.check(_lib.load().ma_subsample_conv1_dw_f32(_p(dact), _p(x), b, tt, idim, _p(cmvn_mean), _p(cmvn_istd),
                                                     dact.shape[-1], _p(dw), _p(db), _p(rw), rw.numel(), _s()), "conv1_dw")


def grad_overflow(g, flag):
    _lib.check(_lib.load().ma_grad_overflow_f32(_p(g), g.numel(), _p(flag), _s()), "grad_overflow")


def adam(param, grad, m, v, lr_t, beta1, beta2, eps, inv_scale, overflow=None, mirror=None):
    """mirror (bf16, param.numel()): also receives the bf16 conversion of the updated parameters (one launch instead of Adam + cast);
    returns True when it did (False: the shape is not covered and the caller casts)."""
    lib = _lib.load()
    if mirror is not None:
        rc = lib.ma_adam_mirror_f32(_p(param), _p(grad), _p(m), _p(v), param.numel(), float(lr_t), float(beta1), float(beta2), float(eps),
                                    float(inv_scale), _p(overflow), _p(mirror), _s())
        if rc != _lib.MA_ERR_UNSUPPORTED:
            _lib.check(rc, "adam_mirror")
            return True
    _lib.check(lib.ma_adam_f32(_p(param), _p(grad), _p(m), _p(v), param.numel(), float(lr_t), float(beta1),
                               float(beta2), float(eps), float(inv_scale), _p(overflow), _s()), "adam")
    return False


def attention_fwd(qkv, pos, bias_u, bias_v, mask, batch, T, heads=4, d_k=64):
    """Training forward: (ctx (B*T, 256) bf16, lse (B, H, T) f32)."""
    t = _t()
    lib = _lib.load()
    ctx = t.empty((batch * T, heads * d_k), dtype=t.bfloat16, device=qkv.device)
    lse = t.empty((batch, heads, T), dtype=t.float32, device=qkv.device)
    ws_bytes = lib.ma_relpos_attention_workspace_bytes(batch, T, heads, d_k)
    ws = t.empty(ws_bytes, dtype=t.uint8, device=qkv.device)
    fn = lib.ma_relpos_attention_train_bf16
    if mask is not None and mask.dim() == 3:  # (B, T, T) chunk masks
        assert tuple(mask.shape) == (batch, T, T) and mask.dtype == t.float32 and mask.is_contiguous()
        fn = lib.ma_relpos_attention_train_qmask_bf16
    _lib.check(fn(_p(qkv), qkv.stride(0), _p(pos), pos.stride(0), _p(bias_u), _p(bias_v), _p(mask), batch, T, heads, d_k,
                  _p(ctx), ctx.stride(0), _p(ws), ws_bytes, _p(lse), _s()), "attention_fwd")
    return ctx, lse


def attention_bwd(qkv, pos, bias_u, bias_v, mask, ctx, dctx, lse, batch, T, dpos, dbias_u, dbias_v, heads=4, d_k=64, ws=None):
    """-> dqkv (B*T, 768) bf16; dpos (T, 256), dbias_u/v (H, 64) float32 accumulate - or, with dpos = None and a caller-owned
    workspace `ws`, the partial sums stay in `ws` for the caller's batched reduction (ma_relpos_attention_bwd_layout)."""
    t = _t()
    lib = _lib.load()
    dqkv = t.empty((batch * T, 3 * heads * d_k), dtype=t.bfloat16, device=qkv.device)
    ws_bytes = lib.ma_relpos_attention_bwd_workspace_bytes(batch, T, heads, d_k)
    if ws is None:
        ws = t.empty(ws_bytes, dtype=t.uint8, device=qkv.device)
    ws_bytes = ws.numel() * ws.element_size()
    fn = lib.ma_relpos_attention_bwd_bf16
    if mask is not None and mask.dim() == 3:
        assert tuple(mask.shape) == (batch, T, T) and mask.dtype == t.float32 and mask.is_contiguous()
        fn = lib.ma_relpos_attention_bwd_qmask_bf16
    _lib.check(fn(_p(qkv), qkv.stride(0), _p(pos), pos.stride(0), _p(bias_u), _p(bias_v), _p(mask), _p(ctx), ctx.stride(0),
                  _p(dctx), dctx.stride(0), _p(lse), batch, T, heads, d_k, _p(dqkv), dqkv.stride(0), _p(dpos),
                  dpos.stride(0) if dpos is not None else 0, _p(dbias_u), _p(dbias_v), _p(ws), ws_bytes, _s()), "attention_bwd")
    return dqkv


def ctc_loss_grad(logits, V, batch, T, ys_pad, hlens, ys_lens, grad_scale, blank=0):
    """logits (B*T, ld>=V) float32 -> (loss scalar tensor, per-utterance nll, dlogits (B*T, ld) bf16 scaled by grad_scale)."""
    t = _t()
    lib = _lib.load()
    dev = logits.device
    ys_pad = ys_pad.to(t.int32).contiguous()
    hlens = hlens.to(t.int32).contiguous()
    ys_lens = ys_lens.to(t.int32).contiguous()
    per = t.empty(batch, dtype=t.float32, device=dev)
    lse = t.empty(batch * T, dtype=t.float32, device=dev)
    out = t.empty(1, dtype=t.float32, device=dev)
    dlog = t.empty((batch * T, logits.stride(0)), dtype=t.bfloat16, device=dev)
    ws_bytes = lib.ma_ctc_grad_workspace_bytes(batch, T, ys_pad.shape[1])
    ws = t.empty(ws_bytes, dtype=t.uint8, device=dev)
    _lib.check(lib.ma_ctc_loss_grad_f32(_p(logits), logits.stride(0), batch, T, V, _p(ys_pad), ys_pad.shape[1],
                                        _p(hlens), _p(ys_lens), blank, 1, float(grad_scale), _p(per), _p(lse), _p(out),
                                        _p(dlog), dlog.stride(0), _p(ws), ws_bytes, _s()), "ctc_loss_grad")
    return out[0], per, dlog


# ---- training forms of the packed dense layers (csrc/gemm_k256.hip, csrc/rows_packed.hip) --------------------------------------
def _train_epi(mode, bias=None, aux=None, out2=None, residual=None, row_scale=None, alpha=1.0, p=0.0, seed=0, salt=0, ln1=None,
               ln2=None, ln_row_scale=None, ln_out=None, ln_mid=None, eps=1e-5):
    t = _t()
    e = _lib.TrainEpilogue()
    e.mode = mode
    e.bias = bias.data_ptr() if bias is not None else None
    if aux is not None:
        e.aux, e.ld_aux = aux.data_ptr(), aux.stride(0)
    if out2 is not None:
        e.out2, e.ldo2 = out2.data_ptr(), out2.stride(0)
    if residual is not None:
        e.residual, e.ldr = residual.data_ptr(), residual.stride(0)
    e.row_scale = row_scale.data_ptr() if row_scale is not None else None
    e.alpha, e.p, e.seed, e.salt = float(alpha), float(p), int(seed), int(salt)
    if ln1 is not None:
        e.ln_gamma1, e.ln_beta1 = ln1[0].data_ptr(), ln1[1].data_ptr()
        e.ln_out, e.ld_ln, e.ln_out_bf16 = ln_out.data_ptr(), ln_out.stride(0), 1 if ln_out.dtype == t.bfloat16 else 0
    if ln2 is not None:
        e.ln_gamma2, e.ln_beta2 = ln2[0].data_ptr(), ln2[1].data_ptr()
        e.ln_mid, e.ld_mid = ln_mid.data_ptr(), ln_mid.stride(0)
    e.ln_row_scale = ln_row_scale.data_ptr() if ln_row_scale is not None else None
    e.ln_eps = float(eps)
    return e


def dense_act_drop(a, packed, n, bias, p, seed, salt, act=_lib.ACT_SWISH):
    """w_1 forward (K = 256): -> (u (M, n) bf16 = a W^T + b, h (M, n) bf16 = dropout(act(u))), one launch; act = Swish or ReLU."""
    import ctypes

    t = _t()
    m = a.shape[0]
    u = t.empty((m, n), dtype=t.bfloat16, device=a.device)
    h = t.empty((m, n), dtype=t.bfloat16, device=a.device)
    e = _train_epi(1, bias=bias, out2=h, p=p, seed=seed, salt=salt)
    e.act = 2 if act == _lib.ACT_RELU else 0
    _lib.check(_lib.load().ma_gemm_k256_train_bf16(_p(a), a.stride(0), _p(packed), _p(u), u.stride(0), m, n, ctypes.byref(e), _s()),
               "dense_act_drop")
    return u, h


def dense_act_drop_bwd(dy, packed, n, u, p, seed, salt, act=_lib.ACT_SWISH):
    """w_1 backward (K = 256): du (M, n) bf16 = (dy W2) * swish'(u) * keep / (1 - p); `packed` = the k256 packing of W2^T (n, 256)."""
    import ctypes

    t = _t()
    m = dy.shape[0]
    du = t.empty((m, n), dtype=t.bfloat16, device=dy.device)
    e = _train_epi(2, aux=u, p=p, seed=seed, salt=salt)
    e.act = 2 if act == _lib.ACT_RELU else 0
    _lib.check(_lib.load().ma_gemm_k256_train_bf16(_p(dy), dy.stride(0), _p(packed), _p(du), du.stride(0), m, n, ctypes.byref(e), _s()),
               "dense_act_drop_bwd")
    return du


def dense_plain(a, packed, n, k, bias=None):
    """out (M, n) bf16 = a (M, k) W^T (+ bias) on a packed weight: K = 256 (any n % 256 == 0) or n = 256 (any k % 64 == 0)."""
    import ctypes

    t = _t()
    m = a.shape[0]
    out = t.empty((m, n), dtype=t.bfloat16, device=a.device)
    e = _train_epi(4, bias=bias)
    lib = _lib.load()
    if k == 256:
        _lib.check(lib.ma_gemm_k256_train_bf16(_p(a), a.stride(0), _p(packed), _p(out), out.stride(0), m, n, ctypes.byref(e), _s()),
                   "dense_plain")
    else:
        _lib.check(lib.ma_gemm_rows_train_bf16(_p(a), a.stride(0), m, k, _p(packed), _p(out), out.stride(0), ctypes.byref(e), _s()),
                   "dense_plain")
    return out


def rows_train_parts(m):
    """Workgroups (= partial (dgamma | dbeta) vectors of dense_lnbwd) that ma_gemm_rows_train_bf16 launches for m rows."""
    return int(_lib.load().ma_gemm_rows_train_parts(m))


def dense_lnbwd(a, packed, k, x, gamma, g, partials, nxt=None, row_scale=None, eps=1e-5):
    """An input-gradient product (N = 256, k % 64 == 0, k != 256) with the LayerNorm backward that consumes it in the epilogue:
    dy = bf16(a W^T) * row_scale; g (M, 256) float32 += dLN/dx(dy) for the LayerNorm of input x and weight gamma, in place; the
    per-workgroup partial (dgamma | dbeta) vectors go to `partials` (float32, >= rows_train_parts(M) * 512); with
    nxt = (alpha, p, seed, salt, row_scale_next or None) also dy_next (M, 256) bf16 = dropout(g * alpha * row_scale_next).
    = dense_plain + layernorm_bwd_next in one launch (round 4).  Returns dy_next (or None)."""
    import ctypes

    t = _t()
    m = a.shape[0]
    assert k != 256 and g.dtype == t.float32 and x.dtype == t.float32 and partials.numel() >= rows_train_parts(m) * 512
    e = _lib.TrainEpilogue()
    e.mode = 5
    e.residual, e.ldr = x.data_ptr(), x.stride(0)
    e.ln_gamma1 = gamma.data_ptr()
    e.ln_eps = float(eps)
    e.row_scale = row_scale.data_ptr() if row_scale is not None else None
    e.ln_mid = partials.data_ptr()
    dy_next = None
    if nxt is not None:
        alpha, p, seed, salt, rs_next = nxt
        dy_next = t.empty((m, 256), dtype=t.bfloat16, device=a.device)
        e.ln_out, e.ld_ln, e.ln_out_bf16 = dy_next.data_ptr(), dy_next.stride(0), 1
        e.alpha, e.p, e.seed, e.salt = float(alpha), float(p), int(seed), int(salt)
        e.ln_row_scale = rs_next.data_ptr() if rs_next is not None else None
    _lib.check(_lib.load().ma_gemm_rows_train_bf16(_p(a), a.stride(0), m, k, _p(packed), _p(g), g.stride(0), ctypes.byref(e), _s()),
               "dense_lnbwd")
    return dy_next


def dense_join(a, packed, k, bias, residual, alpha, p, seed, salt, row_scale=None, ln1=None, ln2=None, ln_row_scale=None,
               ln_out_dtype=None, eps=1e-5):
    """Branch join (N = 256): x_out (M, 256) float32 = residual + alpha * dropout(bf16((a W^T + b) * row_scale)), and optionally
    LayerNorm(x_out; ln1) [* ln_row_scale] (-> bf16, or ln_out_dtype), or the chain ln_mid = LayerNorm(x_out; ln1) float32,
    ln_out = LayerNorm(ln_mid; ln2).  Returns (x_out, ln_out, ln_mid)."""
    import ctypes

    t = _t()
    m = a.shape[0]
    out = t.empty((m, 256), dtype=t.float32, device=a.device)
    ln_out = ln_mid = None
    if ln1 is not None:
        ln_out = t.empty((m, 256), dtype=ln_out_dtype or t.bfloat16, device=a.device)
    if ln2 is not None:
        ln_mid = t.empty((m, 256), dtype=t.float32, device=a.device)
    e = _train_epi(3, bias=bias, residual=residual, row_scale=row_scale, alpha=alpha, p=p, seed=seed, salt=salt, ln1=ln1, ln2=ln2,
                   ln_row_scale=ln_row_scale, ln_out=ln_out, ln_mid=ln_mid, eps=eps)
    lib = _lib.load()
    if k == 256:
        _lib.check(lib.ma_gemm_k256_train_bf16(_p(a), a.stride(0), _p(packed), _p(out), out.stride(0), m, 256, ctypes.byref(e), _s()),
                   "dense_join")
    else:
        _lib.check(lib.ma_gemm_rows_train_bf16(_p(a), a.stride(0), m, k, _p(packed), _p(out), out.stride(0), ctypes.byref(e), _s()),
                   "dense_join")
    return out, ln_out, ln_mid


def dense_join_splitk(a, w, bias, residual, alpha, p, seed, salt, ln1=None, eps=1e-5):
    """dense_join for a long contraction over few rows, on the weight w (256, K) as it lies in the bf16 mirror: the product is split
    over K, the launch that sums the splits carries the join (ma_gemm_bf16_splitk_join_f32).  Returns (x_out, ln_out)."""
    import ctypes

    t = _t()
    m, k = a.shape
    assert tuple(w.shape) == (256, k)
    lib = _lib.load()
    out = t.empty((m, 256), dtype=t.float32, device=a.device)
    ln_out = t.empty((m, 256), dtype=t.bfloat16, device=a.device) if ln1 is not None else None
    e = _train_epi(3, bias=bias, residual=residual, alpha=alpha, p=p, seed=seed, salt=salt, ln1=ln1, ln_out=ln_out, eps=eps)
    ws = _host.workspace(lib.ma_gemm_splitk_workspace_bytes(m, 256, k), a.device)
    _lib.check(lib.ma_gemm_bf16_splitk_join_f32(_p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), m, 256, k,
                                                ctypes.byref(e), _p(ws), ws.numel(), _s()), "dense_join_splitk")
    return out, ln_out


_NO_TAPE = {}


def ffn_train(a, packed, hidden, b1, p_hidden, seed, salt_hidden, b2, residual, alpha, p_join, salt_join, ln1=None, ln2=None,
              ln_row_scale=None, eps=1e-5, tape_derivative=False, tape=True):
    """The whole feed-forward module in training mode, one launch (ma_ffn_train_bf16; `packed` = the feed-forward block format of
    W1 and W2): -> (u, h (M, hidden) bf16 [the tape], x_out (M, 256) float32, ln_out, ln_mid) - what dense_act_drop followed by
    dense_join return.  tape_derivative: u is gk = swish'(pre-activation) * keep / (1 - p) instead (what ffn_train_bwd reads)."""
    import ctypes

    t = _t()
    m = a.shape[0]
    if tape:
        uh, ldu = t.empty((2, m, hidden), dtype=t.bfloat16, device=a.device), hidden
    else:  # forward only (tape=False): u / h are not kept - two scratch areas of 2 * hidden bytes, row stride 0 (130 MB less per module)
        key = (a.device, hidden)
        if key not in _NO_TAPE:
            _NO_TAPE[key] = t.empty((2, hidden), dtype=t.bfloat16, device=a.device)
        uh, ldu = _NO_TAPE[key], 0
    out = t.empty((m, 256), dtype=t.float32, device=a.device)
    ln_out = ln_mid = None
    if ln1 is not None:
        ln_out = t.empty((m, 256), dtype=t.bfloat16, device=a.device)
    if ln2 is not None:
        ln_mid = t.empty((m, 256), dtype=t.float32, device=a.device)
    e = _train_epi(3, bias=b2, residual=residual, alpha=alpha, p=p_join, seed=seed, salt=salt_join, ln1=ln1, ln2=ln2,
                   ln_row_scale=ln_row_scale, ln_out=ln_out, ln_mid=ln_mid, eps=eps)
    _lib.check(_lib.load().ma_ffn_train_bf16(_p(a), a.stride(0), m, hidden, _p(packed), _p(b1), _p(uh[0]), _p(uh[1]), ldu,
                                             1 if tape_derivative else 0, float(p_hidden), int(seed), int(salt_hidden), _p(out),
                                             out.stride(0), ctypes.byref(e), _s()), "ffn_train")
    if not tape:
        return None, None, out, ln_out, ln_mid
    return uh[0], uh[1], out, ln_out, ln_mid


def ffn_train_parts(m):
    """Workgroups (= partial (dgamma | dbeta) vectors) of ffn_train_bwd for m rows."""
    return int(_lib.load().ma_ffn_train_parts(m))


def ffn_train_bwd(dy, packed_t, hidden, gk, x, gamma, g, partials, nxt=None, eps=1e-5, chain=None):
    """The feed-forward module's backward in one launch (ma_ffn_train_bwd_bf16): du (M, hidden) bf16 = bf16(dy W2) * gk [returned: the
    operand of w_1's weight gradient]; da = du W1 goes straight into the backward of the LayerNorm in front of the module (input x,
    weight gamma): g (M, 256) float32 += dLN/dx(da) in place, per-workgroup (dgamma | dbeta) partials -> `partials`
    (>= ffn_train_parts(M) * 512 floats), and with nxt = (alpha, p, seed, salt, row_scale_next or None) dy_next (M, 256) bf16 =
    dropout(g * alpha * row_scale_next).  `packed_t` = the feed-forward block format of (W2^T, W1^T).  Returns (du, dy_next).
    chain = (x2, gamma2, partials2, nxt2) (nxt must be None then): a second LayerNorm backward on the finished rows, for the LayerNorm
    (input x2, weight gamma2) whose OUTPUT the rows of g are: g is replaced by its backward and dy_next comes from nxt2."""
    import ctypes

    t = _t()
    m = dy.shape[0]
    assert g.dtype == t.float32 and x.dtype == t.float32 and partials.numel() >= ffn_train_parts(m) * 512
    du = t.empty((m, hidden), dtype=t.bfloat16, device=dy.device)
    e = _lib.TrainEpilogue()
    e.mode = 5
    e.residual, e.ldr = x.data_ptr(), x.stride(0)
    e.ln_gamma1 = gamma.data_ptr()
    e.ln_eps = float(eps)
    e.ln_mid = partials.data_ptr()
    dy_next = None
    if nxt is not None:
        alpha, p, seed, salt, rs_next = nxt
        dy_next = t.empty((m, 256), dtype=t.bfloat16, device=dy.device)
        e.ln_out, e.ld_ln, e.ln_out_bf16 = dy_next.data_ptr(), dy_next.stride(0), 1
        e.alpha, e.p, e.seed, e.salt = float(alpha), float(p), int(seed), int(salt)
        e.ln_row_scale = rs_next.data_ptr() if rs_next is not None else None
    e2 = None
    if chain is not None:
        assert nxt is None
        x2, gamma2, partials2, nxt2 = chain
        assert x2.dtype == t.float32 and partials2.numel() >= ffn_train_parts(m) * 512
        e2 = _lib.TrainEpilogue()
        e2.mode = 5
        e2.residual, e2.ldr = x2.data_ptr(), x2.stride(0)
        e2.ln_gamma1 = gamma2.data_ptr()
        e2.ln_eps = float(eps)
        e2.ln_mid = partials2.data_ptr()
        if nxt2 is not None:
            alpha, p, seed, salt, rs_next = nxt2
            dy_next = t.empty((m, 256), dtype=t.bfloat16, device=dy.device)
            e2.ln_out, e2.ld_ln, e2.ln_out_bf16 = dy_next.data_ptr(), dy_next.stride(0), 1
            e2.alpha, e2.p, e2.seed, e2.salt = float(alpha), float(p), int(seed), int(salt)
            e2.ln_row_scale = rs_next.data_ptr() if rs_next is not None else None
    _lib.check(_lib.load().ma_ffn_train_bwd_bf16(_p(dy), dy.stride(0), m, hidden, _p(packed_t), _p(gk), _p(du), hidden, _p(g),
                                                 g.stride(0), ctypes.byref(e), ctypes.byref(e2) if e2 is not None else None, _s()),
               "ffn_train_bwd")
    return du, dy_next


def layernorm_bwd_next(x, gamma, dy, g, dgamma, dbeta, nxt, row_scale=None, accumulate=True, eps=1e-5, partials=None):
    """layernorm_bwd, plus the next branch's dropout_bwd on the finished rows: nxt = (alpha, p, seed, salt, row_scale or None)
    -> (g, dy_next (M, 256) bf16)."""
    t = _t()
    ws = partials if partials is not None else _reduce_ws(x.device)
    alpha, p, seed, salt, rs_next = nxt
    dy_next = t.empty((x.shape[0], 256), dtype=t.bfloat16, device=x.device)
    _lib.check(_lib.load().ma_layernorm_bwd_next_f32(_p(x), x.stride(0), x.shape[0], x.shape[1], _p(gamma), float(eps), _p(row_scale),
                                                     _p(dy), dy.stride(0), 1 if dy.dtype == t.bfloat16 else 0, _p(g), g.stride(0),
                                                     1 if accumulate else 0, None if partials is not None else _p(dgamma),
                                                     None if partials is not None else _p(dbeta), _p(ws),
                                                     ws.numel() * ws.element_size(), _p(dy_next), dy_next.stride(0), float(alpha),
                                                     _p(rs_next), float(p), seed, salt, _s()), "layernorm_bwd_next")
    return g, dy_next
