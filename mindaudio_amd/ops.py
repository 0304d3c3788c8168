"""Thin Python wrappers over the Conformer C-ABI entry points (include/mindaudio_amd.h).
Tensors are torch HIP tensors used as device buffers; all arithmetic happens in the HIP kernels."""
import ctypes

from . import _host, _lib


def _epilogue(bias=None, residual=None, row_scale=None, alpha=1.0, act=_lib.ACT_NONE, out_bf16=True):
    e = _lib.GemmEpilogue()
    e.bias = bias.data_ptr() if bias is not None else None
    e.residual = residual.data_ptr() if residual is not None else None
    e.row_scale = row_scale.data_ptr() if row_scale is not None else None
    e.ldr = residual.stride(0) if residual is not None else 0
    e.alpha = float(alpha)
    e.act = int(act)
    e.out_bf16 = 1 if out_bf16 else 0
    return e


def gemm(a, w, bias=None, residual=None, row_scale=None, alpha=1.0, act=_lib.ACT_NONE, out_dtype=None, out=None):
    """out (M, N) = act(a (M, K) @ w (N, K)^T + bias) * alpha * row_scale[:, None] (+ residual).
    a, w: bf16 device tensors (K contiguous); bias/row_scale/residual float32."""
    t = _host.torch()
    lib = _lib.load()
    assert a.dtype == t.bfloat16 and w.dtype == t.bfloat16 and a.dim() == 2 and w.dim() == 2
    assert a.stride(1) == 1 and w.stride(1) == 1 and a.shape[1] == w.shape[1]
    m, k = a.shape
    n = w.shape[0]
    out_dtype = out_dtype or t.bfloat16
    if out is None:
        out = t.empty((m, n), dtype=out_dtype, device=a.device)
    assert out.dtype == out_dtype and out.stride(1) == 1 and tuple(out.shape) == (m, n)
    for x in (bias, residual, row_scale):
        assert x is None or x.dtype == t.float32
    e = _epilogue(bias, residual, row_scale, alpha, act, out_dtype == t.bfloat16)
    rc = lib.ma_gemm_bf16(_host.ptr(a), a.stride(0), _host.ptr(w), w.stride(0), _host.ptr(out), out.stride(0), m, n, k,
                          ctypes.byref(e), _host.current_stream_ptr())
    _lib.check(rc, "gemm_bf16")
    return out


def conv2d_3x3s2_pack(w):
    """Fragment-ordered packed copy of w (Cout, 3, 3, C) bf16 for conv2d_3x3s2_packed; None if the shape is not covered."""
    t = _host.torch()
    lib = _lib.load()
    assert w.dtype == t.bfloat16 and w.dim() == 4 and w.is_contiguous() and w.shape[1] == 3 and w.shape[2] == 3
    cout, c = w.shape[0], w.shape[3]
    nbytes = lib.ma_conv2d_3x3s2_packed_bytes(c, cout)
    if nbytes < 0:
        return None
    packed = t.empty((nbytes // 2,), dtype=t.bfloat16, device=w.device)
    _lib.check(lib.ma_conv2d_3x3s2_pack_bf16(_host.ptr(w), c, cout, _host.ptr(packed), _host.current_stream_ptr()),
               "conv2d_3x3s2_pack_bf16")
    return packed


def conv2d_3x3s2_packed(act, packed, bias, relu=True, out=None):
    """conv2d_3x3s2_nhwc on a packed weight: act (B, H, W, 256) bf16 NHWC -> (B, Ho, Wo, 256) bf16 (into `out` if given)."""
    t = _host.torch()
    lib = _lib.load()
    assert act.dtype == t.bfloat16 and act.is_contiguous() and bias.dtype == t.float32
    b, h, wd, c = act.shape
    cout = bias.numel()
    ho, wo = (h - 3) // 2 + 1, (wd - 3) // 2 + 1
    if out is None:
        out = t.empty((b, ho, wo, cout), dtype=t.bfloat16, device=act.device)
    assert out.shape == (b, ho, wo, cout) and out.dtype == t.bfloat16 and out.is_contiguous()
    rc = lib.ma_conv2d_3x3s2_packed_nhwc_bf16(_host.ptr(act), b, h, wd, c, _host.ptr(packed), cout, _host.ptr(bias),
                                              1 if relu else 0, _host.ptr(out), _host.current_stream_ptr())
    _lib.check(rc, "conv2d_3x3s2_packed")
    return out


def gemm_rows_pack(w):
    """Fragment-ordered packed copy of w (256, K) bf16 for gemm_rows_packed; None if the shape is not covered."""
    t = _host.torch()
    lib = _lib.load()
    assert w.dtype == t.bfloat16 and w.dim() == 2 and w.stride(1) == 1
    n, k = w.shape
    nbytes = lib.ma_gemm_rows_packed_bytes(n, k)
    if nbytes < 0 or w.stride(0) % 8:
        return None
    packed = t.empty((n, nbytes // (2 * n)), dtype=t.bfloat16, device=w.device)  # K zero-padded to a multiple of 192
    _lib.check(lib.ma_gemm_rows_pack_bf16(_host.ptr(w), w.stride(0), n, k, _host.ptr(packed), _host.current_stream_ptr()),
               "gemm_rows_pack_bf16")
    return packed


def gemm_rows_packed(a, packed, bias, alpha=1.0, out=None):
    """out (M, 256) float32 = alpha * (a (M, K) bf16 @ W^T + bias) on a packed weight (gemm_rows_pack)."""
    t = _host.torch()
    lib = _lib.load()
    assert a.dtype == t.bfloat16 and a.dim() == 2 and a.stride(1) == 1 and 0 <= packed.shape[1] - a.shape[1] < 192
    assert bias.dtype == t.float32
    m, k = a.shape
    n = packed.shape[0]
    if out is None:
        out = t.empty((m, n), dtype=t.float32, device=a.device)
    assert out.dtype == t.float32 and out.stride(1) == 1 and tuple(out.shape) == (m, n)
    rc = lib.ma_gemm_rows_packed_f32(_host.ptr(a), a.stride(0), m, k, _host.ptr(packed), n, _host.ptr(bias), float(alpha),
                                     _host.ptr(out), out.stride(0), _host.current_stream_ptr())
    _lib.check(rc, "gemm_rows_packed_f32")
    return out


def gemm_k256_pack(w):
    """Fragment-ordered packed copy of w (N, 256) bf16 for gemm_packed; None if the shape is not covered (N % 256, K != 256)."""
    t = _host.torch()
    lib = _lib.load()
    assert w.dtype == t.bfloat16 and w.dim() == 2 and w.stride(1) == 1
    n, k = w.shape
    nbytes = lib.ma_gemm_k256_packed_bytes(n, k)
    if nbytes < 0:
        return None
    packed = t.empty((n, k), dtype=t.bfloat16, device=w.device)
    _lib.check(lib.ma_gemm_k256_pack_bf16(_host.ptr(w), w.stride(0), n, k, _host.ptr(packed), _host.current_stream_ptr()),
               "gemm_k256_pack_bf16")
    return packed


def gemm_packed(a, packed, bias=None, residual=None, row_scale=None, alpha=1.0, act=_lib.ACT_NONE, out_dtype=None, out=None):
    """ops.gemm on a packed weight (gemm_k256_pack): a (M, 256) bf16, packed (N, 256)."""
    t = _host.torch()
    lib = _lib.load()
    assert a.dtype == t.bfloat16 and a.dim() == 2 and a.stride(1) == 1 and a.shape[1] == packed.shape[1]
    m, k = a.shape
    n = packed.shape[0]
    out_dtype = out_dtype or t.bfloat16
    if out is None:
        out = t.empty((m, n), dtype=out_dtype, device=a.device)
    assert out.dtype == out_dtype and out.stride(1) == 1 and tuple(out.shape) == (m, n)
    e = _epilogue(bias, residual, row_scale, alpha, act, out_dtype == t.bfloat16)
    rc = lib.ma_gemm_k256_packed_bf16(_host.ptr(a), a.stride(0), _host.ptr(packed), _host.ptr(out), out.stride(0), m, n, k,
                                      ctypes.byref(e), _host.current_stream_ptr())
    _lib.check(rc, "gemm_k256_packed_bf16")
    return out


def gemm_packed_ln(a, packed, ln_gamma, ln_beta, ln_row_scale=None, eps=1e-5, bias=None, residual=None, row_scale=None,
                   alpha=1.0, act=_lib.ACT_NONE, out=None):
    """gemm_packed (N = 256, float32 out) + the LayerNorm behind it in one launch: returns (out, bf16 LayerNorm(out) * ln_row_scale)."""
    t = _host.torch()
    lib = _lib.load()
    assert a.dtype == t.bfloat16 and a.dim() == 2 and a.stride(1) == 1 and a.shape[1] == packed.shape[1]
    m, k = a.shape
    n = packed.shape[0]
    if out is None:
        out = t.empty((m, n), dtype=t.float32, device=a.device)
    assert out.dtype == t.float32 and out.stride(1) == 1 and tuple(out.shape) == (m, n)
    ln_out = t.empty((m, n), dtype=t.bfloat16, device=a.device)
    e = _epilogue(bias, residual, row_scale, alpha, act, False)
    rc = lib.ma_gemm_k256_packed_ln_bf16(_host.ptr(a), a.stride(0), _host.ptr(packed), _host.ptr(out), out.stride(0), m, n, k,
                                         ctypes.byref(e), _host.ptr(ln_gamma), _host.ptr(ln_beta), float(eps),
                                         _opt(ln_row_scale), _host.ptr(ln_out), ln_out.stride(0), _host.current_stream_ptr())
    _lib.check(rc, "gemm_k256_packed_ln_bf16")
    return out, ln_out


def ffn_pack_weights(w1, w2):
    """Fragment-ordered packed copy of (w1 (hidden, 256), w2 (256, hidden)) bf16 for ffn_packed; redo after a weight update."""
    t = _host.torch()
    lib = _lib.load()
    assert w1.dtype == t.bfloat16 and w2.dtype == t.bfloat16 and w1.is_contiguous() and w2.is_contiguous()
    hidden, d = w1.shape
    assert tuple(w2.shape) == (d, hidden)
    nbytes = lib.ma_ffn_packed_bytes(d, hidden)
    _lib.check(min(nbytes, 0), "ffn_packed_bytes")
    packed = t.empty((nbytes // 2,), dtype=t.bfloat16, device=w1.device)
    _lib.check(lib.ma_ffn_pack_weights_bf16(_host.ptr(w1), _host.ptr(w2), d, hidden, _host.ptr(packed),
                                            _host.current_stream_ptr()), "ffn_pack_weights_bf16")
    return packed


def ffn_packed(a, packed, b1, b2, x, g1=None, be1=None, g2=None, be2=None, alpha=0.5, eps=1e-5, out_dtype=None, ln_in=None):
    """ffn / ffn_ln on packed weights (ffn_pack_weights).  g1 None: x += alpha * FFN(a) in place, returns x.
    Otherwise as ffn_ln: returns the LayerNorm output (bf16 default).  ln_in = (gamma0, beta0): a is LayerNorm(x) computed
    inside the kernel (pass a=None)."""
    t = _host.torch()
    lib = _lib.load()
    assert x.dtype == t.float32 and x.stride(1) == 1
    if ln_in is None:
        assert a.dtype == t.bfloat16 and a.stride(1) == 1 and tuple(a.shape) == tuple(x.shape)
    m, d = x.shape
    hidden = b1.numel()
    mode = 0 if g1 is None else (2 if g2 is not None else 1)
    out_dtype = out_dtype or t.bfloat16
    out = t.empty((m, d), dtype=out_dtype, device=x.device) if mode else None
    rc = lib.ma_ffn_packed_bf16(_opt(a) if ln_in is None else None, a.stride(0) if ln_in is None else 0, _host.ptr(packed),
                                _host.ptr(b1), _host.ptr(b2), _host.ptr(x), x.stride(0), m, d, hidden, float(alpha), mode,
                                _opt(g1), _opt(be1), _opt(g2), _opt(be2), float(eps), _opt(out), out.stride(0) if mode else 0,
                                1 if out_dtype == t.bfloat16 else 0, _opt(ln_in[0]) if ln_in else None,
                                _opt(ln_in[1]) if ln_in else None, _host.current_stream_ptr())
    _lib.check(rc, "ffn_packed_bf16")
    return out if mode else x


def ffn_qkv_pack(w):
    """Packed copy of a dense weight w (N, 256) bf16 for the `qkv=` tail of ffn_packed / ffn_packed_pair; None if not covered."""
    t = _host.torch()
    lib = _lib.load()
    assert w.dtype == t.bfloat16 and w.dim() == 2 and w.stride(1) == 1
    n, k = w.shape
    if k != 256 or lib.ma_ffn_qkv_packed_bytes(n) < 0:
        return None
    packed = t.empty((n * k,), dtype=t.bfloat16, device=w.device)
    _lib.check(lib.ma_ffn_qkv_pack_bf16(_host.ptr(w), w.stride(0), n, _host.ptr(packed), _host.current_stream_ptr()),
               "ffn_qkv_pack_bf16")
    return packed


def ffn_packed_qkv(a, packed, b1, b2, x, g1, be1, qkv_packed, qkv_bias, alpha=0.5, eps=1e-5, ln_in=None):
    """x += alpha FFN(a) in place, then qkv = bf16(LN(x; g1, be1) @ Wq^T + qkv_bias) on the same tile; returns qkv (M, N).
    ln_in = (gamma0, beta0): a = LayerNorm(x; gamma0, beta0) computed inside the kernel (pass a=None)."""
    t = _host.torch()
    lib = _lib.load()
    assert x.dtype == t.float32 and x.stride(1) == 1
    if ln_in is None:
        assert a.dtype == t.bfloat16 and a.stride(1) == 1
    m, d = x.shape
    n = qkv_bias.numel()
    out = t.empty((m, n), dtype=t.bfloat16, device=x.device)
    rc = lib.ma_ffn_packed_qkv_bf16(_host.ptr(a) if ln_in is None else None, a.stride(0) if ln_in is None else 0,
                                    _host.ptr(packed), _host.ptr(b1), _host.ptr(b2), _host.ptr(x),
                                    x.stride(0), m, d, b1.numel(), float(alpha), _opt(ln_in[0]) if ln_in else None,
                                    _opt(ln_in[1]) if ln_in else None, _host.ptr(g1), _host.ptr(be1), float(eps),
                                    _host.ptr(qkv_packed), _host.ptr(qkv_bias), n, _host.ptr(out), out.stride(0),
                                    _host.current_stream_ptr())
    _lib.check(rc, "ffn_packed_qkv_bf16")
    return out


def ffn_packed_pair(packed_a, b1_a, b2_a, packed_b, b1_b, b2_b, x, ln_in, ln_mid, ln_next, ln_out, alpha=0.5, eps=1e-5, qkv=None):
    """Two position-wise FFNs on the same rows in one launch (the last FFN of a Conformer block and the macaron FFN of the next):
        x1 = x + alpha FFN_A(LN(x; ln_in));   x2 = LN(x1; ln_mid);   x <- x2 + alpha FFN_B(LN(x2; ln_next))   (in place)
    returns bf16 LN(x; ln_out) — or, with qkv = (packed weight from ffn_qkv_pack, bias), bf16(LN(x; ln_out) @ Wq^T + bias).
    Each ln_* is a (gamma, beta) pair; packed_* from ffn_pack_weights."""
    t = _host.torch()
    lib = _lib.load()
    assert x.dtype == t.float32 and x.stride(1) == 1
    m, d = x.shape
    if qkv is not None:
        n = qkv[1].numel()
        out = t.empty((m, n), dtype=t.bfloat16, device=x.device)
        rc = lib.ma_ffn_packed_pair_qkv_bf16(_host.ptr(packed_a), _host.ptr(b1_a), _host.ptr(b2_a), _host.ptr(packed_b),
                                             _host.ptr(b1_b), _host.ptr(b2_b), _host.ptr(x), x.stride(0), m, d, b1_a.numel(),
                                             float(alpha), _host.ptr(ln_in[0]), _host.ptr(ln_in[1]), _host.ptr(ln_mid[0]),
                                             _host.ptr(ln_mid[1]), _host.ptr(ln_next[0]), _host.ptr(ln_next[1]),
                                             _host.ptr(ln_out[0]), _host.ptr(ln_out[1]), float(eps), _host.ptr(qkv[0]),
                                             _host.ptr(qkv[1]), n, _host.ptr(out), out.stride(0), _host.current_stream_ptr())
        _lib.check(rc, "ffn_packed_pair_qkv_bf16")
        return out
    out = t.empty((m, d), dtype=t.bfloat16, device=x.device)
    rc = lib.ma_ffn_packed_pair_bf16(_host.ptr(packed_a), _host.ptr(b1_a), _host.ptr(b2_a), _host.ptr(packed_b), _host.ptr(b1_b),
                                     _host.ptr(b2_b), _host.ptr(x), x.stride(0), m, d, b1_a.numel(), float(alpha),
                                     _host.ptr(ln_in[0]), _host.ptr(ln_in[1]), _host.ptr(ln_mid[0]), _host.ptr(ln_mid[1]),
                                     _host.ptr(ln_next[0]), _host.ptr(ln_next[1]), _host.ptr(ln_out[0]), _host.ptr(ln_out[1]),
                                     float(eps), _host.ptr(out), out.stride(0), _host.current_stream_ptr())
    _lib.check(rc, "ffn_packed_pair_bf16")
    return out


def conv2d_3x3s2_nhwc(act, w, bias=None, relu=True, out_dtype=None):
    """act (B, H, W, C) bf16 NHWC, w (Cout, 3, 3, C) bf16 -> (B, Ho, Wo, Cout)."""
    t = _host.torch()
    lib = _lib.load()
    assert act.dtype == t.bfloat16 and w.dtype == t.bfloat16 and act.is_contiguous() and w.is_contiguous()
    b, h, wd, c = act.shape
    cout = w.shape[0]
    ho, wo = (h - 3) // 2 + 1, (wd - 3) // 2 + 1
    out_dtype = out_dtype or t.bfloat16
    out = t.empty((b, ho, wo, cout), dtype=out_dtype, device=act.device)
    e = _epilogue(bias, None, None, 1.0, _lib.ACT_RELU if relu else _lib.ACT_NONE, out_dtype == t.bfloat16)
    rc = lib.ma_conv2d_3x3s2_nhwc_bf16(_host.ptr(act), b, h, wd, c, _host.ptr(w), cout, _host.ptr(out),
                                       ctypes.byref(e), _host.current_stream_ptr())
    _lib.check(rc, "conv2d_3x3s2_nhwc_bf16")
    return out


def _opt(x):
    return _host.ptr(x) if x is not None else None


def layernorm(x, gamma, beta, eps=1e-5, row_scale=None, out_dtype=None, out=None):
    """x (rows, D) float32 -> LayerNorm(x) [* row_scale[:, None]] as bf16 (default) or float32."""
    t = _host.torch()
    lib = _lib.load()
    assert x.dtype == t.float32 and x.dim() == 2 and x.stride(1) == 1
    out_dtype = out_dtype or t.bfloat16
    if out is None:
        out = t.empty(x.shape, dtype=out_dtype, device=x.device)
    rc = lib.ma_layernorm_f32(_host.ptr(x), x.stride(0), x.shape[0], x.shape[1], _host.ptr(gamma), _host.ptr(beta),
                              float(eps), _opt(row_scale), _host.ptr(out), out.stride(0),
                              1 if out_dtype == t.bfloat16 else 0, _host.current_stream_ptr())
    _lib.check(rc, "layernorm")
    return out


def layernorm2(x, g1, b1, g2, b2, eps=1e-5, out2_dtype=None):
    """In place x <- LN(x; g1, b1) (float32) and returns LN(x_new; g2, b2) as bf16 (default) or float32."""
    t = _host.torch()
    lib = _lib.load()
    assert x.dtype == t.float32 and x.dim() == 2 and x.stride(1) == 1
    out2_dtype = out2_dtype or t.bfloat16
    out2 = t.empty(x.shape, dtype=out2_dtype, device=x.device)
    rc = lib.ma_layernorm2_f32(_host.ptr(x), x.stride(0), x.shape[0], x.shape[1], _host.ptr(g1), _host.ptr(b1),
                               _host.ptr(g2), _host.ptr(b2), float(eps), _host.ptr(x), x.stride(0), _host.ptr(out2),
                               out2.stride(0), 1 if out2_dtype == t.bfloat16 else 0, _host.current_stream_ptr())
    _lib.check(rc, "layernorm2")
    return out2


def subsample_conv1(x, w, bias, cmvn_mean=None, cmvn_istd=None, out=None):
    """x (B, T, idim) float32; w (C, 3, 3), bias (C) float32 -> NHWC bf16 (B, T1, F1, C) after CMVN, conv, ReLU."""
    t = _host.torch()
    lib = _lib.load()
    assert x.dtype == t.float32 and x.dim() == 3 and w.is_contiguous()  # x: any strides (e.g. a transposed fbank output)
    b, tt, idim = x.shape
    c = w.shape[0]
    shape = (b, (tt - 3) // 2 + 1, (idim - 3) // 2 + 1, c)
    if out is None:
        out = t.empty(shape, dtype=t.bfloat16, device=x.device)
    assert out.shape == shape and out.dtype == t.bfloat16 and out.is_contiguous()
    rc = lib.ma_subsample_conv1_strided_nhwc(_host.ptr(x), x.stride(0), x.stride(1), x.stride(2), b, tt, idim, _opt(cmvn_mean),
                                             _opt(cmvn_istd), _host.ptr(w), _host.ptr(bias), c, _host.ptr(out),
                                             _host.current_stream_ptr())
    _lib.check(rc, "subsample_conv1")
    return out


def subsample_fused_pack(w1, w2, idim):
    """Packed copy of the subsampling layer's two convolution weights for subsample_fused: w1 (C, 9) float32 (conv1, as bf16
    head + tail fragments), w2 (C, 3, 3, C) bf16 (conv2, fragment order); None if (idim, C) is not covered."""
    t = _host.torch()
    lib = _lib.load()
    assert w2.dtype == t.bfloat16 and w2.dim() == 4 and w2.is_contiguous() and w2.shape[1] == 3 and w2.shape[2] == 3
    c = w2.shape[0]
    nbytes = lib.ma_subsample_fused_packed_bytes(idim, c)
    if nbytes < 0 or w2.shape[3] != c:
        return None
    assert w1.dtype == t.float32 and w1.is_contiguous() and tuple(w1.shape) == (c, 9)
    packed = t.empty((nbytes // 2,), dtype=t.bfloat16, device=w2.device)
    _lib.check(lib.ma_subsample_fused_pack_bf16(_host.ptr(w1), _host.ptr(w2), idim, c, _host.ptr(packed),
                                                _host.current_stream_ptr()), "subsample_fused_pack_bf16")
    return packed


def subsample_fused(x, packed, b1, b2, cmvn_mean=None, cmvn_istd=None, out=None):
    """CMVN + both convolutions of Conv2dSubsampling4 in one launch: x (B, T, idim) float32 (any strides) -> NHWC bf16
    (B, T2, F2, C); packed from subsample_fused_pack, b1 / b2 (C) float32."""
    t = _host.torch()
    lib = _lib.load()
    assert x.dtype == t.float32 and x.dim() == 3 and b1.dtype == t.float32 and b2.dtype == t.float32
    b, tt, idim = x.shape
    c = b1.numel()
    t1, f1 = (tt - 3) // 2 + 1, (idim - 3) // 2 + 1
    shape = (b, (t1 - 3) // 2 + 1, (f1 - 3) // 2 + 1, c)
    if out is None:
        out = t.empty(shape, dtype=t.bfloat16, device=x.device)
    assert out.shape == shape and out.dtype == t.bfloat16 and out.is_contiguous()
    rc = lib.ma_subsample_fused_bf16(_host.ptr(x), x.stride(0), x.stride(1), x.stride(2), b, tt, idim, _opt(cmvn_mean),
                                     _opt(cmvn_istd), _host.ptr(packed), _host.ptr(b1), _host.ptr(b2), c, _host.ptr(out),
                                     _host.current_stream_ptr())
    _lib.check(rc, "subsample_fused")
    return out


def relpos_attention(qkv, pos, bias_u, bias_v, mask, batch, T, heads=4, d_k=64, out=None):
    """qkv (B*T, 768) bf16; pos (T, 256) bf16; bias_u/v (heads, d_k) f32; mask (B, T) f32, (B, T, T) f32 (a per-query mask: the
    streaming configuration's chunk masks, padding folded in) or None -> ctx (B*T, 256)."""
    t = _host.torch()
    lib = _lib.load()
    assert qkv.dtype == t.bfloat16 and pos.dtype == t.bfloat16 and qkv.stride(1) == 1 and pos.stride(1) == 1
    if out is None:
        out = t.empty((batch * T, heads * d_k), dtype=t.bfloat16, device=qkv.device)
    ws_bytes = lib.ma_relpos_attention_workspace_bytes(batch, T, heads, d_k)
    ws = t.empty(ws_bytes, dtype=t.uint8, device=qkv.device)
    if mask is not None and mask.dim() == 3:
        assert tuple(mask.shape) == (batch, T, T) and mask.dtype == t.float32 and mask.is_contiguous()
        rc = lib.ma_relpos_attention_qmask_bf16(_host.ptr(qkv), qkv.stride(0), _host.ptr(pos), pos.stride(0), _host.ptr(bias_u),
                                                _host.ptr(bias_v), _host.ptr(mask), batch, T, heads, d_k, _host.ptr(out),
                                                out.stride(0), _host.ptr(ws), ws_bytes, _host.current_stream_ptr())
        _lib.check(rc, "relpos_attention_qmask")
        return out
    rc = lib.ma_relpos_attention_bf16(_host.ptr(qkv), qkv.stride(0), _host.ptr(pos), pos.stride(0), _host.ptr(bias_u),
                                      _host.ptr(bias_v), _opt(mask), batch, T, heads, d_k, _host.ptr(out),
                                      out.stride(0), _host.ptr(ws), ws_bytes, _host.current_stream_ptr())
    _lib.check(rc, "relpos_attention")
    return out


def convmodule_mid(y, dw, bn_scale, bn_shift, batch, T, out=None):
    """y (B*T, 2C) bf16 -> swish(bn(depthwise(glu(y)))) as bf16 (B*T, C)."""
    t = _host.torch()
    lib = _lib.load()
    c, ks = dw.shape
    assert y.dtype == t.bfloat16 and y.shape[1] == 2 * c and y.stride(1) == 1
    if out is None:
        out = t.empty((batch * T, c), dtype=t.bfloat16, device=y.device)
    rc = lib.ma_convmodule_mid_bf16(_host.ptr(y), y.stride(0), batch, T, c, _host.ptr(dw), ks, _host.ptr(bn_scale),
                                    _host.ptr(bn_shift), _host.ptr(out), out.stride(0), _host.current_stream_ptr())
    _lib.check(rc, "convmodule_mid")
    return out


def convmid_pw2(y, dw, bn_scale, bn_shift, pw2_packed, pw2_bias, mask_rows, x, batch, T):
    """In place x += mask * (convmodule_mid(y) @ Wp2^T + bias): conv-module middle + pointwise_conv2 + residual in one launch."""
    t = _host.torch()
    lib = _lib.load()
    c, ks = dw.shape
    assert y.dtype == t.bfloat16 and y.shape[1] == 2 * c and y.stride(1) == 1 and x.dtype == t.float32 and x.stride(1) == 1
    rc = lib.ma_convmid_pw2_bf16(_host.ptr(y), y.stride(0), batch, T, c, _host.ptr(dw), ks, _host.ptr(bn_scale),
                                 _host.ptr(bn_shift), _host.ptr(pw2_packed), _host.ptr(pw2_bias), _opt(mask_rows), _host.ptr(x),
                                 x.stride(0), _host.current_stream_ptr())
    _lib.check(rc, "convmid_pw2")
    return x


def convmodule(a, pw1_packed, pw1_bias, dw, bn_scale, bn_shift, pw2_packed, pw2_bias, mask_rows, x, batch, T):
    """In place x += mask * ConvolutionModule_after_LayerNorm(a): pointwise_conv1 + GLU + depthwise + BN + Swish + pointwise_conv2
    + residual in one launch.  a (B*T, 256) bf16 = norm_conv(x) * mask; packed weights from gemm_k256_pack."""
    t = _host.torch()
    lib = _lib.load()
    c, ks = dw.shape
    assert a.dtype == t.bfloat16 and a.shape[1] == c and a.stride(1) == 1 and x.dtype == t.float32 and x.stride(1) == 1
    rc = lib.ma_convmodule_bf16(_host.ptr(a), a.stride(0), batch, T, c, _host.ptr(pw1_packed), _host.ptr(pw1_bias), _host.ptr(dw),
                                ks, _host.ptr(bn_scale), _host.ptr(bn_shift), _host.ptr(pw2_packed), _host.ptr(pw2_bias),
                                _opt(mask_rows), _host.ptr(x), x.stride(0), _host.current_stream_ptr())
    _lib.check(rc, "convmodule")
    return x


def attn_out_convmodule(ctx, wo_packed, wo_bias, ln_gamma, ln_beta, pw1_packed, pw1_bias, dw, bn_scale, bn_shift, pw2_packed,
                        pw2_bias, mask_rows, x, batch, T, eps=1e-5, out=None):
    """out <- x' + mask * ConvModule(LN(x') * mask) with x' = x + ctx @ Wo^T + wo_bias: the attention output projection,
    norm_conv and the whole ConvolutionModule in one launch.  ctx (B*T, 256) bf16.  `out` is another buffer of x's shape (a new one
    if None), never x itself: a tile reads its neighbours' residual rows (see include/mindaudio_amd.h)."""
    t = _host.torch()
    lib = _lib.load()
    c, ks = dw.shape
    assert ctx.dtype == t.bfloat16 and ctx.shape[1] == c and ctx.stride(1) == 1 and x.dtype == t.float32 and x.stride(1) == 1
    if out is None:
        out = t.empty_like(x)
    assert out.dtype == t.float32 and out.shape == x.shape and out.stride() == x.stride()
    rc = lib.ma_attn_out_convmodule_bf16(_host.ptr(ctx), ctx.stride(0), _host.ptr(wo_packed), _host.ptr(wo_bias),
                                         _host.ptr(ln_gamma), _host.ptr(ln_beta), float(eps), batch, T, c, _host.ptr(pw1_packed),
                                         _host.ptr(pw1_bias), _host.ptr(dw), ks, _host.ptr(bn_scale), _host.ptr(bn_shift),
                                         _host.ptr(pw2_packed), _host.ptr(pw2_bias), _opt(mask_rows), _host.ptr(x), _host.ptr(out),
                                         x.stride(0), _host.current_stream_ptr())
    _lib.check(rc, "attn_out_convmodule")
    return out


def cast_bf16(x):
    """float32 device tensor -> bf16 copy (round to nearest even)."""
    t = _host.torch()
    lib = _lib.load()
    assert x.dtype == t.float32 and x.is_contiguous() and x.numel() % 4 == 0
    y = t.empty(x.shape, dtype=t.bfloat16, device=x.device)
    _lib.check(lib.ma_cast_f32_bf16(_host.ptr(x), _host.ptr(y), x.numel(), _host.current_stream_ptr()), "cast_bf16")
    return y


def ctc_loss(logits, batch, T, ys_pad, hlens, ys_lens, blank=0, zero_infinity=True):
    """logits (B*T, V) float32; ys_pad (B, Lmax) int32; hlens, ys_lens (B,) int32.
    Returns (loss scalar tensor = sum(per-utterance CTC) / B, per-utterance losses)."""
    t = _host.torch()
    lib = _lib.load()
    assert logits.dtype == t.float32 and logits.stride(1) == 1 and logits.shape[0] == batch * T
    ys_pad = ys_pad.to(t.int32).contiguous()
    hlens = hlens.to(t.int32).contiguous()
    ys_lens = ys_lens.to(t.int32).contiguous()
    per = t.empty(batch, dtype=t.float32, device=logits.device)
    lse = t.empty(batch * T, dtype=t.float32, device=logits.device)
    out = t.empty(1, dtype=t.float32, device=logits.device)
    rc = lib.ma_ctc_loss_f32(_host.ptr(logits), logits.stride(0), batch, T, logits.shape[1], _host.ptr(ys_pad),
                             ys_pad.shape[1], _host.ptr(hlens), _host.ptr(ys_lens), blank, 1 if zero_infinity else 0,
                             _host.ptr(per), _host.ptr(lse), _host.ptr(out), _host.current_stream_ptr())
    _lib.check(rc, "ctc_loss")
    return out[0], per


def ctc_greedy_search(logits, batch, T, V, mask=None, blank=0):
    """logits (B*T, ld >= V) float32 -> (best (B, T) int32 masked, best_logp (B, T) f32, hyp (B, T) int32, hyp_len (B,))."""
    t = _host.torch()
    lib = _lib.load()
    dev = logits.device
    best = t.empty((batch, T), dtype=t.int32, device=dev)
    logp = t.empty((batch, T), dtype=t.float32, device=dev)
    hyp = t.empty((batch, T), dtype=t.int32, device=dev)
    hyp_len = t.empty((batch,), dtype=t.int32, device=dev)
    rc = lib.ma_ctc_greedy_search_f32(_host.ptr(logits), logits.stride(0), batch, T, V, _opt(mask), blank, _host.ptr(best),
                                      _host.ptr(logp), _host.ptr(hyp), _host.ptr(hyp_len), _host.current_stream_ptr())
    _lib.check(rc, "ctc_greedy_search")
    return best, logp, hyp, hyp_len
