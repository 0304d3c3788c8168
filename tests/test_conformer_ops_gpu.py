"""GPU parity of the Conformer building-block kernels against plain PyTorch float32/float64 references of the
same op computed on the SAME bf16-rounded inputs (so the only differences are accumulation order and the
final rounding): float32 outputs within 2e-5 of the output scale, bf16 outputs within one bf16 ulp (2^-8)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def t():
    import torch

    assert torch.cuda.is_available()
    return torch


def _rand(t, *shape, seed=0, scale=1.0):
    g = t.Generator(device="cpu").manual_seed(seed)
    return (t.randn(*shape, generator=g) * scale)


# (the last three run on the 256 x 256 8-phase kernel: ragged M and N edges, an odd leading dimension, a long contraction)
@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (250, 256, 256), (7968, 2048, 256), (1000, 256, 2048),
                                   (333, 4233, 256), (513, 768, 256), (64, 256, 4864), (4133, 3850, 1088),
                                   (8192, 1024, 1024), (2049, 8200, 2048)])
def test_gemm_plain_and_transpose_detecting(t, m, n, k):
    from mindaudio_amd import ops

    a = _rand(t, m, k, seed=1).bfloat16().cuda()
    w = _rand(t, n, k, seed=2, scale=1.0 / math.sqrt(k)).bfloat16().cuda()
    bias = _rand(t, n, seed=3).cuda()
    ref = a.double() @ w.double().T + bias.double()
    out32 = ops.gemm(a, w, bias=bias, out_dtype=t.float32)
    scale = float(ref.abs().max())
    assert float((out32.double() - ref).abs().max()) <= 2e-5 * scale
    out16 = ops.gemm(a, w, bias=bias)
    assert out16.dtype == t.bfloat16
    assert float((out16.double() - ref).abs().max()) <= 2 ** -8 * scale * 1.01


def test_gemm_epilogues(t):
    from mindaudio_amd import _lib, ops

    m, n, k = 777, 512, 256
    a = _rand(t, m, k, seed=4).bfloat16().cuda()
    w = _rand(t, n, k, seed=5, scale=0.06).bfloat16().cuda()
    bias = _rand(t, n, seed=6).cuda()
    res = _rand(t, m, n, seed=7).cuda()
    rs = (t.rand(m, generator=t.Generator().manual_seed(8)) > 0.3).float().cuda()
    z = a.double() @ w.double().T + bias.double()
    # swish -> bf16 (FFN w_1), relu, residual + 0.5 * (...) in f32 (FFN w_2), row mask (conv module)
    got = ops.gemm(a, w, bias=bias, act=_lib.ACT_SWISH).double()
    want = z * t.sigmoid(z)
    assert float((got - want).abs().max()) <= 2 ** -8 * float(want.abs().max()) * 1.01
    got = ops.gemm(a, w, bias=bias, act=_lib.ACT_RELU, out_dtype=t.float32).double()
    assert float((got - z.clamp(min=0)).abs().max()) <= 2e-5 * float(z.abs().max())
    got = ops.gemm(a, w, bias=bias, residual=res, alpha=0.5, out_dtype=t.float32).double()
    want = res.double() + 0.5 * z
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    got = ops.gemm(a, w, bias=bias, residual=res, row_scale=rs, out_dtype=t.float32).double()
    want = res.double() + z * rs.double()[:, None]
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    # strided A (a column slice of a wider buffer) and in-place residual update
    wide = _rand(t, m, 3 * k, seed=9).bfloat16().cuda()
    got = ops.gemm(wide[:, k:2 * k], w, out_dtype=t.float32).double()
    want = wide[:, k:2 * k].double() @ w.double().T
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    x = res.clone()
    ops.gemm(a, w, bias=bias, residual=x, alpha=0.5, out_dtype=t.float32, out=x)
    assert float((x.double() - (res.double() + 0.5 * z)).abs().max()) <= 2e-5 * float(z.abs().max())


@pytest.mark.parametrize("m,n", [(16384, 512), (80896, 512), (20011, 128), (16390, 1024)])
def test_gemm_k512_weight_stationary(t, m, n):
    """ma_gemm_bf16 with K = 512, bf16 output, >= 16 384 rows runs the weight-stationary persistent kernel (gemm_ws512_kernel: ECAPA's
    1 x 1 convolutions at C = 512): plain, and with the whole ECAPA epilogue (bias -> ReLU -> BatchNorm affine -> tanh -> row scale);
    ragged last row tile, strided A and output, 1 / 4 / 8 column blocks."""
    import ctypes

    from mindaudio_amd import _host, _lib, ops

    k = 512
    a = _rand(t, m, k, seed=21).bfloat16().cuda()
    w = _rand(t, n, k, seed=22, scale=1.0 / math.sqrt(k)).bfloat16().cuda()
    bias = _rand(t, n, seed=23).cuda()
    z = a.double() @ w.double().T + bias.double()
    got = ops.gemm(a, w, bias=bias)
    assert float((got.double() - z).abs().max()) <= 2 ** -8 * float(z.abs().max()) * 1.01
    # the ECAPA epilogue through the C-ABI, A and out as column slices of wider buffers
    wide = _rand(t, m, k + 64, seed=24).bfloat16().cuda()
    av = wide[:, 64:]
    cs, ct = (1 + 0.1 * _rand(t, n, seed=25)).cuda(), (0.1 * _rand(t, n, seed=26)).cuda()
    rs = (t.rand(m, generator=t.Generator().manual_seed(27)) > 0.1).float().cuda()
    out = t.zeros(m, n + 8, dtype=t.bfloat16, device="cuda")
    e = _lib.GemmEpilogue()
    e.bias, e.row_scale, e.col_scale, e.col_shift = bias.data_ptr(), rs.data_ptr(), cs.data_ptr(), ct.data_ptr()
    e.alpha, e.act, e.act2, e.out_bf16 = 1.0, _lib.ACT_RELU, _lib.ACT_TANH, 1
    rc = _lib.load().ma_gemm_bf16(_host.ptr(av), av.stride(0), _host.ptr(w), w.stride(0), _host.ptr(out), out.stride(0), m, n, k,
                                  ctypes.byref(e), _host.current_stream_ptr())
    assert rc == 0
    z = av.double() @ w.double().T + bias.double()
    want = t.tanh(z.clamp(min=0) * cs.double() + ct.double()) * rs.double()[:, None]
    assert float((out[:, :n].double() - want).abs().max()) <= 2 ** -8 * 1.01 + 1e-3
    assert float(out[:, n:].abs().max()) == 0.0


@pytest.mark.parametrize("b,h,wd", [(2, 21, 9), (3, 99, 39), (1, 7, 5), (5, 131, 39)])
def test_conv2d_3x3s2_packed(t, b, h, wd):
    """Subsampling conv 2 on fragment-packed weights: same result as the general implicit-GEMM kernel."""
    from mindaudio_amd import ops

    c = cout = 256
    x = _rand(t, b, c, h, wd, seed=21).bfloat16()
    w = _rand(t, cout, c, 3, 3, seed=22, scale=1.0 / math.sqrt(9 * c)).bfloat16()
    bias = _rand(t, cout, seed=23)
    act = x.permute(0, 2, 3, 1).contiguous().cuda()
    wk = w.permute(0, 2, 3, 1).contiguous().cuda()
    pk = ops.conv2d_3x3s2_pack(wk)
    assert pk is not None and t.equal(pk.view(t.int16).sort().values, wk.view(t.int16).flatten().sort().values)
    for relu in (True, False):
        got = ops.conv2d_3x3s2_packed(act, pk, bias.cuda(), relu=relu)
        ref = t.nn.functional.conv2d(x.double(), w.double(), bias.double(), stride=2)
        if relu:
            ref = t.nn.functional.relu(ref)
        gd = got.permute(0, 3, 1, 2).double().cpu()
        assert gd.shape == ref.shape
        assert float((gd - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max()) * 1.01
        # and bit-identical to the general kernel (same k order inside every accumulation chain)
        assert t.equal(got, ops.conv2d_3x3s2_nhwc(act, wk, bias=bias.cuda(), relu=relu))
    assert ops.conv2d_3x3s2_pack(wk[:, :, :, :128].contiguous()) is None


@pytest.mark.parametrize("b,tt,ks", [(3, 249, 15), (2, 33, 7), (5, 64, 15), (1, 5, 3)])
def test_convmid_pw2_fused(t, b, tt, ks):
    """conv-module middle + pointwise_conv2 + mask + residual in one launch == the two-kernel path."""
    from mindaudio_amd import ops

    c = 256
    y = _rand(t, b * tt, 2 * c, seed=31).bfloat16().cuda()
    dw = _rand(t, c, ks, seed=32, scale=0.3).cuda()
    sc, sh = (1 + 0.1 * _rand(t, c, seed=33)).cuda(), (0.1 * _rand(t, c, seed=34)).cuda()
    w2 = _rand(t, c, c, seed=35, scale=1.0 / 16).bfloat16().cuda()
    b2 = _rand(t, c, seed=36).cuda()
    mask = (t.rand(b * tt, generator=t.Generator().manual_seed(37)) > 0.2).float().cuda()
    x0 = _rand(t, b * tt, c, seed=38).cuda()
    pk = ops.gemm_k256_pack(w2)
    z = ops.convmodule_mid(y, dw, sc, sh, b, tt)
    want = x0.clone()
    ops.gemm(z, w2, bias=b2, row_scale=mask, residual=want, out_dtype=t.float32, out=want)
    got = x0.clone()
    assert ops.convmid_pw2(y, dw, sc, sh, pk, b2, mask, got, b, tt) is got
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    got2 = x0.clone()
    ops.convmid_pw2(y, dw, sc, sh, pk, b2, None, got2, b, tt)
    want2 = x0.clone()
    ops.gemm(z, w2, bias=b2, residual=want2, out_dtype=t.float32, out=want2)
    assert float((got2 - want2).abs().max()) <= 1e-5 * float(want2.abs().max())


@pytest.mark.parametrize("b,tt,ks", [(3, 249, 15), (2, 33, 7), (5, 64, 15), (1, 5, 3), (2, 32, 15)])
def test_convmodule_one_launch(t, b, tt, ks):
    """pointwise_conv1 + GLU + depthwise + BN + Swish + pointwise_conv2 + mask + residual in one launch, against a float64
    evaluation of layers/convolution.py:96-127 on the same bf16 inputs/weights (the GLU here works on the float32 accumulators
    and is rounded to bf16 once; the z tile is bf16 as in the two-kernel path: tolerance = a few bf16 ulps of the branch)."""
    from mindaudio_amd import ops

    c = 256
    a = _rand(t, b * tt, c, seed=41).bfloat16()
    w1 = _rand(t, 2 * c, c, seed=42, scale=1.0 / 16).bfloat16()
    b1 = _rand(t, 2 * c, seed=43, scale=0.2)
    dw = _rand(t, c, ks, seed=32, scale=0.3)
    sc, sh = 1 + 0.1 * _rand(t, c, seed=33), 0.1 * _rand(t, c, seed=34)
    w2 = _rand(t, c, c, seed=35, scale=1.0 / 16).bfloat16()
    b2 = _rand(t, c, seed=36)
    mask = (t.rand(b * tt, generator=t.Generator().manual_seed(37)) > 0.2).float()
    x0 = _rand(t, b * tt, c, seed=38)
    # float64 reference
    y = a.double() @ w1.double().T + b1.double()
    y = (y[:, :c] * t.sigmoid(y[:, c:])).view(b, tt, c).transpose(1, 2)                       # (B, C, T)
    z = t.nn.functional.conv1d(y, dw.double().unsqueeze(1), padding=ks // 2, groups=c)
    z = z * sc.double()[None, :, None] + sh.double()[None, :, None]
    z = (z * t.sigmoid(z)).transpose(1, 2).reshape(b * tt, c)
    branch = z @ w2.double().T + b2.double()
    for m_ in (mask, None):
        want = x0.double() + (branch * m_.double()[:, None] if m_ is not None else branch)
        got = x0.clone().cuda()
        out = ops.convmodule(a.cuda(), ops.gemm_k256_pack(w1.cuda()), b1.cuda(), dw.cuda(), sc.cuda(), sh.cuda(),
                             ops.gemm_k256_pack(w2.cuda()), b2.cuda(), m_.cuda() if m_ is not None else None, got, b, tt)
        assert out is got
        err = (got.double().cpu() - want).abs().max()
        assert float(err) <= 2 ** -7 * float(branch.abs().max()), float(err)


@pytest.mark.parametrize("b,tt,ks", [(3, 249, 15), (2, 33, 7), (5, 64, 15), (1, 5, 3), (2, 32, 15), (150, 224, 15)])
def test_attn_out_convmodule_one_launch(t, b, tt, ks):
    """linear_out + residual + norm_conv + ConvolutionModule in one launch == output projection with the LayerNorm epilogue
    (ops.gemm_packed_ln) followed by ops.convmodule (same arithmetic per row; halo frames recomputed).  Out of place: a tile reads the
    residual rows of its halo frames, i.e. its neighbours' rows - the (150, 224) case is 1 050 tiles of 7 per utterance, more than the
    chip holds at once, where the in-place form of rounds 2-4 let late tiles read rows their neighbours had already updated."""
    from mindaudio_amd import ops

    c = 256
    ctx = _rand(t, b * tt, c, seed=51).bfloat16().cuda()
    wo = _rand(t, c, c, seed=52, scale=1.0 / 16).bfloat16().cuda()
    bo = _rand(t, c, seed=53, scale=0.2).cuda()
    lg, lb = (1 + 0.1 * _rand(t, c, seed=54)).cuda(), (0.1 * _rand(t, c, seed=55)).cuda()
    w1 = _rand(t, 2 * c, c, seed=42, scale=1.0 / 16).bfloat16().cuda()
    b1 = _rand(t, 2 * c, seed=43, scale=0.2).cuda()
    dw = _rand(t, c, ks, seed=32, scale=0.3).cuda()
    sc, sh = (1 + 0.1 * _rand(t, c, seed=33)).cuda(), (0.1 * _rand(t, c, seed=34)).cuda()
    w2 = _rand(t, c, c, seed=35, scale=1.0 / 16).bfloat16().cuda()
    b2 = _rand(t, c, seed=36).cuda()
    mask = (t.rand(b * tt, generator=t.Generator().manual_seed(37)) > 0.2).float().cuda()
    x0 = _rand(t, b * tt, c, seed=38).cuda()
    po, p1, p2 = ops.gemm_k256_pack(wo), ops.gemm_k256_pack(w1), ops.gemm_k256_pack(w2)
    for m_ in (mask, None):
        want = x0.clone()
        _, a = ops.gemm_packed_ln(ctx, po, lg, lb, ln_row_scale=m_, bias=bo, residual=want, out=want)
        ops.convmodule(a, p1, b1, dw, sc, sh, p2, b2, m_, want, b, tt)
        xin, buf = x0.clone(), t.full_like(x0, float("nan"))
        got = ops.attn_out_convmodule(ctx, po, bo, lg, lb, p1, b1, dw, sc, sh, p2, b2, m_, xin, b, tt, out=buf)
        assert got is buf and t.equal(xin, x0)
        d, top = (got - want).abs(), float(want.abs().max())
        # same arithmetic per row; in a large case a few elements of the bf16 tile a = LN(x') round the other way (x' is summed in
        # another order) and move the rows that convolve them by one bf16 step of a: bounded, and rare
        assert float(d.max()) <= (1e-5 if d.numel() < 1 << 20 else 1e-3) * top, float(d.max())
        assert int((d > 1e-5 * top).sum()) <= 1e-3 * d.numel()  # (one flipped element of a reaches 15 frames x 256 channels)
    with pytest.raises(ValueError):  # x_out overlapping x (MA_ERR_INVALID_ARG): refused, not raced
        ops.attn_out_convmodule(ctx, po, bo, lg, lb, p1, b1, dw, sc, sh, p2, b2, None, xin, b, tt, out=xin)


@pytest.mark.parametrize("m,n", [(64, 256), (777, 512), (15936, 768), (1, 256), (130, 1024)])
def test_gemm_k256_packed(t, m, n):
    """K = 256 dense layers on fragment-packed weights: same contract (and epilogues) as ops.gemm."""
    from mindaudio_amd import _lib, ops

    k = 256
    a = _rand(t, m, k, seed=14).bfloat16().cuda()
    w = _rand(t, n, k, seed=15, scale=0.06).bfloat16().cuda()
    bias = _rand(t, n, seed=16).cuda()
    res = _rand(t, m, n, seed=17).cuda()
    rs = (t.rand(m, generator=t.Generator().manual_seed(18)) > 0.3).float().cuda()
    pk = ops.gemm_k256_pack(w)
    assert pk is not None and t.equal(pk.view(t.int16).flatten().sort().values, w.view(t.int16).flatten().sort().values)
    assert ops.gemm_k256_pack(w[:, :128].contiguous()) is None and ops.gemm_k256_pack(w[:100]) is None
    z = a.double() @ w.double().T + bias.double()
    got = ops.gemm_packed(a, pk, bias=bias, out_dtype=t.float32).double()
    assert float((got - z).abs().max()) <= 2e-5 * float(z.abs().max())
    # bit-identical to the general kernel?  both accumulate k in the same order inside one MFMA chain per output
    assert t.equal(ops.gemm_packed(a, pk, bias=bias), ops.gemm(a, w, bias=bias))
    got = ops.gemm_packed(a, pk, bias=bias, act=_lib.ACT_SWISH).double()
    want = z * t.sigmoid(z)
    assert float((got - want).abs().max()) <= 2 ** -8 * float(want.abs().max()) * 1.01
    got = ops.gemm_packed(a, pk, bias=bias, act=_lib.ACT_RELU, out_dtype=t.float32).double()
    assert float((got - z.clamp(min=0)).abs().max()) <= 2e-5 * float(z.abs().max())
    got = ops.gemm_packed(a, pk, bias=bias, residual=res, row_scale=rs, alpha=0.5, out_dtype=t.float32).double()
    want = res.double() + 0.5 * z * rs.double()[:, None]
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    if n == 256:  # LayerNorm of the output rows fused behind the epilogue
        gam, bet = (1 + 0.1 * _rand(t, n, seed=24)).cuda(), (0.1 * _rand(t, n, seed=25)).cuda()
        x2 = res.clone()
        o, a_ln = ops.gemm_packed_ln(a, pk, gam, bet, ln_row_scale=rs, bias=bias, residual=x2, out=x2)
        want_o = ops.gemm_packed(a, pk, bias=bias, residual=res, out_dtype=t.float32)
        assert o is x2 and t.equal(o, want_o)
        want_ln = ops.layernorm(want_o, gam, bet, row_scale=rs).float()
        assert a_ln.dtype == t.bfloat16 and float((a_ln.float() - want_ln).abs().max()) <= 2 ** -7 * float(want_ln.abs().max())
    wide = _rand(t, m, 3 * k, seed=19).bfloat16().cuda()  # strided A, in-place residual
    x = res.clone()
    ops.gemm_packed(wide[:, k:2 * k], pk, residual=x, out_dtype=t.float32, out=x)
    want = res.double() + wide[:, k:2 * k].double() @ w.double().T
    assert float((x.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("m,k", [(15936, 4864), (1000, 192), (77, 1024), (130, 448), (64, 64)])
def test_gemm_rows_packed(t, m, k):
    """The embed layer's Dense(19 * 256 -> 256) * sqrt(d) on a fragment-packed weight: same numbers as ops.gemm."""
    from mindaudio_amd import ops

    n = 256
    a = _rand(t, m, k, seed=51).bfloat16().cuda()
    w = _rand(t, n, k, seed=52, scale=1.0 / math.sqrt(k)).bfloat16().cuda()
    bias = _rand(t, n, seed=53).cuda()
    pk = ops.gemm_rows_pack(w)
    kpad = (k // 64 + 2) // 3 * 192
    assert pk is not None and tuple(pk.shape) == (n, kpad)
    assert float(pk.double().abs().sum()) == float(w.double().abs().sum())  # a permutation of W plus zero padding
    assert ops.gemm_rows_pack(w[:128].contiguous()) is None and ops.gemm_rows_pack(w[:, :32].contiguous()) is None
    ref = (a.double() @ w.double().T + bias.double()) * 16.0
    got = ops.gemm_rows_packed(a, pk, bias, alpha=16.0)
    assert got.dtype == t.float32
    assert float((got.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # strided A (rows of a wider buffer) and a strided output
    wide = _rand(t, m, k + 64, seed=54).bfloat16().cuda()
    outw = t.zeros(m, n + 32, device="cuda")
    ops.gemm_rows_packed(wide[:, 64:], pk, bias, out=outw[:, :n])
    want = wide[:, 64:].double() @ w.double().T + bias.double()
    assert float((outw[:, :n].double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert float(outw[:, n:].abs().max()) == 0.0


def test_gemm_rejects_bad_shapes(t):
    from mindaudio_amd import ops

    a = t.zeros(16, 100, dtype=t.bfloat16, device="cuda")
    w = t.zeros(16, 100, dtype=t.bfloat16, device="cuda")
    with pytest.raises(NotImplementedError):
        ops.gemm(a, w)  # K % 64 != 0


@pytest.mark.parametrize("b,h,wd,c,cout", [(2, 21, 9, 64, 128), (3, 99, 39, 256, 256)])
def test_conv2d_3x3s2_implicit_gemm(t, b, h, wd, c, cout):
    from mindaudio_amd import ops

    x = _rand(t, b, c, h, wd, seed=11).bfloat16()
    w = _rand(t, cout, c, 3, 3, seed=12, scale=1.0 / math.sqrt(9 * c)).bfloat16()
    bias = _rand(t, cout, seed=13)
    ref = t.nn.functional.relu(t.nn.functional.conv2d(x.double(), w.double(), bias.double(), stride=2))
    got = ops.conv2d_3x3s2_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(0, 2, 3, 1).contiguous().cuda(),
                                bias=bias.cuda(), relu=True, out_dtype=t.float32)
    got = got.permute(0, 3, 1, 2).double().cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_layernorm(t):
    from mindaudio_amd import ops

    x = (_rand(t, 1001, 256, seed=20) * 3 + 0.7).cuda()
    g = _rand(t, 256, seed=21).cuda()
    bta = _rand(t, 256, seed=22).cuda()
    rs = (t.rand(1001, generator=t.Generator().manual_seed(23)) > 0.2).float().cuda()
    xd = x.double()
    mean = xd.mean(-1, keepdim=True)
    var = ((xd - mean) ** 2).mean(-1, keepdim=True)
    ref = (xd - mean) / t.sqrt(var + 1e-5) * g.double() + bta.double()
    got = ops.layernorm(x, g, bta, out_dtype=t.float32).double()
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    got = ops.layernorm(x, g, bta, row_scale=rs).double()
    ref2 = ref * rs.double()[:, None]
    assert float((got - ref2).abs().max()) <= 2 ** -8 * float(ref2.abs().max()) * 1.01


# C = 256 runs the 8-rows-per-workgroup kernel (ragged last group, odd idim, both input layouts); C = 64 the general one
@pytest.mark.parametrize("tt,idim,c,cmvn,transposed", [(103, 80, 256, True, False), (103, 80, 256, False, True),
                                                        (20, 23, 256, True, True), (3, 3, 256, False, False),
                                                        (35, 80, 64, True, False)])
def test_subsample_conv1(t, tt, idim, c, cmvn, transposed):
    from mindaudio_amd import ops

    x = _rand(t, 3, tt, idim, seed=30)
    w = _rand(t, c, 1, 3, 3, seed=31, scale=0.3)
    bias = _rand(t, c, seed=32, scale=0.1)
    mean = _rand(t, idim, seed=33)
    istd = t.rand(idim, generator=t.Generator().manual_seed(34)) + 0.5
    xin = (x - mean) * istd if cmvn else x
    ref = t.nn.functional.relu(t.nn.functional.conv2d(xin.double().unsqueeze(1), w.double(), bias.double(), stride=2))
    xd = x.cuda()
    if transposed:  # the (B, n_mels, T) fbank output viewed as (B, T, n_mels)
        xd = xd.transpose(1, 2).contiguous().transpose(1, 2)
        assert xd.stride(1) == 1
    got = ops.subsample_conv1(xd, w.reshape(c, 9).contiguous().cuda(), bias.cuda(), mean.cuda() if cmvn else None,
                              istd.cuda() if cmvn else None)
    got = got.permute(0, 3, 1, 2).double().cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max()) * 1.01


# T = 1000: the north-star utterance (37 tiles of 128 positions, the last one ragged); 103 / 41: ragged last tiles with few output rows;
# 7: one output row, one tile; both input layouts, with and without CMVN
# (12, 1000): 12 x 74 tiles = 888 workgroups, more than three resident rounds
@pytest.mark.parametrize("b,tt,cmvn,transposed", [(2, 1000, True, True), (3, 103, True, False), (2, 41, False, True),
                                                   (1, 7, True, False), (5, 300, False, False), (12, 1000, True, False)])
def test_subsample_fused(t, b, tt, cmvn, transposed):
    """CMVN + conv1 + ReLU + conv2 + ReLU in one launch (act1 only in LDS, computed on the matrix pipe from bf16 head + tail splits)
    against the float64 reference and against the two-kernel path it replaces (act1 from a float32 FMA chain; another summation
    order in conv2)."""
    from mindaudio_amd import ops

    idim, c = 80, 256
    x = _rand(t, b, tt, idim, seed=40)
    w1 = _rand(t, c, 1, 3, 3, seed=41, scale=0.3)
    b1 = _rand(t, c, seed=42, scale=0.1)
    w2 = _rand(t, c, c, 3, 3, seed=43, scale=1.0 / math.sqrt(9 * c)).bfloat16()
    b2 = _rand(t, c, seed=44, scale=0.1)
    mean = _rand(t, idim, seed=45)
    istd = t.rand(idim, generator=t.Generator().manual_seed(46)) + 0.5
    xd = x.cuda()
    if transposed:  # the (B, n_mels, T) fbank output viewed as (B, T, n_mels)
        xd = xd.transpose(1, 2).contiguous().transpose(1, 2)
        assert xd.stride(1) == 1
    w1d, b1d, b2d = w1.reshape(c, 9).contiguous().cuda(), b1.cuda(), b2.cuda()
    w2k = w2.permute(0, 2, 3, 1).contiguous().cuda()  # (Cout, 3, 3, C)
    md, sd = (mean.cuda(), istd.cuda()) if cmvn else (None, None)
    pk = ops.subsample_fused_pack(w1d, w2k, idim)
    n2 = w2k.numel()
    assert pk is not None and t.equal(pk[:n2].view(t.int16).sort().values, w2k.view(t.int16).flatten().sort().values)
    assert ops.subsample_fused_pack(w1d, w2k, 40) is None
    got = ops.subsample_fused(xd, pk, b1d, b2d, md, sd)
    # the two kernels it replaces
    act1 = ops.subsample_conv1(xd, w1d, b1d, md, sd)
    two = ops.conv2d_3x3s2_packed(act1, ops.conv2d_3x3s2_pack(w2k), b2d, relu=True)
    assert got.shape == two.shape and got.dtype == t.bfloat16
    ref = t.nn.functional.relu(t.nn.functional.conv2d(act1.permute(0, 3, 1, 2).double().cpu(), w2.double(), b2.double(), stride=2))
    scale = float(ref.abs().max())
    gd = got.permute(0, 3, 1, 2).double().cpu()
    assert gd.shape == ref.shape
    # (ref uses the stand-alone kernel's act1: about one act1 element in 200 differs from the fused kernel's by one bf16 ulp)
    assert float((gd - ref).abs().max()) <= 2 ** -7 * scale
    # bf16 results of two float32 summation orders: equal except where the sums straddle a rounding boundary
    diff = (got.float() - two.float()).abs()
    assert float(diff.max()) <= 2 ** -7 * scale and float((diff > 0).float().mean()) < 0.08
    # and end to end against float64 from the raw input (act1 rounded to bf16 as both paths do)
    xin = (x - mean) * istd if cmvn else x
    a1 = t.nn.functional.relu(t.nn.functional.conv2d(xin.double().unsqueeze(1), w1.double(), b1.double(), stride=2))
    a1 = a1.float().bfloat16().double()
    ref2 = t.nn.functional.relu(t.nn.functional.conv2d(a1, w2.double(), b2.double(), stride=2))
    assert float((gd - ref2).abs().max()) <= 2 ** -7 * scale


@pytest.mark.parametrize("b,tt", [(2, 64), (3, 249), (1, 301), (70, 249)])
def test_relpos_attention(t, b, tt):
    from mindaudio_amd import ops

    h, dk = 4, 64
    qkv = _rand(t, b * tt, 768, seed=40).bfloat16()
    pos = _rand(t, tt, 256, seed=41).bfloat16()
    u = _rand(t, h, dk, seed=42, scale=0.2)
    v = _rand(t, h, dk, seed=43, scale=0.2)
    lens = ([tt, max(1, tt - 37), max(1, tt // 2)] + [max(1, tt - 3 * i) for i in range(3, b)])[:b]
    mask = t.zeros(b, tt)
    for i, n in enumerate(lens):
        mask[i, :n] = 1.0
    q = qkv[:, :256].double().view(b, tt, h, dk)
    k = qkv[:, 256:512].double().view(b, tt, h, dk).transpose(1, 2)
    vv = qkv[:, 512:].double().view(b, tt, h, dk).transpose(1, 2)
    p = pos.double().view(1, tt, h, dk).transpose(1, 2)
    # the kernel rounds (q + u), (q + v) to bf16 before the MFMA: mirror that rounding in the reference
    qu = (q.float() + u).bfloat16().double().transpose(1, 2)
    qv = (q.float() + v).bfloat16().double().transpose(1, 2)
    scores = (qu @ k.transpose(-1, -2) + qv @ p.transpose(-1, -2)) / 8.0
    scores = scores + (mask[:, None, None, :] == 0).double() * (-10000.0)
    ref = (t.softmax(scores, -1) @ vv).transpose(1, 2).reshape(b * tt, 256)
    got = ops.relpos_attention(qkv.cuda(), pos.cuda(), u.cuda(), v.cuda(), mask.cuda(), b, tt).double().cpu()
    # probabilities are rounded to bf16 before P.V: ~2^-8 relative on sums of <= 1-weighted values
    assert float((got - ref).abs().max()) <= 1.5e-2 * float(ref.abs().max())
    assert float((got - ref).abs().mean()) <= 2e-3 * float(ref.abs().max())
    got_nomask = ops.relpos_attention(qkv.cuda(), pos.cuda(), u.cuda(), v.cuda(), None, b, tt).double().cpu()
    ref_nomask = (t.softmax((qu @ k.transpose(-1, -2) + qv @ p.transpose(-1, -2)) / 8.0, -1) @ vv).transpose(1, 2).reshape(b * tt, 256)
    assert float((got_nomask - ref_nomask).abs().max()) <= 1.5e-2 * float(ref_nomask.abs().max())
    # per-(query, key) masks (B, T, T): static chunks of 16 frames with one chunk of left context, padding folded in
    # (utils/mask.py:169-199 subsequent_chunk_mask & the pad mask, mask.py:262-267)
    idx = t.arange(tt)
    chunk = ((idx[None, :] // 16) <= (idx[:, None] // 16)) & ((idx[None, :] // 16) >= (idx[:, None] // 16) - 1)
    m3 = (chunk[None] & (mask[:, None, :] > 0)).float()
    scores3 = (qu @ k.transpose(-1, -2) + qv @ p.transpose(-1, -2)) / 8.0 + (m3[:, None] == 0).double() * (-10000.0)
    ref3 = (t.softmax(scores3, -1) @ vv).transpose(1, 2).reshape(b * tt, 256)
    got3 = ops.relpos_attention(qkv.cuda(), pos.cuda(), u.cuda(), v.cuda(), m3.cuda(), b, tt).double().cpu()
    assert float((got3 - ref3).abs().max()) <= 1.5e-2 * float(ref3.abs().max())


def test_convmodule_mid(t):
    from mindaudio_amd import ops

    b, tt, c, ks = 3, 77, 256, 15
    y = _rand(t, b * tt, 2 * c, seed=50).bfloat16()
    dw = _rand(t, c, ks, seed=51, scale=0.3)
    sc = t.rand(c, generator=t.Generator().manual_seed(52)) + 0.5
    sh = _rand(t, c, seed=53, scale=0.2)
    yd = y.double().view(b, tt, 2 * c)
    glu = (yd[..., :c] * t.sigmoid(yd[..., c:])).transpose(1, 2)
    z = t.nn.functional.conv1d(glu, dw.double().unsqueeze(1), padding=ks // 2, groups=c).transpose(1, 2)
    z = z * sc.double() + sh.double()
    ref = (z * t.sigmoid(z)).reshape(b * tt, c)
    got = ops.convmodule_mid(y.cuda(), dw.cuda(), sc.cuda(), sh.cuda(), b, tt).double().cpu()
    assert float((got - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max()) * 1.01 + 1e-4


@pytest.mark.parametrize("m,hidden", [(64, 256), (200, 2048), (15936, 2048), (1, 512)])
def test_ffn_packed_pair(t, m, hidden):
    """Last FFN of a block + macaron FFN of the next in one launch == the two single launches (same arithmetic on the same rows:
    the only difference is that x2 / a' stay on chip), and both follow the float64 chain."""
    from mindaudio_amd import ops

    d = 256
    wa1, wb1 = (_rand(t, hidden, d, seed=s_, scale=1.0 / 16).bfloat16().cuda() for s_ in (111, 112))
    wa2, wb2 = (_rand(t, d, hidden, seed=s_, scale=1.0 / math.sqrt(hidden)).bfloat16().cuda() for s_ in (113, 114))
    ba1, bb1 = (_rand(t, hidden, seed=s_, scale=0.3).cuda() for s_ in (115, 116))
    ba2, bb2 = (_rand(t, d, seed=s_, scale=0.3).cuda() for s_ in (117, 118))
    lns = [((1 + 0.1 * _rand(t, d, seed=120 + 2 * i)).cuda(), (0.1 * _rand(t, d, seed=121 + 2 * i)).cuda()) for i in range(4)]
    x = (_rand(t, m, d, seed=119) * 3 + 0.5).cuda()
    pa, pb = ops.ffn_pack_weights(wa1, wa2), ops.ffn_pack_weights(wb1, wb2)
    # two launches: FFN_A (+ norm_final + next norm_ff_macaron), then FFN_B (+ norm_mha)
    x_two = x.clone()
    a2 = ops.ffn_packed(None, pa, ba1, ba2, x_two, lns[1][0], lns[1][1], lns[2][0], lns[2][1], ln_in=lns[0])
    out_two = ops.ffn_packed(a2, pb, bb1, bb2, x_two, lns[3][0], lns[3][1])
    x_one = x.clone()
    out_one = ops.ffn_packed_pair(pa, ba1, ba2, pb, bb1, bb2, x_one, lns[0], lns[1], lns[2], lns[3])
    assert out_one.dtype == t.bfloat16 and out_one.shape == (m, d)
    assert t.equal(x_one, x_two) and t.equal(out_one, out_two)

    def ln(v, gb):
        mu = v.mean(-1, keepdim=True)
        return (v - mu) / t.sqrt(((v - mu) ** 2).mean(-1, keepdim=True) + 1e-5) * gb[0].double().cpu() + gb[1].double().cpu()

    def ffn(a, w1, b1, w2, b2):
        z = a.bfloat16().double() @ w1.double().cpu().T + b1.double().cpu()
        return (z * t.sigmoid(z)).bfloat16().double() @ w2.double().cpu().T + b2.double().cpu()

    # ... and with linear_q/k/v of the attention behind it as the launch's tail: same x, qkv == the K = 256 dense layer on out_one
    wq = _rand(t, 768, d, seed=131, scale=1.0 / 16).bfloat16().cuda()
    bq = _rand(t, 768, seed=132, scale=0.3).cuda()
    pq = ops.ffn_qkv_pack(wq)
    assert pq is not None and t.equal(pq.view(t.int16).sort().values, wq.flatten().view(t.int16).sort().values)
    assert ops.ffn_qkv_pack(wq[:100]) is None
    want_qkv = ops.gemm(out_one, wq, bias=bq)
    x_q = x.clone()
    qkv = ops.ffn_packed_pair(pa, ba1, ba2, pb, bb1, bb2, x_q, lns[0], lns[1], lns[2], lns[3], qkv=(pq, bq))
    assert qkv.shape == (m, 768) and t.equal(x_q, x_one)

    def close(got, want):  # the bias is the accumulator's initial value here, an epilogue add in ops.gemm: last-bit differences
        d_ = (got.float() - want.float()).abs()
        return float(d_.max()) <= 2 ** -7 * float(want.float().abs().max()) and float((d_ > 0).float().mean()) < 0.1

    assert close(qkv, want_qkv)
    # other tail widths: 2, 4 and 8 column blocks per wave (the tail's loop body runs 0, 1 and 3 times around its two peeled blocks)
    for nq_cols in (256, 512, 1024):
        wq2 = _rand(t, nq_cols, d, seed=140 + nq_cols, scale=1.0 / 16).bfloat16().cuda()
        bq2 = _rand(t, nq_cols, seed=141 + nq_cols, scale=0.3).cuda()
        x_q2 = x.clone()
        got2 = ops.ffn_packed_pair(pa, ba1, ba2, pb, bb1, bb2, x_q2, lns[0], lns[1], lns[2], lns[3], qkv=(ops.ffn_qkv_pack(wq2), bq2))
        assert got2.shape == (m, nq_cols) and t.equal(x_q2, x_one) and close(got2, ops.gemm(out_one, wq2, bias=bq2))
    assert ops.ffn_qkv_pack(_rand(t, 384, d, seed=150).bfloat16().cuda()) is None  # whole pairs of column blocks only
    # single FFN + LayerNorm + linear_q/k/v
    a_in = _rand(t, m, d, seed=133).bfloat16().cuda()
    x_s1, x_s2 = x.clone(), x.clone()
    ln_s = ops.ffn_packed(a_in, pa, ba1, ba2, x_s1, lns[3][0], lns[3][1])
    qkv_s = ops.ffn_packed_qkv(a_in, pa, ba1, ba2, x_s2, lns[3][0], lns[3][1], pq, bq)
    assert t.equal(x_s1, x_s2) and close(qkv_s, ops.gemm(ln_s, wq, bias=bq))
    # ... with the LayerNorm of the FFN input folded in as well
    x_s3, x_s4 = x.clone(), x.clone()
    ln_s3 = ops.ffn_packed(None, pa, ba1, ba2, x_s3, lns[3][0], lns[3][1], ln_in=lns[0])
    qkv_s4 = ops.ffn_packed_qkv(None, pa, ba1, ba2, x_s4, lns[3][0], lns[3][1], pq, bq, ln_in=lns[0])
    assert t.equal(x_s3, x_s4) and close(qkv_s4, ops.gemm(ln_s3, wq, bias=bq))
    if m <= 200:
        xd = x.double().cpu()
        x1 = xd + 0.5 * ffn(ln(xd, lns[0]), wa1, ba1, wa2, ba2)
        x2 = ln(x1, lns[1])
        x3 = x2 + 0.5 * ffn(ln(x2, lns[2]), wb1, bb1, wb2, bb2)
        assert float((x_one.double().cpu() - x3).abs().max()) <= 3e-2
        assert float((out_one.double().cpu() - ln(x3, lns[3])).abs().max()) <= 4e-2


@pytest.mark.parametrize("m,hidden,mode", [(64, 256, 0), (250, 2048, 0), (15936, 2048, 1), (7968, 2048, 2), (64 * 3 + 1, 256, 2),
                                           (100, 512, 1), (1, 2048, 0)])
def test_ffn_packed(t, m, hidden, mode):
    """Hidden-slice-owner FFN on fragment-packed weights: same contract as ffn / ffn_ln."""
    from mindaudio_amd import ops

    d = 256
    a = _rand(t, m, d, seed=90).bfloat16()
    w1 = _rand(t, hidden, d, seed=91, scale=1.0 / 16).bfloat16()
    b1 = _rand(t, hidden, seed=92, scale=0.3)
    w2 = _rand(t, d, hidden, seed=93, scale=1.0 / math.sqrt(hidden)).bfloat16()
    b2 = _rand(t, d, seed=94, scale=0.3)
    x = _rand(t, m, d, seed=95) * 3 + 0.5
    g1, be1 = 1 + 0.1 * _rand(t, d, seed=96), 0.1 * _rand(t, d, seed=97)
    g2, be2 = 1 + 0.1 * _rand(t, d, seed=98), 0.1 * _rand(t, d, seed=99)
    z = a.double() @ w1.double().T + b1.double()
    h = (z * t.sigmoid(z)).bfloat16().double()
    xs = x.double() + 0.5 * (h @ w2.double().T + b2.double())

    def ln(v, g, b):
        mu = v.mean(-1, keepdim=True)
        return (v - mu) / t.sqrt(((v - mu) ** 2).mean(-1, keepdim=True) + 1e-5) * g.double() + b.double()

    packed = ops.ffn_pack_weights(w1.cuda(), w2.cuda())
    assert packed.numel() == 2 * d * hidden
    # the packing is a permutation of the weights
    assert t.equal(packed.view(t.int16).sort().values.cpu(), t.cat((w1.flatten(), w2.flatten())).view(t.int16).sort().values)
    xg = x.clone().cuda()
    tol = 3e-3 * float(xs.abs().max())
    if mode == 0:
        r = ops.ffn_packed(a.cuda(), packed, b1.cuda(), b2.cuda(), xg, alpha=0.5)
        assert r is xg and float((xg.double().cpu() - xs).abs().max()) <= tol
        return
    if mode == 1:
        out = ops.ffn_packed(a.cuda(), packed, b1.cuda(), b2.cuda(), xg, g1.cuda(), be1.cuda(), out_dtype=t.float32)
        assert float((xg.double().cpu() - xs).abs().max()) <= tol
        assert float((out.double().cpu() - ln(xs, g1, be1)).abs().max()) <= 2e-2
    else:
        out = ops.ffn_packed(a.cuda(), packed, b1.cuda(), b2.cuda(), xg, g1.cuda(), be1.cuda(), g2.cuda(), be2.cuda(),
                             out_dtype=t.float32)
        y1 = ln(xs, g1, be1)
        assert float((xg.double().cpu() - y1).abs().max()) <= 2e-2
        assert float((out.double().cpu() - ln(y1, g2, be2)).abs().max()) <= 3e-2
    outb = ops.ffn_packed(a.cuda(), packed, b1.cuda(), b2.cuda(), x.clone().cuda(), g1.cuda(), be1.cuda(),
                          *((g2.cuda(), be2.cuda()) if mode == 2 else ()))
    assert outb.dtype == t.bfloat16 and float((outb.float().cpu().double() - out.double().cpu()).abs().max()) <= 3e-2
    # LayerNorm of the INPUT folded into the tile staging: a = LN(x; g0, b0)
    g0, b0 = 1 + 0.1 * _rand(t, d, seed=101), 0.1 * _rand(t, d, seed=102)
    a_ln = ln(x.double(), g0, b0).bfloat16()
    x1, x2 = x.clone().cuda(), x.clone().cuda()
    args = (g1.cuda(), be1.cuda()) + ((g2.cuda(), be2.cuda()) if mode == 2 else ())
    o1 = ops.ffn_packed(a_ln.cuda(), packed, b1.cuda(), b2.cuda(), x1, *args, out_dtype=t.float32)
    o2 = ops.ffn_packed(None, packed, b1.cuda(), b2.cuda(), x2, *args, out_dtype=t.float32, ln_in=(g0.cuda(), b0.cuda()))
    # identical up to bf16 roundings of a that flip on float32-vs-float64 LayerNorm arithmetic
    assert float((x1 - x2).abs().max()) <= 2e-2 * float(x1.abs().max()) and float((o1 - o2).abs().max()) <= 5e-2
    assert float((x1 - x2).abs().mean()) <= 2e-4 * float(x1.abs().max())
    # run-to-run determinism (fixed reduction order across the four waves)
    xg2 = x.clone().cuda()
    ops.ffn_packed(a.cuda(), packed, b1.cuda(), b2.cuda(), xg2, g1.cuda(), be1.cuda(), *((g2.cuda(), be2.cuda()) if mode == 2 else ()),
                   out_dtype=t.float32)
    assert t.equal(xg2, xg)


