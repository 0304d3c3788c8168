"""GPU parity: EcapaTDNN forward (mindaudio_amd.models.EcapaTDNN, HIP kernels) vs the PyTorch-CPU float32 oracle
(oracle/ecapa_oracle.py, parity unpinned).  The device multiplies in bf16 with float32 accumulation: the embedding is
compared by relative RMS error and cosine similarity."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def build(c=512, seed=3, asp_gain=1.0):
    from mindaudio_amd.models import EcapaTDNN
    from oracle import ecapa_oracle as E

    torch.manual_seed(seed)
    ref = E.EcapaTDNN(80, channels=(c, c, c, c, 3 * c)).eval()
    with torch.no_grad():
        if asp_gain != 1.0:  # random-init attention logits are ~ +-0.1 (near-uniform pooling weights): make the softmax matter
            ref.asp.conv.weight.mul_(asp_gain)
        for m in ref.modules():  # non-trivial BatchNorm statistics
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.8, 1.2)
                m.bias.normal_(0, 0.1)
    dut = EcapaTDNN(80, channels=(c, c, c, c, 3 * c)).eval()
    missing, unexpected = dut.load_state_dict(ref.state_dict(), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return ref, dut.cuda().prepare()


@pytest.mark.parametrize("b,t", [(3, 57), (2, 300)])
def test_ecapa_forward_matches_oracle(b, t):
    ref, dut = build()
    x = torch.randn(b, t, 80)
    with torch.no_grad():
        want = ref(x)
    got = dut(x.cuda()).cpu()
    assert got.shape == want.shape == (b, 192)
    rel = float((got - want).norm() / want.norm())
    cos = torch.nn.functional.cosine_similarity(got, want, dim=1)
    assert rel < 3e-2, rel
    assert float(cos.min()) > 0.999


def test_weights_loaded_after_a_forward_are_the_ones_the_next_forward_uses():
    """prepare() keeps bf16 / packed copies of the weights; a load_state_dict behind a forward has to drop them (the Conformer encoder,
    decoder and CTC head have the same hook; the decoder's copy once outlived sync_to_module - round 6)."""
    _, dut = build(seed=3)
    x = torch.randn(2, 57, 80).cuda()
    first = dut(x).clone()
    ref2, fresh = build(seed=4)
    dut.load_state_dict(ref2.state_dict(), strict=False)
    got, want = dut(x), fresh(x)
    assert torch.equal(got, want) and not torch.equal(got, first)


def test_ecapa_conv_taps_and_epilogue_against_torch():
    """The tap-convolution GEMM mode and the ReLU -> BatchNorm -> tanh epilogue on their own, float32 reference."""
    import ctypes

    from mindaudio_amd import _host, _lib
    from mindaudio_amd.models.ecapatdnn import HALO, _conv

    torch.manual_seed(0)
    b, t, cin, cout, k, d = 2, 41, 64, 128, 3, 3
    x = torch.randn(b, cin, t)
    conv = torch.nn.Conv1d(cin, cout, k, dilation=d, padding=d)
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout) * 0.1
    want = torch.tanh(torch.relu(conv(x.bfloat16().float())) * scale[None, :, None] + shift[None, :, None])
    tp = t + 2 * HALO
    full = torch.zeros(b * tp + 2 * HALO, cin, dtype=torch.bfloat16)
    view = full[HALO:HALO + b * tp].view(b, tp, cin)
    view[:, HALO:HALO + t] = x.transpose(1, 2).bfloat16()
    full = full.cuda()
    w = conv.weight.detach().permute(0, 2, 1).contiguous().bfloat16().view(cout, -1).cuda()
    out = torch.empty(b * tp, cout, dtype=torch.bfloat16, device="cuda")
    rs = torch.zeros(b, tp)
    rs[:, HALO:HALO + t] = 1
    _conv(full[HALO:].data_ptr(), cin, b * tp, cin, w, k, d, out.data_ptr(), cout, cout, conv.bias.detach().cuda(),
          _lib.ACT_RELU, (scale.cuda(), shift.cuda()), act2=_lib.ACT_TANH, row_scale=rs.view(-1).cuda())
    got = out.float().cpu().view(b, tp, cout)
    assert float(got[:, :HALO].abs().max()) == 0.0 and float(got[:, HALO + t:].abs().max()) == 0.0
    got = got[:, HALO:HALO + t].transpose(1, 2)
    wq = conv.weight.detach().bfloat16().float()
    want = torch.tanh(torch.relu(torch.nn.functional.conv1d(x.bfloat16().float(), wq, conv.bias, dilation=d, padding=d))
                      * scale[None, :, None] + shift[None, :, None])
    assert float((got - want).abs().max()) < 1.5e-2


@pytest.mark.parametrize("c", [512, 1024])
def test_ecapa_cfg5_full_shape(c):
    """BASELINE cfg 5 at its stated shape — (256, 300, 80) — for the class default C = 512 (ecapatdnn.py:333) and the
    example's C = 1024 (examples/ECAPA-TDNN/train_speaker_embeddings.py:468-472): the oracle on a sub-batch (eval-mode
    BatchNorm: every utterance is independent of its batch), and batch-size independence / finiteness on the full batch."""
    ref, dut = build(c=c, seed=4)
    g = torch.Generator().manual_seed(c)
    x = torch.randn(256, 300, 80, generator=g)
    got = dut(x.cuda()).cpu()
    assert got.shape == (256, 192) and bool(torch.isfinite(got).all())
    idx = [0, 1, 127, 255]
    with torch.no_grad():
        want = ref(x[idx])
    rel = float((got[idx] - want).norm() / want.norm())
    cos = torch.nn.functional.cosine_similarity(got[idx], want, dim=1)
    assert rel < 3e-2 and float(cos.min()) > 0.999, (rel, float(cos.min()))
    # an utterance's embedding does not depend on what else is in the batch (same kernels, different tiling of rows)
    alone = dut(x[idx].cuda()).cpu()
    assert float((alone - got[idx]).abs().max()) <= 2e-2 * float(got[idx].abs().max())
    # and the launch is deterministic
    assert torch.equal(dut(x.cuda()).cpu(), got)


@pytest.mark.parametrize("c,b,t", [(512, 3, 57), (512, 2, 300), (1024, 2, 300), (512, 2, 376), (1024, 3, 376)])
def test_res2net_chain_in_one_launch_equals_the_separate_launches(c, b, t):
    """ma_res2net_fused_bf16 (the 7 dilated convolutions + adds of a Res2NetBlock, ecapatdnn.py:66-114, one utterance per
    workgroup) against the one-launch-per-cell form on the same weights: same bf16 rounding points, so the embeddings agree to
    accumulation-order round-off; both are held to the float32 oracle by the tests above."""
    ref, dut = build(c=c, seed=5)
    x = torch.randn(b, t, 80, generator=torch.Generator().manual_seed(t)).cuda()
    assert dut.fuse_res2net
    fused = dut(x)
    dut.fuse_res2net = False
    plain = dut(x)
    assert float((fused - plain).abs().max()) <= 2e-2 * float(plain.abs().max())
    with torch.no_grad():
        want = ref(x.cpu())
    assert float((fused.cpu() - want).norm() / want.norm()) < 3e-2


@pytest.mark.parametrize("c,b,t", [(512, 3, 57), (512, 2, 300), (1024, 2, 300), (512, 2, 9)])
def test_asp_logits_and_pooling_in_one_launch_equal_the_two_launches(c, b, t):
    """ma_asp_fused_bf16 (logits = a1 Wc^T + b and the attentive statistics pooling, ecapatdnn.py:284-308, in one launch with
    float32 logits) against the GEMM launch (bf16 logits) + ma_asp_pool_bf16 on the same weights, and both against the oracle."""
    ref, dut = build(c=c, seed=7, asp_gain=40.0)  # (peaky attention: uniform weights would be far outside the tolerances below)
    x = torch.randn(b, t, 80, generator=torch.Generator().manual_seed(t + 1)).cuda()
    assert dut.fuse_asp
    fused = dut(x)
    dut.fuse_asp = False
    plain = dut(x)
    assert float((fused - plain).abs().max()) <= 2e-2 * float(plain.abs().max())
    with torch.no_grad():
        want = ref(x.cpu())
    assert float((fused.cpu() - want).norm() / want.norm()) < 3e-2
    assert torch.equal(fused, (setattr(dut, "fuse_asp", True), dut(x))[1])  # deterministic


@pytest.mark.parametrize("c,b,t", [(512, 16, 57), (512, 3, 120), (1024, 32, 100)])
def test_se_excitation_and_embedding_linear_in_single_launches(c, b, t):
    """ma_se_gate_bf16 (both 1 x 1 convolutions of the SE excitation, ecapatdnn.py:150-156, float32 hidden vector) and
    ma_linear_small_bf16 (the embedding Linear, ecapatdnn.py:429-431; batch % 16 == 0, else the GEMM launch) against the
    ma_gemm_bf16 launches on the same weights, and against the oracle."""
    ref, dut = build(c=c, seed=8)
    x = torch.randn(b, t, 80, generator=torch.Generator().manual_seed(b + t)).cuda()
    assert dut.fuse_se
    fused = dut(x)
    dut.fuse_se = False
    plain = dut(x)
    assert float((fused - plain).abs().max()) <= 2e-2 * float(plain.abs().max())
    with torch.no_grad():
        want = ref(x.cpu())
    assert float((fused.cpu() - want).norm() / want.norm()) < 3e-2


def test_res2net_long_utterances_fall_back():
    ref, dut = build(c=512, seed=6)
    x = torch.randn(2, 500, 80)  # T + 2H > 384 rows: the chain runs as separate launches
    with torch.no_grad():
        want = ref(x)
    got = dut(x.cuda()).cpu()
    assert float((got - want).norm() / want.norm()) < 3e-2


# (300, 33, 1536): 1 800 workgroups = seven resident rounds (every kernel once above the chip's residency, against float64)
@pytest.mark.parametrize("b,t,c", [(1, 1, 256), (2, 15, 256), (3, 16, 512), (2, 17, 256), (3, 33, 1536), (5, 300, 1536), (2, 64, 3072),
                                   (300, 33, 1536)])
def test_asp_fused_entry_point_against_float64(b, t, c):
    """ma_asp_fused_bf16 alone (ecapatdnn.py:296-308) at tile boundaries of the frame axis (16-frame tiles, prefetch three tiles
    ahead: T = 1, 15, 16, 17, 33) against a float64 evaluation of the same bf16 operands."""
    from mindaudio_amd import _host, _lib
    from mindaudio_amd.models.ecapatdnn import HALO

    lib = _lib.load()
    g = torch.Generator().manual_seed(100 * t + c)
    tp = t + 2 * HALO
    a1 = torch.tanh(torch.randn(b, tp, 128, generator=g)).bfloat16()
    w = (torch.randn(c, 128, generator=g) * 0.3).bfloat16()
    x = torch.randn(b, tp, c, generator=g).bfloat16()
    sc, sh = torch.rand(2 * c, generator=g) + 0.5, torch.randn(2 * c, generator=g) * 0.1
    a1d, wd, xd, scd, shd = a1.cuda().view(b * tp, 128), w.cuda(), x.cuda().view(b * tp, c), sc.cuda(), sh.cuda()
    out = torch.empty(b, 2 * c, dtype=torch.bfloat16, device="cuda")
    rc = lib.ma_asp_fused_bf16(a1d.data_ptr(), 128, wd.data_ptr(), xd.data_ptr(), c, b, t, HALO, c, 128, 1e-12, scd.data_ptr(),
                               shd.data_ptr(), out.data_ptr(), _host.current_stream_ptr())
    _lib.check(rc, "asp_fused")
    A, X = a1[:, HALO:HALO + t].double(), x[:, HALO:HALO + t].double()
    logits = A @ w.double().t()                                   # (b, t, c); the bias cancels in the softmax over t
    p_ = torch.softmax(logits, dim=1)
    mean = (p_ * X).sum(1)
    std = ((p_ * (X - mean[:, None]) ** 2).sum(1)).clamp(min=1e-12).sqrt()
    want = torch.cat([mean, std], 1) * sc.double() + sh.double()
    got = out.float().cpu().double()
    assert float((got - want).abs().max()) <= 2e-2 * float(want.abs().max()) + 1e-2, float((got - want).abs().max())


@pytest.mark.parametrize("b,c,s_", [(1, 512, 128), (7, 512, 64), (3, 1024, 128), (16, 1024, 8)])
def test_se_gate_entry_point_against_float64(b, c, s_):
    from mindaudio_amd import _host, _lib

    lib = _lib.load()
    g = torch.Generator().manual_seed(c + s_)
    mean = torch.randn(b, c, generator=g).bfloat16()
    w1, b1 = (torch.randn(s_, c, generator=g) / 16).bfloat16(), torch.randn(s_, generator=g) * 0.2
    w2, b2 = (torch.randn(c, s_, generator=g) / 8).bfloat16(), torch.randn(c, generator=g) * 0.2
    gate = torch.empty(b, c, dtype=torch.bfloat16, device="cuda")
    dev = [v.cuda() for v in (mean, w1, b1, w2, b2)]
    rc = lib.ma_se_gate_bf16(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(),
                             gate.data_ptr(), b, c, s_, _host.current_stream_ptr())
    _lib.check(rc, "se_gate")
    h = torch.relu(mean.double() @ w1.double().t() + b1.double())
    want = torch.sigmoid(h @ w2.double().t() + b2.double())
    assert float((gate.float().cpu().double() - want).abs().max()) <= 5e-3


@pytest.mark.parametrize("m,n,k", [(16, 16, 3072), (256, 192, 3072), (32, 64, 6144)])
def test_linear_small_entry_point_against_float64(m, n, k):
    from mindaudio_amd import _host, _lib

    lib = _lib.load()
    g = torch.Generator().manual_seed(m + n)
    a, w, bias = torch.randn(m, k, generator=g).bfloat16(), (torch.randn(n, k, generator=g) / 50).bfloat16(), torch.randn(n, generator=g)
    out = torch.empty(m, n, device="cuda")
    ad, wd, bd = a.cuda(), w.cuda(), bias.cuda()
    rc = lib.ma_linear_small_bf16(ad.data_ptr(), k, wd.data_ptr(), k, bd.data_ptr(), out.data_ptr(), n, m, n, k, _host.current_stream_ptr())
    _lib.check(rc, "linear_small")
    want = a.double() @ w.double().t() + bias.double()
    assert float((out.cpu().double() - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-4
    # shapes outside the kernel's tiling are refused (callers then run ma_gemm_bf16)
    assert lib.ma_linear_small_bf16(ad.data_ptr(), k, wd.data_ptr(), k, bd.data_ptr(), out.data_ptr(), n, m - 1, n, k, None) == _lib.MA_ERR_UNSUPPORTED


# (64, 800, 20, 2) / (128, 780, 9, 3): one workgroup per utterance, 800 / 780 workgroups = more than three resident rounds
@pytest.mark.parametrize("cc,b,t,d", [(64, 1, 1, 2), (64, 2, 20, 3), (64, 3, 57, 4), (128, 2, 33, 2), (128, 1, 300, 3), (64, 2, 376, 4),
                                       (128, 2, 376, 4), (64, 800, 20, 2), (128, 780, 9, 3)])
def test_res2net_fused_entry_point_against_float64(cc, b, t, d):
    """ma_res2net_fused_bf16 alone (Res2NetBlock, ecapatdnn.py:66-114: y_0 = x_0, y_i = BN(ReLU(conv_{k=3, dilation d}(x_i + y_{i-1})))
    on 8 channel groups) against a float64 chain with the kernel's rounding points (bf16 operands, bf16 y, bf16 x_i + y_{i-1})."""
    from mindaudio_amd import _host, _lib
    from mindaudio_amd.models.ecapatdnn import HALO

    lib = _lib.load()
    g = torch.Generator().manual_seed(1000 * cc + 10 * t + d)
    c, tp, H = 8 * cc, t + 2 * HALO, HALO
    x = torch.zeros(b, tp, c)
    x[:, H:H + t] = torch.randn(b, t, c, generator=g)
    x = x.bfloat16()
    w = (torch.randn(7, cc, 3 * cc, generator=g) / (3 * cc) ** 0.5).bfloat16()   # (step, out, tap * cc + in)
    bias, sc, sh = torch.randn(7, cc, generator=g) * 0.1, torch.rand(7, cc, generator=g) + 0.5, torch.randn(7, cc, generator=g) * 0.1
    xd = torch.zeros(b * tp + 2 * H, c, dtype=torch.bfloat16, device="cuda")
    xd[H:H + b * tp] = x.view(b * tp, c).cuda()
    yd = torch.zeros_like(xd)
    dev = [v.cuda().contiguous() for v in (w, bias, sc, sh)]
    rc = lib.ma_res2net_fused_bf16(xd[H:].data_ptr(), c, yd[H:].data_ptr(), c, b, t, H, cc, 8, d, dev[0].data_ptr(), dev[1].data_ptr(),
                                   dev[2].data_ptr(), dev[3].data_ptr(), _host.current_stream_ptr())
    _lib.check(rc, "res2net_fused")
    got = yd[H:H + b * tp].view(b, tp, c).float().cpu()
    X = x.double()
    want = torch.zeros(b, tp, c, dtype=torch.float64)
    want[..., :cc] = X[..., :cc]
    keep = torch.zeros(tp, dtype=torch.float64)
    keep[H:H + t] = 1
    prev = None
    for i in range(1, 8):
        inp = X[..., i * cc:(i + 1) * cc] if prev is None else (X[..., i * cc:(i + 1) * cc] + prev).bfloat16().double()
        pad = torch.zeros(b, tp + 2 * d, cc, dtype=torch.float64)
        pad[:, d:d + tp] = inp
        W = w[i - 1].double().view(cc, 3, cc)
        out = sum(pad[:, k * d:k * d + tp] @ W[:, k].t() for k in range(3)) + bias[i - 1].double()
        out = (torch.relu(out) * sc[i - 1].double() + sh[i - 1].double()) * keep[None, :, None]
        prev = out.bfloat16().double()
        want[..., i * cc:(i + 1) * cc] = prev
    err = (got.double() - want).abs()
    # bf16 outputs of a 7-deep chain: a rounding flip early in the chain moves later groups by a bf16 step of their inputs
    assert float(err.max()) <= 4e-2 * float(want.abs().max()), float(err.max())
    assert float(err.mean()) <= 2e-3 * float(want.abs().mean() + 1e-6)
    assert float(got[:, :H].abs().max()) == 0.0 and float(got[:, H + t:].abs().max()) == 0.0  # halo frames stay zero


# (800, 20, 512, 128): one workgroup per utterance, 800 workgroups
@pytest.mark.parametrize("b,t,c,s_", [(1, 1, 512, 128), (3, 57, 512, 128), (2, 300, 512, 64), (2, 333, 512, 128), (3, 100, 1024, 128),
                                      (2, 300, 1024, 16), (800, 20, 512, 128)])
def test_se_block_entry_point_equals_the_three_launches(b, t, c, s_):
    """ma_se_block_bf16 (squeeze + excitation + scale + residual, ecapatdnn.py:150-157, 246) against ma_time_mean_bf16 +
    ma_se_gate_bf16 + ma_se_apply_bf16 on the same buffers (same rounding points: bf16 mean and gate), halo frames zero, and
    against float64."""
    from mindaudio_amd import _host, _lib
    from mindaudio_amd.models.ecapatdnn import HALO

    lib = _lib.load()
    g = torch.Generator().manual_seed(7 * t + c + s_)
    H, tp = HALO, t + 2 * HALO
    x = torch.zeros(b, tp, c)
    x[:, H:H + t] = torch.randn(b, t, c, generator=g)
    res = torch.zeros(b, tp, 3 * c)                      # the residual lives in a wider buffer (the concatenated block outputs)
    res[:, H:H + t, :c] = torch.randn(b, t, c, generator=g)
    w1, b1 = (torch.randn(s_, c, generator=g) / 16).bfloat16(), torch.randn(s_, generator=g) * 0.2
    w2, b2 = (torch.randn(c, s_, generator=g) / 4).bfloat16(), torch.randn(c, generator=g) * 0.2
    xd, rd = x.bfloat16().cuda().view(b * tp, c), res.bfloat16().cuda().view(b * tp, 3 * c)
    dev = [v.cuda() for v in (w1, b1, w2, b2)]
    out1 = torch.full((b * tp, 3 * c), 7.0, dtype=torch.bfloat16, device="cuda")
    out2 = out1.clone()
    s = _host.current_stream_ptr()
    _lib.check(lib.ma_se_block_bf16(xd.data_ptr(), c, dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(),
                                    rd.data_ptr(), 3 * c, out1[:, c:].data_ptr(), 3 * c, b, t, H, c, s_, s), "se_block")
    mean = torch.empty(b, c, dtype=torch.bfloat16, device="cuda")
    gate = torch.empty(b, c, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.ma_time_mean_bf16(xd.data_ptr(), c, b, t, H, c, mean.data_ptr(), s), "time_mean")
    _lib.check(lib.ma_se_gate_bf16(mean.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(),
                                   gate.data_ptr(), b, c, s_, s), "se_gate")
    _lib.check(lib.ma_se_apply_bf16(xd.data_ptr(), c, gate.data_ptr(), rd.data_ptr(), 3 * c, out2[:, c:].data_ptr(), 3 * c, b, t, H, c, s),
               "se_apply")
    got, three = out1[:, c:2 * c].float().cpu().view(b, tp, c), out2[:, c:2 * c].float().cpu().view(b, tp, c)
    assert float(got[:, :H].abs().max()) == 0.0 and float(got[:, H + t:].abs().max()) == 0.0
    assert torch.equal(out1[:, :c], out2[:, :c]) and torch.equal(out1[:, 2 * c:], out2[:, 2 * c:])  # nothing outside the slice
    # same rounding points; the mean's float32 summation order differs (a bf16 step of the mean can move a gate by one bf16 step)
    assert float((got - three).abs().max()) <= 2 ** -6 * float(three.abs().max())
    X = x.bfloat16().double()[:, H:H + t]
    m_ = X.mean(1).float().bfloat16().double()
    gt = torch.sigmoid(torch.relu(m_ @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double())
    want = X * gt[:, None] + res.bfloat16().double()[:, H:H + t, :c]
    assert float((got[:, H:H + t].double() - want).abs().max()) <= 3e-2 * float(want.abs().max())



def test_random_batch_shapes_through_ecapa_against_the_oracle():
    """Shapes between the pinned ones: 16 random (batch 1 ... 24, 9 ... 700 frames; the frame counts around the 16- / 32- / 64-frame tiles
    of the fused Res2Net, SE and attentive-pooling launches first) at C = 512 with a peaky pooling softmax, all fused forms on."""
    import numpy as np

    ref, dut = build(512, seed=5, asp_gain=6.0)
    rng = np.random.RandomState(8)
    picks = [9, 15, 16, 17, 31, 33, 63, 65, 127, 129]
    for case in range(16):
        b = int(rng.randint(1, 25))
        t = picks[case] if case < len(picks) else int(rng.randint(9, 700))
        x = torch.from_numpy(rng.randn(b, t, 80).astype(np.float32))
        with torch.no_grad():
            want = ref(x)
        got = dut(x.cuda()).cpu()
        rel = float((got - want).norm() / want.norm())
        cos = float(torch.nn.functional.cosine_similarity(got, want, dim=1).min())
        assert got.shape == want.shape and rel < 3e-2 and cos > 0.999, (case, b, t, rel, cos)
