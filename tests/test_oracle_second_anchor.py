"""Second, independent anchor for the parity-UNPINNED half of the oracle (oracle/conformer_oracle.py, the mel bank of
oracle/speech_features.py).  MindSpore cannot run here, so nothing can pin these against the reference itself; what these tests
remove is single-author risk: every block of the oracle is re-derived from the reference's text in a DIFFERENT formulation
(library primitives with other code paths, einsum contractions, explicit per-filter loops) that shares no code with the oracle,
in float64 where the formulation allows.  CPU only."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import conformer_oracle as C
from oracle import speech_features as O


def test_layernorm_is_biased_variance_eps_inside_sqrt():
    """layers/layernorm.py:53-60 == torch's layer_norm (biased variance, eps inside the square root)."""
    torch.manual_seed(0)
    ln = C.LayerNorm(256)
    with torch.no_grad():
        ln.gamma.uniform_(0.5, 1.5)
        ln.beta.normal_()
    x = torch.randn(7, 33, 256) * 3 + 1
    want = F.layer_norm(x.double(), (256,), ln.gamma.double(), ln.beta.double(), eps=1e-5)
    assert torch.allclose(ln(x).double(), want, atol=2e-6)


def test_feed_forward_swish():
    """positionwise_feed_forward.py:33-46 with Swish (swish.py:14-16: x * sigmoid(x)) == F.silu between two F.linear."""
    torch.manual_seed(1)
    ff = C.PositionwiseFeedForward(256, 2048, 0.0).eval()
    x = torch.randn(5, 17, 256)
    want = F.linear(F.silu(F.linear(x.double(), ff.w_1.weight.double(), ff.w_1.bias.double())), ff.w_2.weight.double(),
                    ff.w_2.bias.double())
    assert torch.allclose(ff(x).double(), want, atol=1e-5)


@pytest.mark.parametrize("train", [False, True])
def test_convolution_module(train):
    """layers/convolution.py:83-129 re-derived with F.glu / grouped F.conv1d / F.batch_norm on the (B*T, C) view the reference
    normalises (padded frames included: convolution.py:113-121), float64."""
    torch.manual_seed(2)
    b, t, c, k = 3, 29, 256, 15
    cm = C.ConvolutionModule(c, k)
    with torch.no_grad():
        cm.norm.running_mean.normal_(0, 0.3)
        cm.norm.running_var.uniform_(0.5, 2.0)
        cm.norm.weight.uniform_(0.5, 1.5)
        cm.norm.bias.normal_(0, 0.2)
    cm.train(train)
    x = torch.randn(b, t, c)
    mask = torch.ones(b, 1, t)
    mask[1, 0, 20:] = 0
    mask[2, 0, 11:] = 0
    rm, rv = cm.norm.running_mean.clone().double(), cm.norm.running_var.clone().double()
    got = cm(x, mask).double()
    d = lambda p: p.detach().double()  # noqa: E731
    y = (x.double() * mask.double().transpose(1, 2)).transpose(1, 2)                   # (B, C, T), masked
    y = F.glu(F.conv1d(y, d(cm.pointwise_conv1.weight), d(cm.pointwise_conv1.bias)), dim=1)
    y = F.conv1d(y, d(cm.depthwise_conv.weight), d(cm.depthwise_conv.bias), padding=(k - 1) // 2, groups=c)
    rows = y.transpose(1, 2).reshape(b * t, c)
    rows = F.batch_norm(rows, rm, rv, d(cm.norm.weight), d(cm.norm.bias), training=train, momentum=0.1, eps=1e-5)
    y = F.silu(rows).reshape(b, t, c).transpose(1, 2)
    y = F.conv1d(y, d(cm.pointwise_conv2.weight), d(cm.pointwise_conv2.bias)) * mask.double()
    assert torch.allclose(got, y.transpose(1, 2), atol=2e-5)
    if train:  # running statistics moved with momentum 0.1 and the unbiased variance
        assert torch.allclose(cm.norm.running_mean.double(), rm, atol=1e-5)
        assert torch.allclose(cm.norm.running_var.double(), rv, atol=1e-5)


def test_relpos_attention_einsum():
    """layers/attention.py:214-235 written as einsum contractions straight from the reference's text: scores =
    ((q + u) k^T + (q + v) p^T) / sqrt(d_k) with NO relative shift, mask -> + (mask == 0) * -10000 (attention.py:100-107),
    softmax over keys, context, linear_out."""
    torch.manual_seed(3)
    b, t, h, dk = 3, 23, 4, 64
    att = C.RelPositionMultiHeadedAttention(h, h * dk, 0.0).eval()
    x = torch.randn(b, t, h * dk)
    pos = torch.randn(1, t, h * dk)
    mask = torch.ones(b, 1, t)
    mask[1, 0, 15:] = 0
    mask[2, 0, 5:] = 0
    d = lambda p: p.detach().double()  # noqa: E731
    xd = x.double()
    q = (xd @ d(att.linear_q.weight).T + d(att.linear_q.bias)).reshape(b, t, h, dk)
    k = (xd @ d(att.linear_k.weight).T + d(att.linear_k.bias)).reshape(b, t, h, dk)
    v = (xd @ d(att.linear_v.weight).T + d(att.linear_v.bias)).reshape(b, t, h, dk)
    p = (pos.double() @ d(att.linear_pos.weight).T).reshape(t, h, dk)
    ac = torch.einsum("bihd,bjhd->bhij", q + d(att.pos_bias_u), k)
    bd = torch.einsum("bihd,jhd->bhij", q + d(att.pos_bias_v), p)
    scores = (ac + bd) / math.sqrt(dk) + (mask.double() == 0).double()[:, :, None, :] * -10000.0
    ctx = torch.einsum("bhij,bjhd->bihd", torch.softmax(scores, dim=-1), v).reshape(b, t, h * dk)
    want = ctx @ d(att.linear_out.weight).T + d(att.linear_out.bias)
    assert torch.allclose(att(x, mask, pos).double(), want, atol=2e-5)


def test_decoder_attention_divides_by_dk():
    """layers/attention.py:150-152: q AND k are both scaled by 1/sqrt(d_k) -> scores / d_k."""
    torch.manual_seed(4)
    b, t1, t2, h, dk = 2, 5, 9, 4, 64
    att = C.MultiHeadedAttention(h, h * dk).eval()
    qx, kx = torch.randn(b, t1, h * dk), torch.randn(b, t2, h * dk)
    mask = torch.ones(b, 1, t2)
    mask[1, 0, 6:] = 0
    d = lambda p: p.detach().double()  # noqa: E731
    q = (qx.double() @ d(att.linear_q.weight).T + d(att.linear_q.bias)).reshape(b, t1, h, dk)
    k = (kx.double() @ d(att.linear_k.weight).T + d(att.linear_k.bias)).reshape(b, t2, h, dk)
    v = (kx.double() @ d(att.linear_v.weight).T + d(att.linear_v.bias)).reshape(b, t2, h, dk)
    scores = torch.einsum("bihd,bjhd->bhij", q, k) / dk + (mask.double() == 0).double()[:, :, None, :] * -10000.0
    ctx = torch.einsum("bhij,bjhd->bihd", torch.softmax(scores, -1), v).reshape(b, t1, h * dk)
    want = ctx @ d(att.linear_out.weight).T + d(att.linear_out.bias)
    assert torch.allclose(att(qx, kx, kx, mask).double(), want, atol=2e-5)


def test_encoder_layer_wiring():
    """models/conformer.py:100-161: macaron half-steps, pre-norm residual branches, final LayerNorm - assembled here from the
    layer's OWN sub-modules in the order the reference's construct() lists them (checks the wiring, not the blocks)."""
    torch.manual_seed(5)
    layer = C.ConformerEncoderLayer(256, 4, 512, 15, 0.0, 0.0).eval()
    x, pos = torch.randn(2, 19, 256), torch.randn(1, 19, 256)
    mask = torch.ones(2, 1, 19)
    mask[1, 0, 12:] = 0
    r = x
    r = r + 0.5 * layer.feed_forward_macaron(layer.norm_ff_macaron(r))      # :109-112
    r = r + layer.self_attn(layer.norm_mha(r), mask, pos)                   # :117-135
    r = r + layer.conv_module(layer.norm_conv(r), mask)                     # :139-143
    r = r + 0.5 * layer.feed_forward(layer.norm_ff(r))                      # :147-151
    r = layer.norm_final(r)                                                 # :153-156
    assert torch.allclose(layer(x, mask, pos, mask), r, atol=1e-6)


def test_subsampling_front_end():
    """layers/subsampling.py:40-78: two valid 3x3 stride-2 convolutions + ReLU, flatten (c, f), Dense, x sqrt(d); positional table
    rows 0..T'-1 of the sinusoid (embedding.py:36-44, 86-88) - convolutions re-derived with F.unfold (im2col) in float64."""
    torch.manual_seed(6)
    emb = C.Conv2dSubsampling4(80, 256, 0.0).eval()
    x = torch.randn(2, 67, 80)
    got, pos = emb(x)
    d = lambda p: p.detach().double()  # noqa: E731

    def conv(inp, w, bias):  # inp (B, C, H, W) -> valid 3x3 stride 2 via unfold
        bsz, _, hh, ww = inp.shape
        cols = F.unfold(inp, kernel_size=3, stride=2)                       # (B, C*9, L)
        out = torch.einsum("ok,bkl->bol", w.reshape(w.shape[0], -1), cols) + bias[None, :, None]
        return out.reshape(bsz, w.shape[0], (hh - 3) // 2 + 1, (ww - 3) // 2 + 1)

    y = torch.relu(conv(x.double()[:, None], d(emb.conv1.weight), d(emb.conv1.bias)))
    y = torch.relu(conv(y, d(emb.conv2.weight), d(emb.conv2.bias)))
    bsz, c, t, f = y.shape
    y = (y.permute(0, 2, 1, 3).reshape(bsz, t, c * f) @ d(emb.out.weight).T + d(emb.out.bias)) * math.sqrt(256)
    assert torch.allclose(got.double(), y, atol=5e-5)
    k = np.arange(t)[:, None] * np.exp(np.arange(0, 256, 2) * -(math.log(10000.0) / 256))[None, :]
    assert np.allclose(pos[0, :, 0::2].numpy(), np.sin(k), atol=2e-4) and np.allclose(pos[0, :, 1::2].numpy(), np.cos(k), atol=2e-4)


def test_ctc_loss_against_brute_force_path_sum():
    """loss/ctc_loss.py:10-64: -log sum over all alignments (blank 0) of the product of per-frame softmax probabilities, summed
    over the batch and divided by the batch size - enumerated explicitly for a tiny case."""
    import itertools

    torch.manual_seed(7)
    ctc = C.CTC(4, 8).eval()
    hs = torch.randn(2, 5, 8)
    hlens, ys, ylens = torch.tensor([5, 4]), torch.tensor([[1, 2], [3, 0]]), torch.tensor([2, 1])
    lp = torch.log_softmax(hs.double() @ ctc.ctc_lo.weight.detach().double().T + ctc.ctc_lo.bias.detach().double(), -1)
    total = 0.0
    for bi in range(2):
        tt, target = int(hlens[bi]), ys[bi, :int(ylens[bi])].tolist()
        acc = 0.0
        for path in itertools.product(range(4), repeat=tt):
            collapsed = [s for s, prev in zip(path, (None,) + path[:-1]) if s != prev and s != 0]
            if collapsed == target:
                acc += math.exp(sum(float(lp[bi, ti, s]) for ti, s in enumerate(path)))
        total += -math.log(acc)
    assert abs(float(ctc(hs, hlens, ys, ylens)) - total / 2) < 1e-5


def test_label_smoothing_loss_closed_form():
    """loss/label_smoothing_loss.py:24-117: KL(true_dist || softmax) with true_dist = (1 - s) one-hot + s / (V - 1) elsewhere, padded
    targets (-1) ignored, summed and divided by the batch size."""
    torch.manual_seed(8)
    b, l, v, s = 3, 6, 11, 0.1
    logits = torch.randn(b, l, v)
    tgt = torch.randint(0, v, (b, l))
    masks = torch.ones(b, 1, l)
    masks[1, 0, 4:] = 0
    masks[2, 0, 2:] = 0
    logp = torch.log_softmax(logits.double(), -1)
    total = 0.0
    for bi in range(b):
        for li in range(l):
            if masks[bi, 0, li] == 0:
                continue
            for vi in range(v):
                pt = 1.0 - s if vi == int(tgt[bi, li]) else s / (v - 1)
                total += pt * (math.log(pt) - float(logp[bi, li, vi]))
    got = C.label_smoothing_loss(logits, tgt.masked_fill(masks[:, 0] == 0, -1), masks, s)
    assert abs(float(got) - total / b) < 1e-5


def test_htk_mel_bank_per_filter_closed_form():
    """SURVEY a3 / MelScale(mel_type=HTK, norm=NONE): fb[f, m] = max(0, min((f - f_m) / (f_{m+1} - f_m), (f_{m+2} - f) /
    (f_{m+2} - f_{m+1}))) on all_freqs = linspace(0, sr // 2, n_stft), mel = 2595 log10(1 + f / 700) - evaluated filter by filter
    with scalar arithmetic, against the oracle's vectorised table (and the product's host table, which the CPU suite already ties to
    the oracle bit for bit)."""
    n_freqs, n_mels, sr, f_min, f_max = 257, 80, 16000, 0.0, 8000.0
    mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)        # noqa: E731
    inv = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)        # noqa: E731
    pts = [inv(mel(f_min) + (mel(f_max) - mel(f_min)) * i / (n_mels + 1)) for i in range(n_mels + 2)]
    want = np.zeros((n_freqs, n_mels))
    for fi in range(n_freqs):
        f = (sr // 2) * fi / (n_freqs - 1)
        for m in range(n_mels):
            up = (f - pts[m]) / (pts[m + 1] - pts[m])
            down = (pts[m + 2] - f) / (pts[m + 2] - pts[m + 1])
            want[fi, m] = max(0.0, min(up, down))
    got = O.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sr)
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-12
    # adjacent triangles share their edges: between the first and the last centre the filters sum to one at every bin
    freqs = np.linspace(0, sr // 2, n_freqs)
    inside = (freqs >= pts[1]) & (freqs <= pts[-2])
    assert np.abs(got[inside].sum(1) - 1.0).max() < 1e-12


def test_slaney_mel_scale_and_norm_known_answers():
    """mel_type / norm = SLANEY (spectrum.py:625-626; MelScale mirrors torchaudio / librosa's Slaney tables).  Known answers of the
    Auditory-Toolbox warp: 3 mel = 200 Hz, 1 kHz = 15 mel, 6.4 kHz = 42 mel; area normalisation: every triangle integrates to one
    over Hz.  The product's host table (mindaudio_amd._host) must equal the oracle's."""
    from mindaudio_amd import _host

    assert np.allclose(O.mel_to_hz_slaney([1, 2, 3, 4, 5]), [200.0 / 3, 400.0 / 3, 200.0, 800.0 / 3, 1000.0 / 3])
    assert np.allclose(O.hz_to_mel_slaney([110.0, 220.0, 440.0, 1000.0, 6400.0]), [1.65, 3.3, 6.6, 15.0, 42.0])
    f = np.linspace(1.0, 8000.0, 97)
    assert np.abs(O.mel_to_hz_slaney(O.hz_to_mel_slaney(f)) - f).max() < 1e-9
    n_freqs, sr = 2049, 16000
    fb = O.melscale_fbanks(n_freqs, 0.0, 8000.0, 40, sr, norm="slaney", mel_type="slaney")
    df = (sr // 2) / (n_freqs - 1)
    assert np.abs(fb.sum(0) * df - 1.0).max() < 2e-2                       # unit area (trapezoid error of a sampled triangle)
    plain = O.melscale_fbanks(n_freqs, 0.0, 8000.0, 40, sr, norm="none", mel_type="slaney")
    assert np.abs(plain.max(0) - 1.0).max() < 5e-2 and np.all(plain >= 0)
    for norm in ("none", "slaney"):
        for mt in ("htk", "slaney"):
            a = O.melscale_fbanks(257, 20.0, 7600.0, 80, sr, norm, mt)
            b = _host.mel_fbanks_f64(257, 20.0, 7600.0, 80, sr, norm, mt)
            assert np.abs(a - b).max() < 1e-12
    assert _host.mel_enum("NormType.SLANEY", "norm") == "slaney" and _host.mel_enum("HTK", "mel_type") == "htk"
    with pytest.raises(ValueError):
        _host.mel_enum("l2", "norm")


def test_fbank_against_direct_dft_definition():
    """features.fbank = 10 log10(max(mel(|STFT|^2), 1e-10)) with the batch-global top_db floor: the power spectrogram computed from the
    DFT DEFINITION (an explicit complex exponential matrix, no FFT routine, reflect-padded periodic Hann frames), float64."""
    rng = np.random.RandomState(11)
    x = 0.1 * rng.randn(2, 4000)
    n_fft, hop, n_mels = 512, 160, 40
    pad = np.pad(x, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)          # periodic Hann
    n_frames = 1 + x.shape[1] // hop
    frames = np.stack([pad[:, t * hop:t * hop + n_fft] * win for t in range(n_frames)], 1)      # (B, T, n_fft)
    dft = np.exp(-2j * np.pi * np.outer(np.arange(n_fft // 2 + 1), np.arange(n_fft)) / n_fft)  # (257, 512)
    power = np.abs(frames @ dft.T) ** 2                                                         # (B, T, 257)
    mel = power @ O.melscale_fbanks(n_fft // 2 + 1, 0.0, 8000.0, n_mels, 16000)                 # (B, T, n_mels)
    db = 10.0 * np.log10(np.maximum(mel, 1e-10))
    db = np.maximum(db, db.max() - 80.0).transpose(0, 2, 1)
    got = O.fbank(x, n_mels=n_mels, n_fft=n_fft, hop_length=hop)
    assert got.shape == db.shape and np.abs(got - db).max() < 1e-6
