"""GPU parity: HIP feature kernels (through the C-ABI mirrors) vs the oracle and the reference goldens.

Tolerances.  The reference computes stft in float64 and rounds to complex64; the device computes in
float32.  A 512-point float32 FFT carries ~1e-7 * ||frame|| absolute error, so parity is norm-wise:
|delta| <= TOL_STFT * max|S| over the frame (north_star: 1e-5 rel fp32).  Mel energies are sums of
non-negative terms: relative 1e-5 of the frame's largest mel energy.  dB / ln outputs are compared with
an absolute tolerance where the energy is above the float32 noise floor of its frame.
"""
import numpy as np
import pytest

from oracle import speech_features as O

pytestmark = pytest.mark.gpu

TOL_STFT = 2e-6  # relative to the per-frame max magnitude (bar: 1e-5)
TOL_MEL = 1e-5   # relative to the per-frame max mel energy


@pytest.fixture(scope="module")
def ma():
    import torch

    assert torch.cuda.is_available(), "gpu-marked tests need a HIP device"
    import mindaudio_amd

    return mindaudio_amd


def _frame_rel_err(got, want):
    """max over frames of max|delta| / max|want| (frames on the last axis, freq on -2)."""
    scale = np.abs(want).max(axis=-2, keepdims=True)
    scale = np.maximum(scale, 1e-30)
    return float((np.abs(got - want) / scale).max())


STFT_CASES = {
    "stft_default": dict(),
    "stft_512_160": dict(n_fft=512, hop_length=160),
    "stft_512_win400_hop160": dict(n_fft=512, win_length=400, hop_length=160),
    "stft_nocenter_512_160": dict(n_fft=512, hop_length=160, center=False),
    # n_fft != 512: exact-f32 MFMA DFT path (features_generic.hip)
    "stft_400_160_reflect": dict(n_fft=400, hop_length=160, pad_mode="reflect"),
    "stft_hamming_256": dict(n_fft=256, window="hamming"),
    "stft_1024_256": dict(n_fft=1024, hop_length=256),
}


@pytest.mark.parametrize("tag", sorted(STFT_CASES))
def test_stft_sample_wav_vs_reference_goldens(ma, goldens, sample_wav, tag):
    S = ma.stft(sample_wav, **STFT_CASES[tag])
    assert S.shape == tuple(goldens[tag + "_shape"]) and S.dtype == np.complex64
    assert S.flags["F_CONTIGUOUS"] == bool(goldens[tag + "_fortran"])
    cols = goldens[tag + "_cols"]
    assert _frame_rel_err(S[..., cols], goldens[tag + "_vals"]) <= TOL_STFT
    full = O.stft_vec(sample_wav, **STFT_CASES[tag])
    assert _frame_rel_err(S, full) <= TOL_STFT
    asum = np.abs(S).astype(np.float64).sum()
    assert abs(asum - goldens[tag + "_abs_sum"]) <= 1e-5 * goldens[tag + "_abs_sum"]


@pytest.mark.parametrize("pad_mode", ["constant", "reflect", "edge", "symmetric"])
def test_stft_pad_modes(ma, sample_wav, pad_mode):
    x = sample_wav[1000:9000]
    got = ma.stft(x, n_fft=512, hop_length=160, pad_mode=pad_mode)
    want = O.stft_vec(x, n_fft=512, hop_length=160, pad_mode=pad_mode)
    assert got.shape == want.shape
    assert _frame_rel_err(got, want) <= TOL_STFT


def test_stft_odd_hop_and_misaligned_rows(ma, sample_wav):
    import torch

    x = sample_wav[:20001]
    got = ma.stft(x, n_fft=512, hop_length=131)
    want = O.stft_vec(x, n_fft=512, hop_length=131)
    assert _frame_rel_err(got, want) <= TOL_STFT
    # odd row stride: second row starts at an odd element offset -> scalar-load path
    xb = torch.from_numpy(np.stack([sample_wav[:5001], sample_wav[7:5008]]).astype(np.float32)).cuda()
    got = ma.stft(xb, n_fft=512, hop_length=160).cpu().numpy()
    want = O.stft_vec(xb.cpu().numpy(), n_fft=512, hop_length=160)
    assert _frame_rel_err(got, want) <= TOL_STFT


def test_stft_synthetic_batch_vs_reference_goldens(ma, goldens):
    x = (0.1 * np.random.RandomState(int(goldens["synth_seed"])).randn(4, 160000)).astype(np.float32)
    S = ma.stft(x, n_fft=512, hop_length=160)
    assert S.shape == (4, 257, 1001)
    cols = goldens["stft_synth4_cols"]
    assert _frame_rel_err(S[..., cols], goldens["stft_synth4_vals"]) <= TOL_STFT
    ri = ma.stft(x[:1, :4000], n_fft=512, hop_length=160, return_complex=False)
    assert ri.shape == (1, 257, 26, 2)
    assert np.array_equal(ri[..., 0], S[:1, :, :26].real[..., :26]) or np.allclose(ri[..., 0], ma.stft(x[:1, :4000], n_fft=512, hop_length=160).real)


@pytest.mark.parametrize("n", [512, 513, 671, 672, 673, 1000, 1601])
def test_stft_short_lengths_vs_reference_goldens(ma, goldens, n):
    xs = (0.1 * np.random.RandomState(n).randn(2, n)).astype(np.float32)
    want = goldens["stft_short_%d" % n]
    got = ma.stft(xs, n_fft=512, hop_length=160)
    assert got.shape == want.shape
    assert _frame_rel_err(got, want) <= TOL_STFT


def test_stft_errors_match_reference(ma, sample_wav):
    with pytest.raises(ValueError):
        ma.stft(sample_wav[:300], n_fft=512)  # spectrum.py:182-187
    with pytest.raises(ValueError):
        ma.stft(sample_wav, n_fft=512, hop_length=0)  # spectrum.py:295-296
    with pytest.raises(ValueError):
        ma.stft(sample_wav, n_fft=512, win_length=600)  # spectrum.py:331-334


def test_stft_device_tensor_roundtrip(ma, sample_wav):
    import torch

    x = torch.from_numpy(sample_wav[:32000].astype(np.float32)).cuda()
    S = ma.stft(x, n_fft=512, hop_length=160)
    assert S.is_cuda and S.dtype == torch.complex64 and tuple(S.shape) == (257, 201)
    want = O.stft_vec(sample_wav[:32000], n_fft=512, hop_length=160)
    assert _frame_rel_err(S.cpu().numpy(), want) <= TOL_STFT


# ---- melspectrogram / fbank ---------------------------------------------------------------
def _speechlike(seed, b, n):
    rng = np.random.RandomState(seed)
    env = np.abs(np.sin(np.linspace(0, 7 * np.pi, n)))[None, :] ** 4
    return (0.1 * rng.randn(b, n) * (1e-4 + env)).astype(np.float32)


def test_melspectrogram_vs_oracle(ma, sample_wav):
    for kw in (dict(n_fft=512, hop_length=160, n_mels=80), dict(n_fft=512, n_mels=40),
               dict(n_fft=512, win_length=400, hop_length=160, n_mels=23, f_min=20.0, f_max=7600.0),
               dict(n_fft=512, hop_length=160, n_mels=80, power=1.0)):
        got = ma.melspectrogram(sample_wav, **kw)
        want = O.melspectrogram(sample_wav, **kw)
        assert got.shape == want.shape and got.dtype == np.float32
        assert _frame_rel_err(got, want) <= TOL_MEL


@pytest.mark.parametrize("norm,mel_type", [("slaney", "slaney"), ("none", "slaney"), ("slaney", "htk")])
def test_melspectrogram_slaney_scale_and_norm(ma, sample_wav, norm, mel_type):
    """spectrum.py:625-626: norm / mel_type other than the defaults (a host-built table, same kernel)."""
    for kw in (dict(n_fft=512, hop_length=160, n_mels=80), dict(n_mels=64)):
        got = ma.melspectrogram(sample_wav, norm=norm, mel_type=mel_type, **kw)
        want = O.melspectrogram(sample_wav, norm=norm, mel_type=mel_type, **kw)
        assert got.shape == want.shape and got.dtype == np.float32
        assert _frame_rel_err(got, want) <= TOL_MEL
    with pytest.raises(ValueError):
        ma.melspectrogram(sample_wav, norm="l2")


def _check_db(got, want_db, mel_energy, atol=2e-3):
    """dB parity where the mel energy is above the float32 noise of its frame; floor equality elsewhere."""
    assert got.shape == want_db.shape
    floor = want_db.max() - 80.0
    assert abs(float(got.max()) - float(want_db.max())) <= 1e-4
    assert got.min() >= floor - 1e-4
    strong = mel_energy >= 1e-4 * mel_energy.max(axis=-2, keepdims=True)
    assert np.abs(got - want_db)[strong].max() <= atol
    # everything, including bins near the float32 noise floor: within 0.05 dB
    assert np.abs(got - want_db).max() <= 5e-2


def test_fbank_cfg1_sample_wav(ma, sample_wav):
    got = ma.fbank(sample_wav, n_fft=512)  # tutorial cell: (40, 375)
    assert got.shape == (40, 375)
    _check_db(got, O.fbank(sample_wav, n_fft=512), O.melspectrogram(sample_wav, n_fft=512, n_mels=40))
    kw = dict(n_mels=80, n_fft=512, hop_length=160)
    got = ma.fbank(sample_wav, **kw)
    assert got.shape == (80, 600)
    _check_db(got, O.fbank(sample_wav, **kw), O.melspectrogram(sample_wav, **kw))
    assert ma.fbanks is ma.fbank


def test_fbank_batch_global_floor_and_tile_skipping(ma):
    kw = dict(n_mels=80, n_fft=512, hop_length=160)
    x = _speechlike(5, 6, 48000)
    x[3] *= 1e-5  # one very quiet utterance: floored relative to the LOUDEST utterance of the batch
    got = ma.fbank(x, **kw)
    want = O.fbank(x, **kw)
    _check_db(got, want, O.melspectrogram(x, **kw))
    assert got[3].min() == pytest.approx(got.max() - 80.0, abs=1e-4)
    # per-utterance calls differ from the batched call (reference semantics, spectrum.py:79-89)
    alone = ma.fbank(x[3], **kw)
    assert alone.min() < got[3].min() - 10.0
    # white noise: no tile needs the floor; result must equal the un-floored dB
    noise = (0.1 * np.random.RandomState(1234).randn(3, 16000)).astype(np.float32)
    g2 = ma.fbank(noise, **kw)
    _check_db(g2, O.fbank(noise, **kw), O.melspectrogram(noise, **kw))


def test_fbank_channel_input_groups(ma):
    x = _speechlike(9, 6, 16000).reshape(2, 3, 16000)
    x[1] *= 1e-4
    kw = dict(n_mels=40, n_fft=512, hop_length=160)
    got = ma.fbank(x, **kw)
    want = O.fbank(x, **kw)
    assert got.shape == want.shape == (2, 3, 40, 101)
    # floor is per batch entry here (channels = shape[-3])
    for b in range(2):
        assert got[b].min() >= got[b].max() - 80.0 - 1e-4
    assert np.abs(got - want).max() <= 5e-2


def test_fbank_device_tensor_and_unsupported(ma):
    import torch

    x = torch.from_numpy(_speechlike(2, 2, 16000)).cuda()
    out = ma.fbank(x, n_fft=512, n_mels=40)
    assert out.is_cuda and tuple(out.shape) == (2, 40, 63)  # default hop = n_fft // 2 (spectrum.py:666)
    d = ma.fbank(x, n_fft=512, n_mels=40, deltas=True)  # features.py:264-267: static + delta + delta-delta
    assert d.is_cuda and tuple(d.shape) == (2, 120, 63) and torch.equal(d[:, :40], out)
    with pytest.raises(ValueError):
        ma.fbank(x[:, :300], n_fft=512)


# ---- generic n_fft (reference defaults: fbank n_fft=400, features.py:201) ------------------------------
@pytest.mark.parametrize("n_fft,hop,kw", [(400, 160, dict()), (400, None, dict(center=False)), (256, 100, dict(window="hamming")),
                                           (1024, 256, dict(pad_mode="edge")), (62, 16, dict()), (130, 33, dict(win_length=100))])
def test_stft_generic_nfft_vs_oracle(ma, n_fft, hop, kw):
    x = _speechlike(n_fft, 3, 9001)
    got = ma.stft(x, n_fft=n_fft, hop_length=hop, **kw)
    want = O.stft_vec(x, n_fft=n_fft, hop_length=hop, **kw)
    assert got.shape == want.shape
    assert _frame_rel_err(got, want) <= TOL_STFT


def test_fbank_reference_defaults_nfft400(ma, sample_wav):
    # features.fbank docstring example (features.py:247-251): inputs (10, 16000) -> (10, 40, 81) at defaults
    x = _speechlike(3, 10, 16000)
    got = ma.fbank(x)
    assert got.shape == (10, 40, 81)
    _check_db(got, O.fbank(x), O.melspectrogram(x, n_fft=400, n_mels=40))
    # ECAPA call site shape (examples/ECAPA-TDNN/speaker_verification_cosine.py:53): n_mels=80 at the default n_fft
    got = ma.fbank(sample_wav, n_mels=80)
    want = O.fbank(sample_wav, n_mels=80)
    assert got.shape == want.shape == (80, 1 + len(sample_wav) // 200)
    _check_db(got, want, O.melspectrogram(sample_wav, n_fft=400, n_mels=80))
    for kw in (dict(n_fft=400, hop_length=160, n_mels=80), dict(n_fft=1024, hop_length=256, n_mels=128),
               dict(n_fft=256, n_mels=23, power=1.0)):
        m = ma.melspectrogram(sample_wav, **kw)
        assert _frame_rel_err(m, O.melspectrogram(sample_wav, **kw)) <= TOL_MEL


# ---- amplitude_to_dB vs reference goldens ---------------------------------------------------
@pytest.mark.parametrize("tag", ["db2", "db3", "db4"])
def test_amplitude_to_db_vs_reference_goldens(ma, goldens, tag):
    a = goldens[tag + "_in"]
    for key, kw in (("_power", {}), ("_mag_ref2_top60", dict(stype="magnitude", ref=2.0, top_db=60.0)),
                    ("_notop", dict(top_db=None))):
        got = ma.amplitude_to_dB(a, **kw)
        want = goldens[tag + key]
        assert got.shape == want.shape and got.dtype == a.dtype
        # float32 log10 on the device vs float64: 1e-5 relative on |dB| <= ~100 -> 1e-3 absolute is generous
        assert np.abs(got - want).max() <= 2e-4
    with pytest.raises(UserWarning):
        ma.amplitude_to_dB(a.astype(np.complex64))


# ---- Kaldi-style fbank of the Conformer loader ----------------------------------------------
def _check_ln(got, want, atol=2e-3):
    assert got.shape == want.shape
    e = np.exp(want)
    strong = e >= 1e-4 * e.max(axis=-1, keepdims=True)
    assert np.abs(got - want)[strong].max() <= atol
    assert np.abs(got - want).max() <= 5e-2


def test_kaldi_fbank_sample_wav_vs_reference_goldens(ma, goldens, sample_wav):
    from mindaudio_amd.conformer.dataset import compute_fbank_feats

    f = compute_fbank_feats(sample_wav * (1 << 15), 16000, 25, 10, 80)
    assert f.shape == (598, 80)
    _check_ln(f[goldens["kaldi_wav_rows"]], goldens["kaldi_wav_vals"])
    assert abs(f.sum() - goldens["kaldi_wav_sum"]) <= 1e-5 * abs(goldens["kaldi_wav_sum"])


@pytest.mark.parametrize("n", [16000, 12345, 400, 559, 560])
def test_kaldi_fbank_synthetic_vs_reference_goldens(ma, goldens, n):
    from mindaudio_amd.conformer.dataset import compute_fbank_feats

    got = compute_fbank_feats(goldens["kaldi_synth_in_%d" % n], 16000, 25, 10, 80)
    _check_ln(got, goldens["kaldi_synth_out_%d" % n])


def test_kaldi_fbank_ragged_batch(ma, goldens):
    from mindaudio_amd.conformer.dataset import compute_fbank_feats_batch

    lens = [16000, 12345, 400, 559, 560]
    wavs = np.zeros((len(lens), 16000), np.float32)
    for i, n in enumerate(lens):
        wavs[i, :n] = goldens["kaldi_synth_in_%d" % n]
        wavs[i, n:] = 12345.0  # garbage past the end must not leak into the features
    out, frames = compute_fbank_feats_batch(wavs, lens)
    out = out.cpu().numpy()
    assert out.shape == (5, 98, 80)
    for i, n in enumerate(lens):
        want = goldens["kaldi_synth_out_%d" % n]
        assert int(frames[i]) == want.shape[0]
        _check_ln(out[i, :want.shape[0]], want)
        assert not out[i, want.shape[0]:].any()  # zero rows: pad_sequence padding (dataset.py:563-569)


def test_kaldi_fbank_batch_with_utterances_shorter_than_a_frame(ma, goldens):
    """Lengths below one 25 ms frame (and 0) give no frames: zero rows, frame count 0, and do not disturb their neighbours."""
    from mindaudio_amd.conformer.dataset import compute_fbank_feats_batch

    lens = [399, 12345, 0, 400]
    wavs = np.full((len(lens), 12345), 777.0, np.float32)
    wavs[1] = goldens["kaldi_synth_in_12345"]
    wavs[3, :400] = goldens["kaldi_synth_in_400"]
    out, frames = compute_fbank_feats_batch(wavs, lens)
    out = out.cpu().numpy()
    assert [int(f) for f in frames] == [0, goldens["kaldi_synth_out_12345"].shape[0], 0, 1]
    assert not out[0].any() and not out[2].any()
    _check_ln(out[1, :int(frames[1])], goldens["kaldi_synth_out_12345"])
    _check_ln(out[3, :1], goldens["kaldi_synth_out_400"])
    assert not out[3, 1:].any()


# ---- full cfg-2 size: properties + sampled oracle rows ---------------------------------------
def test_cfg2_full_size_properties(ma):
    import torch

    rng = np.random.RandomState(1234)
    x = (0.1 * rng.randn(64, 160000)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    kw = dict(n_mels=80, n_fft=512, hop_length=160)
    out = ma.fbank(xd, **kw)
    assert tuple(out.shape) == (64, 80, 1001)
    o = out.cpu().numpy()
    assert np.isfinite(o).all() and o.min() >= o.max() - 80.0 - 1e-4
    # sampled rows against the oracle (floor taken from the device's global max: batch-global semantics)
    for b in (0, 17, 63):
        mel = O.melspectrogram(x[b], **kw)
        db = 10.0 * np.log10(np.maximum(mel, 1e-10))
        db = np.maximum(db, o.max() - 80.0)
        assert np.abs(o[b] - db).max() <= 5e-2
        strong = mel >= 1e-4 * mel.max(axis=-2, keepdims=True)
        assert np.abs(o[b] - db)[strong].max() <= 2e-3
    # batching invariance of the un-floored path: row b of the batched melspectrogram == single call
    m_all = ma.melspectrogram(xd, **kw)
    m_one = ma.melspectrogram(xd[5:6], **kw)
    assert torch.equal(m_all[5:6], m_one)
    # linearity of the STFT: stft(a x + b y) = a stft(x) + b stft(y)
    S = ma.stft(xd[:4], n_fft=512, hop_length=160)
    S2 = ma.stft(2.0 * xd[:2] - 3.0 * xd[2:4], n_fft=512, hop_length=160)
    lin = 2.0 * S[:2] - 3.0 * S[2:4]
    assert float((S2 - lin).abs().max() / lin.abs().max()) <= 1e-5
    # Parseval per frame (periodic Hann, full frame): sum |x w|^2 = (|X0|^2 + 2 sum |Xk|^2 + |X256|^2) / 512
    fr = O.frame_vec(np.pad(x[0].astype(np.float64), (256, 256)), 512, 160)[:, 10:20]
    w = O._centered_window("hann", 512, 512)[:, None]
    lhs = ((fr * w) ** 2).sum(axis=0)
    P = (S[0, :, 10:20].abs() ** 2).double().cpu().numpy()
    rhs = (P[0] + 2.0 * P[1:256].sum(axis=0) + P[256]) / 512.0
    assert np.abs(lhs - rhs).max() <= 1e-5 * lhs.max()
