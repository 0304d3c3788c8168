"""magphase / spectrogram / compute_deltas / context_window / mfcc / CMVN statistics.
Pinned by reference outputs where the reference code is NumPy (tests/golden/post_goldens.npz); otherwise vs the oracle."""
import json
import os

import numpy as np
import pytest

from oracle import speech_features as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def pg():
    return np.load(os.path.join(HERE, "golden", "post_goldens.npz"))


def test_oracle_magphase_and_cmvn_vs_reference(pg):
    D = pg["magphase_in"]
    for pw in (1.0, 2.0, 0.5):
        mag, ph = O.magphase(D, pw)
        assert np.allclose(mag, pg["magphase_mag_%g" % pw], rtol=1e-6, atol=0)
        assert np.array_equal(ph, pg["magphase_phase_%g" % pw])
    mean, istd = O.load_cmvn_stats(pg["cmvn_mean_stat"], pg["cmvn_var_stat"], int(pg["cmvn_frames"]))
    assert np.allclose(mean, pg["cmvn_mean"], rtol=1e-12) and np.allclose(istd, pg["cmvn_istd"], rtol=1e-12)


def test_product_load_cmvn_vs_reference(pg, tmp_path):
    from mindaudio_amd.utils.load_files import load_cmvn

    path = str(tmp_path / "global_cmvn")
    with open(path, "w") as f:
        f.write(json.dumps({"mean_stat": pg["cmvn_mean_stat"].tolist(), "var_stat": pg["cmvn_var_stat"].tolist(),
                            "frame_num": int(pg["cmvn_frames"])}))
    mean, istd = load_cmvn(path, True)
    assert np.allclose(mean, pg["cmvn_mean"], rtol=1e-12) and np.allclose(istd, pg["cmvn_istd"], rtol=1e-12)


def test_oracle_context_window_and_deltas_shapes():
    x = np.random.RandomState(0).randn(2, 5, 9).astype(np.float32)
    c = O.context_window(x, 2, 2)
    assert c.shape == (2, 25, 9)
    assert np.array_equal(c[:, 2::5], x)  # the centre tap is the input itself
    assert np.array_equal(c[:, 0::5, 2:], x[:, :, :-2]) and not c[:, 0::5, :2].any()
    d = O.compute_deltas(np.arange(10, dtype=np.float64)[None, None, :] * 3.0)
    assert np.allclose(d[0, 0, 2:-2], 3.0)  # the delta of a ramp is its slope


@pytest.mark.gpu
def test_device_magphase_and_spectrogram(pg, sample_wav):
    import mindaudio_amd as ma

    D = pg["magphase_in"]
    for pw in (1.0, 2.0, 0.5):
        mag, ph = ma.magphase(D, pw)
        want = pg["magphase_mag_%g" % pw]
        assert np.abs(mag - want).max() <= 2e-5 * want.max()
        assert np.abs(ph - pg["magphase_phase_%g" % pw]).max() <= 2e-6
        assert ph[3, 5] == 1.0 + 0.0j
    x = sample_wav[:16000]
    for kw in (dict(), dict(n_fft=512, hop_length=160, power=1.0), dict(n_fft=256, pad=10)):
        got = ma.spectrogram(x, **kw)
        want = O.spectrogram(x, **kw)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-5 * want.max()


@pytest.mark.gpu
def test_device_deltas_context_mfcc_fbank_deltas():
    import mindaudio_amd as ma

    rng = np.random.RandomState(1)
    spec = rng.randn(2, 3, 20, 57).astype(np.float32)
    for win, mode in ((5, "edge"), (7, "reflect"), (3, "constant"), (9, "symmetric")):
        got = ma.compute_deltas(spec, win, mode)
        assert np.abs(got - O.compute_deltas(spec, win, mode)).max() <= 1e-5
    for l, r in ((5, 5), (0, 0), (2, 4), (4, 1)):
        for x in (spec[0, 0], spec[0], spec):
            assert np.array_equal(ma.context_window(x, l, r), O.context_window(x, l, r))
    x = (0.1 * rng.randn(4, 16000)).astype(np.float32)
    got = ma.mfcc(x)
    want = O.mfcc(x)
    assert got.shape == want.shape == (4, 660, 81)
    assert np.abs(got - want).max() <= 2e-2 * np.abs(want).max()
    got = ma.mfcc(x, deltas=False, context=False, n_mels=40, n_mfcc=13, n_fft=512, hop_length=160, norm="none")
    want = O.mfcc(x, deltas=False, context=False, n_mels=40, n_mfcc=13, n_fft=512, hop_length=160, norm="none")
    assert got.shape == (4, 13, 101) and np.abs(got - want).max() <= 2e-3 * np.abs(want).max()
    fb = ma.fbank(x, deltas=True, context=True, n_mels=40, n_fft=512)  # features.py:264-270
    base = ma.fbank(x, n_mels=40, n_fft=512)
    d1 = O.compute_deltas(base)
    want = O.context_window(np.concatenate((base, d1, O.compute_deltas(d1)), axis=-2), 5, 5)
    assert fb.shape == (4, 40 * 3 * 11, 63) and np.abs(fb - want).max() <= 1e-3


@pytest.mark.gpu
def test_device_cmvn_stats_vs_reference(pg):
    from mindaudio_amd.conformer.compute_cmvn_stats import compute_cmvn_stats

    wav = os.path.join(HERE, "golden", "BAC009S0002W0122.wav")
    info = compute_cmvn_stats([wav, wav, wav], batch_size=2)
    assert info["frame_num"] == 3 * int(pg["cmvn_frames"])
    assert np.allclose(np.array(info["mean_stat"]), 3 * pg["cmvn_mean_stat"], rtol=2e-5)
    assert np.allclose(np.array(info["var_stat"]), 3 * pg["cmvn_var_stat"], rtol=2e-5)


# ---- spectrum.istft (spectrum.py:346-474), pinned by the reference run here (tests/golden/istft_goldens.npz) -----------------
ISTFT_CASES = {"default": dict(), "h160": dict(hop_length=160), "nocenter": dict(hop_length=64, center=False),
               "len5000": dict(hop_length=160, length=5000), "len7000": dict(hop_length=160, length=7000),
               "win400": dict(win_length=400, hop_length=100, window="hamming"), "n400": dict(hop_length=100),
               "batch": dict(hop_length=160)}


@pytest.fixture(scope="module")
def ig():
    return np.load(os.path.join(HERE, "golden", "istft_goldens.npz"))


def test_oracle_istft_vs_reference(ig):
    for tag, kw in ISTFT_CASES.items():
        y = O.istft(ig["spec_" + tag], **kw)
        assert y.shape == ig["y_" + tag].shape and y.dtype == np.float64
        assert np.abs(y - ig["y_" + tag]).max() <= 1e-12
    # the reference's round trip (tests/test_spectrum.py:38-41): istft(stft(x)) == x away from the edges
    assert np.abs(ig["y_h160"][256:-256] - ig["wave"][256:256 + ig["y_h160"].size - 512]).max() <= 1e-6


@pytest.mark.gpu
def test_device_istft_vs_reference(ig):
    import torch

    import mindaudio_amd as ma

    for tag, kw in ISTFT_CASES.items():
        want = ig["y_" + tag]
        got = ma.istft(ig["spec_" + tag], **kw)
        assert got.shape == want.shape and got.dtype == np.float64
        scale = max(np.abs(want).max(), 1e-3)
        # float32 arithmetic (the reference accumulates in float64): 2e-6 of the signal scale where the window sum-square the
        # sample is divided by is O(1); where it is tiny (first / last samples without centring, the un-trimmed tail when
        # `length` exceeds the signal) the quotient y / wss ~ (w f) / w^2 amplifies the round-off of f by 1 / w = wss^-1/2
        _, wss = O.istft(ig["spec_" + tag], return_wss=True, **kw)
        tol = 4e-6 * scale / np.sqrt(np.clip(wss, 1e-9, 1.0))
        assert (np.abs(got - want) <= tol).all(), tag
    # device tensors stay on the device; stft -> istft round trip through our own kernels
    x = torch.from_numpy(ig["wave"].astype(np.float32)).cuda()
    y = ma.istft(ma.stft(x, n_fft=512, hop_length=160), hop_length=160)
    assert y.is_cuda and y.dtype == torch.float32 and y.shape[-1] == 5920
    assert float((y[256:-256] - x[256:5920 - 256]).abs().max()) <= 2e-6
    with pytest.raises(ValueError):
        ma.istft(ig["spec_h160"], n_fft=400)


@pytest.mark.gpu
def test_frame_vs_reference_golden():
    """spectrum.frame on the device against the reference's own output (tests/golden/reference_goldens.npz, frame_2x20_8_3)
    and against the oracle on larger inputs; float64 out whatever the input dtype (spectrum.py:299)."""
    import torch

    import mindaudio_amd as ma

    gold = np.load(os.path.join(HERE, "golden", "reference_goldens.npz"))
    x = np.arange(40, dtype=np.float64).reshape(2, 20)
    got = ma.frame(x, 8, 3)
    assert got.dtype == np.float64 and np.array_equal(got, gold["frame_2x20_8_3"])
    rng = np.random.RandomState(3)
    for shape, fl, hop, dt in (((3, 1000), 400, 160, np.float32), ((700,), 512, 128, np.float64), ((2, 2, 333), 64, 7, np.float32)):
        xx = rng.randn(*shape).astype(dt)
        got = ma.frame(xx, fl, hop)
        assert got.dtype == np.float64 and np.array_equal(got, O.frame_ref(xx, fl, hop))
    dev = ma.frame(torch.from_numpy(x).cuda(), 8, 3)
    assert dev.is_cuda and dev.dtype == torch.float64 and np.array_equal(dev.cpu().numpy(), gold["frame_2x20_8_3"])
    with pytest.raises(ValueError):
        ma.frame(x, 8, 0)
    with pytest.raises(ValueError):
        ma.frame(x, 30, 3)


@pytest.mark.gpu
def test_magphase_real_pairs_normalized_spectrogram_and_log_mfcc():
    import mindaudio_amd as ma

    rng = np.random.RandomState(5)
    z = rng.randn(3, 17, 9, 2).astype(np.float32)
    z[0, 0, 0] = 0.0
    mag, ang = ma.magphase(z, 2.0, iscomplex=False)
    zc = z[..., 0].astype(np.float64) + 1j * z[..., 1].astype(np.float64)
    assert np.allclose(mag, np.abs(zc) ** 2, rtol=1e-5, atol=1e-7) and np.allclose(ang, np.angle(zc), atol=2e-6)
    x = (0.1 * rng.randn(2, 4000)).astype(np.float32)
    plain = ma.spectrogram(x, n_fft=512, hop_length=160)
    norm = ma.spectrogram(x, n_fft=512, hop_length=160, normalized=True)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(512) / 512)
    assert np.allclose(norm, plain / np.sum(w * w), rtol=1e-5, atol=1e-12)  # power 2: |S / sqrt(sum w^2)|^2
    got = ma.mfcc(x, deltas=False, context=False, n_mels=23, n_mfcc=20, n_fft=512, log_mels=True)
    mel = O.melspectrogram(x, n_fft=512, n_mels=23)
    want = np.matmul(np.log(mel + 1e-6).transpose(0, 2, 1), O.create_dct(20, 23, "ortho")).transpose(0, 2, 1)
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-3 * np.abs(want).max()
