"""speed perturbation = mindaudio.data.processing.resample(res_type="fft") = scipy.signal.resample
(processing.py:132-176, examples/conformer/dataset.py:398-406).  scipy is what the reference calls, and it is importable here,
so the oracle and the device path are pinned against it directly."""
import numpy as np
import pytest
import scipy.signal

from oracle import speech_features as O


def _wave(n, seed):
    rng = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    return 0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * rng.randn(n) * (1 + np.sin(2 * np.pi * 3 * t))


@pytest.mark.parametrize("n,m", [(1000, 1112), (1001, 910), (4000, 4000), (777, 1555), (1024, 512), (95984, 87259)])
def test_oracle_resample_vs_scipy(n, m):
    x = _wave(n, n)
    assert np.abs(O.resample_fft(x, m) - scipy.signal.resample(x, m)).max() <= 1e-12


def test_resampled_length_matches_reference_arithmetic():
    from mindaudio_amd.data.processing import resampled_length

    # processing.py:164-166 with speed_perturb's arguments (sample_rate * speed, sample_rate)
    for n in (95984, 16000, 160000, 12345):
        for speed in (0.9, 1.1):
            assert resampled_length(n, 16000 * speed, 16000) == int(np.ceil(n * (float(16000) / (16000 * speed))))


@pytest.mark.gpu
@pytest.mark.parametrize("log2l", [11, 14, 19])
def test_device_fft_pow2_vs_numpy(log2l):
    import ctypes

    import torch

    from mindaudio_amd import _host, _lib

    lib = _lib.load()
    L, b = 1 << log2l, 3
    rng = np.random.RandomState(log2l)
    z = (rng.randn(b, L) + 1j * rng.randn(b, L)).astype(np.complex64)
    for inverse in (0, 1):
        d = torch.from_numpy(z).cuda()
        tmp = torch.empty_like(d)
        res = ctypes.c_void_p()
        _lib.check(lib.ma_fft_pow2_c32(_host.ptr(torch.view_as_real(d)), _host.ptr(torch.view_as_real(tmp)), b, L, inverse,
                                       ctypes.byref(res), _host.current_stream_ptr()), "fft")
        out = d if res.value == d.data_ptr() else tmp
        assert res.value in (d.data_ptr(), tmp.data_ptr())
        want = np.fft.ifft(z.astype(np.complex128), axis=-1) * L if inverse else np.fft.fft(z.astype(np.complex128), axis=-1)
        err = np.abs(out.cpu().numpy() - want).max() / np.abs(want).max()
        assert err <= 3e-6, (log2l, inverse, err)


@pytest.mark.gpu
def test_device_resample_vs_scipy(sample_wav):
    import torch

    from mindaudio_amd.data import processing

    # single utterances through the reference-shaped entry point: speed 0.9 and 1.1 on the sample wav, odd / even lengths
    for x, orig in ((sample_wav, 16000 * 0.9), (sample_wav, 16000 * 1.1), (_wave(4001, 1), 17600.000000000002),
                    (_wave(4000, 2), 14400.0), (_wave(3000, 3), 8000)):
        m = processing.resampled_length(x.shape[0], orig, 16000)
        want = scipy.signal.resample(x, m)
        got = processing.resample(x, orig, 16000)
        assert got.shape == want.shape and got.dtype == x.dtype
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), (x.shape, orig, np.abs(got - want).max())
    assert processing.resample(sample_wav, 16000, 16000) is sample_wav  # processing.py:161-162
    # ragged batch, one call
    lens = [16000, 12345, 9999, 16001]
    n_out = [17778, 11223, 11110, 14547]
    host = np.zeros((4, 16004), np.float32)
    for i, n in enumerate(lens):
        host[i, :n] = _wave(n, 10 + i) * 32768.0
    y = processing.resample_batch(torch.from_numpy(host).cuda(), lens, n_out).cpu().numpy()
    for i, (n, m) in enumerate(zip(lens, n_out)):
        want = scipy.signal.resample(host[i, :n].astype(np.float64), m)
        assert np.abs(y[i, :m] - want).max() <= 2e-5 * np.abs(want).max()
        assert not y[i, m:].any()


@pytest.mark.gpu
def test_speed_perturb_in_collate(tmp_path):
    """CollateFunc(use_speed_perturb=True): lengths follow the drawn speeds and the features are those of the resampled waves."""
    import random

    import torch

    from mindaudio_amd.conformer import dataset as D

    waves = [_wave(16000 + 800 * i, 20 + i) * 0.5 for i in range(4)]
    speeds = [0.9, 1.0, 1.1, 0.9]
    out = D.speed_perturb_batch(waves, 16000, torch.device("cuda", 0), speeds)
    for w, s, o in zip(waves, speeds, out):
        if s == 1.0:
            assert o is w
        else:
            m = int(np.ceil(w.shape[0] * (float(16000) / (16000 * s))))
            want = scipy.signal.resample(w, m)
            assert o.shape == (m,) and np.abs(o - want).max() <= 2e-5 * np.abs(want).max()
    random.seed(3)
    drawn = [random.choice(D.SPEEDS) for _ in range(4)]
    random.seed(3)
    got = D.speed_perturb_batch(waves, 16000, torch.device("cuda", 0))
    assert [g.shape[0] for g in got] == [w.shape[0] if s == 1.0 else int(np.ceil(w.shape[0] * (16000.0 / (16000 * s))))
                                         for w, s in zip(waves, drawn)]
