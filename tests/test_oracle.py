"""CPU: pin the oracle (oracle/speech_features.py) against golden vectors produced by the
imported reference (tests/golden/gen_goldens.py).  Tolerances are written per test."""
import numpy as np
import pytest

from oracle import speech_features as O


def _check_stft(goldens, tag, S):
    assert tuple(goldens[tag + "_shape"]) == S.shape
    assert S.dtype == np.complex64
    cols = goldens[tag + "_cols"]
    want = goldens[tag + "_vals"]
    got = S[..., cols]
    # both sides are float64 pocketfft rounded to complex64: allow 2 ulp of the frame norm
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 4e-7 * scale
    assert abs(np.abs(S).astype(np.float64).sum() - goldens[tag + "_abs_sum"]) <= 1e-6 * goldens[tag + "_abs_sum"]


def test_read_wav(goldens, sample_wav):
    assert sample_wav.shape == (int(goldens["wav_len"]),) and sample_wav.dtype == np.float64
    assert np.array_equal(sample_wav[:64], goldens["wav_head"])
    assert sample_wav.sum() == goldens["wav_sum"]
    assert (sample_wav ** 2).sum() == goldens["wav_sqsum"]


STFT_CASES = {
    "stft_default": dict(),
    "stft_512_160": dict(n_fft=512, hop_length=160),
    "stft_400_160_reflect": dict(n_fft=400, hop_length=160, pad_mode="reflect"),
    "stft_512_win400_hop160": dict(n_fft=512, win_length=400, hop_length=160),
    "stft_nocenter_512_160": dict(n_fft=512, hop_length=160, center=False),
    "stft_hamming_256": dict(n_fft=256, window="hamming"),
    "stft_1024_256": dict(n_fft=1024, hop_length=256),
}


@pytest.mark.parametrize("impl", ["vec", "ref"])
@pytest.mark.parametrize("tag", sorted(STFT_CASES))
def test_stft_sample_wav(goldens, sample_wav, tag, impl):
    fn = O.stft_vec if impl == "vec" else O.stft_ref
    S = fn(sample_wav, **STFT_CASES[tag])
    _check_stft(goldens, tag, S)
    if impl == "ref":
        assert S.flags["F_CONTIGUOUS"] == bool(goldens[tag + "_fortran"])


def test_stft_known_shape(sample_wav):
    # tutorial notebook / README.md:79-83: stft(wav, n_fft=512) -> (257, 750)
    assert O.stft_vec(sample_wav, n_fft=512).shape == (257, 750)


def test_stft_f32_input_and_real_view(goldens, sample_wav):
    _check_stft(goldens, "stft_f32in_512_160", O.stft_vec(sample_wav.astype(np.float32), n_fft=512, hop_length=160))
    ri = O.stft_vec(sample_wav[:4000], n_fft=512, hop_length=160, return_complex=False)
    assert ri.shape == goldens["stft_ri_4000"].shape
    assert np.abs(ri - goldens["stft_ri_4000"]).max() <= 4e-7 * np.abs(goldens["stft_ri_4000"]).max()


@pytest.mark.parametrize("impl", ["vec", "ref"])
def test_stft_synthetic_batch(goldens, impl):
    x = (0.1 * np.random.RandomState(int(goldens["synth_seed"])).randn(4, 160000)).astype(np.float32)
    fn = O.stft_vec if impl == "vec" else O.stft_ref
    _check_stft(goldens, "stft_synth4", fn(x, n_fft=512, hop_length=160))


@pytest.mark.parametrize("n", [512, 513, 671, 672, 673, 1000, 1601])
def test_stft_short_lengths(goldens, n):
    xs = (0.1 * np.random.RandomState(n).randn(2, n)).astype(np.float32)
    want = goldens["stft_short_%d" % n]
    for fn in (O.stft_vec, O.stft_ref):
        got = fn(xs, n_fft=512, hop_length=160)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 4e-7 * np.abs(want).max()


def test_stft_errors(sample_wav):
    with pytest.raises(ValueError):
        O.stft_vec(sample_wav[:300], n_fft=512)
    with pytest.raises(ValueError):
        O.stft_vec(sample_wav, n_fft=512, hop_length=0)
    with pytest.raises(ValueError):
        O.stft_vec(sample_wav, n_fft=256, win_length=400)


def test_frame(goldens):
    x = np.arange(40, dtype=np.float64).reshape(2, 20)
    assert np.array_equal(O.frame_ref(x, 8, 3), goldens["frame_2x20_8_3"])
    assert np.array_equal(O.frame_vec(x, 8, 3), goldens["frame_2x20_8_3"])


@pytest.mark.parametrize("tag", ["db2", "db3", "db4"])
def test_amplitude_to_db(goldens, tag):
    a = goldens[tag + "_in"]
    # same float64 numpy ops in the same order: bit-exact
    assert np.array_equal(O.amplitude_to_dB(a), goldens[tag + "_power"])
    assert np.array_equal(O.amplitude_to_dB(a, stype="magnitude", ref=2.0, top_db=60.0), goldens[tag + "_mag_ref2_top60"])
    assert np.array_equal(O.amplitude_to_dB(a, top_db=None), goldens[tag + "_notop"])


def test_amplitude_to_db_batch_global_floor(goldens):
    a = goldens["db3_in"]
    out = O.amplitude_to_dB(a)
    # quiet utterance (index 1) is floored at (max over the WHOLE batch) - 80, spectrum.py:79-89
    assert out[1].min() == out.max() - 80.0
    assert np.array_equal(O.amplitude_to_dB(a.astype(np.float32)), goldens["db3_f32_power"])
    with pytest.raises(UserWarning):
        O.amplitude_to_dB(a.astype(np.complex64))


def test_kaldi_mel_banks(goldens):
    banks, centres = O.kaldi_mel_banks(80, 512, 16000, 20, 8000)
    assert banks.shape == tuple(goldens["kaldi_mel_shape"])
    dense = np.zeros_like(banks)
    dense[goldens["kaldi_mel_rows"], goldens["kaldi_mel_cols"]] = goldens["kaldi_mel_vals"]
    assert np.array_equal(banks, dense)  # same float64 expression order: bit-exact
    assert np.array_equal(centres, goldens["kaldi_mel_centers"])
    assert np.array_equal(O.kaldi_mel_banks(23, 512, 16000, 20, 8000)[0], goldens["kaldi_mel23_dense"])
    assert np.count_nonzero(banks) == 501


@pytest.mark.parametrize("loop", [False, True])
def test_kaldi_fbank_sample_wav(goldens, sample_wav, loop):
    f = O.compute_fbank_feats(sample_wav * (1 << 15), 16000, 25, 10, 80, per_frame_loop=loop)
    assert f.shape == tuple(goldens["kaldi_wav_shape"]) == (598, 80)
    want = goldens["kaldi_wav_vals"]
    got = f[goldens["kaldi_wav_rows"]]
    # natural-log features ~10; float64 on both sides, summation order of the matmul may differ
    assert np.abs(got - want).max() <= 1e-9
    assert abs(f.sum() - goldens["kaldi_wav_sum"]) <= 1e-6
    assert abs(f.min() - goldens["kaldi_wav_min"]) <= 1e-9 and abs(f.max() - goldens["kaldi_wav_max"]) <= 1e-9


@pytest.mark.parametrize("n", [16000, 12345, 400, 559, 560])
def test_kaldi_fbank_synthetic(goldens, n):
    w = goldens["kaldi_synth_in_%d" % n]
    want = goldens["kaldi_synth_out_%d" % n]
    got = O.compute_fbank_feats(w, 16000, 25, 10, 80)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-9
    assert np.array_equal(O.preemphasis(np.arange(10, dtype=np.float64) ** 2), goldens["preemph_10"])


def test_collate_helpers(goldens):
    seqs = [np.arange(12, dtype=np.float32).reshape(4, 3), np.ones((2, 3), np.float32), 2 * np.ones((3, 3), np.float32)]
    assert np.array_equal(O.pad_sequence(seqs, True, 0.0, 6, np.float32), goldens["pad_sequence_f32"])
    ys = [np.array([1, 2, 3, 4, 5], np.int32), np.array([4, 5, 6], np.int32), np.array([7, 8, 9, 10], np.int32)]
    assert np.array_equal(O.pad_sequence(ys, True, -1, 7, np.int32), goldens["pad_sequence_i32"])
    a, b = O.add_sos_eos(ys, 10, 11)
    assert np.array_equal(np.concatenate(a), goldens["add_sos_eos_in"])
    assert np.array_equal(np.concatenate(b), goldens["add_sos_eos_out"])
    assert np.array_equal(O.make_pad_mask([5, 3, 2]), goldens["make_pad_mask_5_3_2"])
    assert np.array_equal(O.make_pad_mask([5, 3, 2], max_len=8), goldens["make_pad_mask_max8"])
    assert np.array_equal(O.subsequent_mask(5), goldens["subsequent_mask_5"])


# ---- unpinned part: cross-checks of the MindSpore-op restatement ---------------------------
def test_spectrogram_matches_torch_stft(sample_wav):
    torch = pytest.importorskip("torch")
    x = torch.from_numpy(sample_wav[:20000])
    for n_fft, hop, win in ((512, 160, 512), (400, 200, 400), (512, 160, 400)):
        S = torch.stft(x, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win, periodic=True, dtype=torch.float64),
                       center=True, pad_mode="reflect", onesided=True, return_complex=True)
        want = (S.abs() ** 2).numpy()
        got = O.spectrogram(sample_wav[:20000], n_fft=n_fft, win_length=win, hop_length=hop)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-10 * want.max()


def test_melscale_fbanks_properties():
    fb = O.melscale_fbanks(257, 0.0, 8000.0, 80, 16000)
    assert fb.shape == (257, 80) and fb.min() >= 0.0 and fb.max() <= 1.0
    # triangular: every FFT bin feeds at most two adjacent filters; every filter is contiguous
    assert (np.count_nonzero(fb, axis=1) <= 2).all()
    for m in range(80):
        nz = np.nonzero(fb[:, m])[0]
        assert nz.size > 0 and np.array_equal(nz, np.arange(nz[0], nz[-1] + 1))
    # interior partition of unity of HTK triangles with norm=None
    s = fb.sum(axis=1)
    k_lo = np.nonzero(fb[:, 0])[0][-1]
    k_hi = np.nonzero(fb[:, -1])[0][0]
    assert np.allclose(s[k_lo + 1:k_hi], 1.0, atol=1e-12)


def test_fbank_shapes_cfg1(sample_wav):
    # tutorial cell: fbank(wav, n_fft=512) -> (40, 375); cfg-2 arguments on the wav -> (80, 600)
    assert O.fbank(sample_wav, n_fft=512).shape == (40, 375)
    out = O.fbank(sample_wav, n_mels=80, n_fft=512, hop_length=160)
    assert out.shape == (80, 600)
    assert out.min() >= out.max() - 80.0
    batch = np.stack([sample_wav[:16000], 1e-4 * sample_wav[16000:32000]])
    ob = O.fbank(batch, n_mels=80, n_fft=512, hop_length=160)
    assert ob.shape == (2, 80, 101)
    assert ob[1].min() == ob.max() - 80.0  # batch-global floor
    assert np.allclose(O.fbank_ref_cost(batch), ob, rtol=0, atol=2e-4)  # c64 rounding of the R flavour


def test_fbank_default_docstring_shape():
    # features.py:246-249 prints (10, 40, 101) for fbanks(np.random.random([10, 16000])); that figure
    # is stale: the default hop is win_length // 2 = 200 (spectrum.py:666) -> 1 + 16000 // 200 = 81.
    assert O.fbank(np.random.RandomState(0).rand(10, 16000)).shape == (10, 40, 81)
    assert O.fbank(np.random.RandomState(0).rand(10, 16000), hop_length=160).shape == (10, 40, 101)
