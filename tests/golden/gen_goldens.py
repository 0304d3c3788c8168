#!/usr/bin/env python3
"""Generate golden vectors by importing the reference's NumPy code in THIS container.

Runs only where /root/reference exists (the build container). Nothing from the
reference travels: this script executes the reference functions on inputs and
stores inputs + outputs (data) as small .npz fixtures under tests/golden/.

`mindspore` is not installed here, and every reference module imports it at top
level, so a throw-away stub is registered in sys.modules first (attribute access
only; no arithmetic goes through the stub).  The functions executed are the pure
NumPy/SciPy ones:

  mindaudio/data/io.py::read                     (io.py:552)
  mindaudio/data/spectrum.py::stft               (spectrum.py:125-278)
  mindaudio/data/spectrum.py::frame              (spectrum.py:281-304)
  mindaudio/data/spectrum.py::amplitude_to_dB    (spectrum.py:25-90)
  examples/conformer/dataset.py::get_mel_banks / compute_fbank_feats (dataset.py:68-168)
  mindaudio/utils/common.py::pad_sequence / add_sos_eos  (common.py:10-88)
  mindaudio/utils/mask.py::make_pad_mask / subsequent_mask (mask.py:19-67)

melspectrogram / features.fbank run inside MindSpore C++ and cannot be executed
here: no golden vectors exist for them ("parity unpinned").

usage: python tests/golden/gen_goldens.py
"""
import importlib.util
import os
import shutil
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class _Anything:
    """Attribute sink: any attribute / call returns another sink."""

    def __init__(self, name="stub"):
        self._name = name

    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything(self._name + "." + k)

    def __call__(self, *a, **k):
        return _Anything(self._name + "()")


class _StubModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything(self.__name__ + "." + k)


def _install_stubs():
    names = [
        "mindspore",
        "mindspore.nn",
        "mindspore.ops",
        "mindspore.dataset",
        "mindspore.dataset.audio",
        "mindspore.dataset.audio.utils",
        "mindspore.dataset.engine",
        "mindspore.common",
        "mindspore.common.dtype",
    ]
    for n in names:
        m = _StubModule(n)
        m.__path__ = []
        sys.modules[n] = m
    for n in names:
        if "." in n:
            parent, child = n.rsplit(".", 1)
            setattr(sys.modules[parent], child, sys.modules[n])
    # isinstance(x, ms.Tensor) is evaluated inside spectrum.frame
    sys.modules["mindspore"].Tensor = type("Tensor", (), {"__init__": lambda self, *a, **k: None})


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    _install_stubs()
    io = _load("ref_io", "mindaudio/data/io.py")
    spectrum = _load("ref_spectrum", "mindaudio/data/spectrum.py")
    # fake `mindaudio` package so examples/conformer/dataset.py imports resolve
    pkg = types.ModuleType("mindaudio")
    pkg.__path__ = []
    sys.modules["mindaudio"] = pkg
    utils = types.ModuleType("mindaudio.utils")
    utils.__path__ = []
    sys.modules["mindaudio.utils"] = utils
    pkg.utils = utils
    pkg.read = io.read
    common = _load("mindaudio.utils.common", "mindaudio/utils/common.py")
    dist = _load("mindaudio.utils.distributed", "mindaudio/utils/distributed.py")
    logm = types.ModuleType("mindaudio.utils.log")
    import logging

    logm.get_logger = lambda *a, **k: logging.getLogger("ref")
    sys.modules["mindaudio.utils.log"] = logm
    mask = _load("mindaudio.utils.mask", "mindaudio/utils/mask.py")
    dataset = _load("ref_conformer_dataset", "examples/conformer/dataset.py")
    return dict(io=io, spectrum=spectrum, common=common, mask=mask, dataset=dataset, dist=dist)


def _cols(n_frames):
    """Columns (frames) kept from a big STFT: edges + a sparse interior sample."""
    idx = sorted(set([0, 1, 2, 3, 4, n_frames - 5, n_frames - 4, n_frames - 3, n_frames - 2, n_frames - 1]
                     + list(range(7, n_frames, 53))))
    return np.array([i for i in idx if 0 <= i < n_frames], dtype=np.int64)


def main():
    ref = load_reference()
    sp, io, ds = ref["spectrum"], ref["io"], ref["dataset"]

    wav_src = os.path.join(REF, "tests/samples/ASR/BAC009S0002W0122.wav")
    wav_dst = os.path.join(HERE, "BAC009S0002W0122.wav")
    if not os.path.exists(wav_dst):
        shutil.copyfile(wav_src, wav_dst)  # data file held by the reference's own tests
    wav, sr = io.read(wav_src)
    assert sr == 16000 and wav.shape == (95984,) and wav.dtype == np.float64

    out = {}
    out["wav_sr"] = np.int64(sr)
    out["wav_len"] = np.int64(wav.shape[0])
    out["wav_head"] = wav[:64].copy()
    out["wav_sum"] = np.float64(wav.sum())
    out["wav_sqsum"] = np.float64((wav**2).sum())

    # ---- stft on the sample wav (cfg 1) -----------------------------------
    def keep(tag, S):
        S = np.asarray(S)
        assert S.dtype == np.complex64
        c = _cols(S.shape[-1])
        out[tag + "_shape"] = np.array(S.shape, dtype=np.int64)
        out[tag + "_cols"] = c
        out[tag + "_vals"] = np.ascontiguousarray(S[..., c])
        out[tag + "_abs_sum"] = np.float64(np.abs(S).astype(np.float64).sum())
        out[tag + "_re_sum"] = np.float64(S.real.astype(np.float64).sum())
        out[tag + "_im_sum"] = np.float64(S.imag.astype(np.float64).sum())
        out[tag + "_fortran"] = np.bool_(S.flags["F_CONTIGUOUS"])

    S = sp.stft(wav)
    assert S.shape == (257, 750)
    keep("stft_default", S)
    keep("stft_512_160", sp.stft(wav, n_fft=512, hop_length=160))
    keep("stft_400_160_reflect", sp.stft(wav, n_fft=400, hop_length=160, pad_mode="reflect"))
    keep("stft_512_win400_hop160", sp.stft(wav, n_fft=512, win_length=400, hop_length=160))
    keep("stft_nocenter_512_160", sp.stft(wav, n_fft=512, hop_length=160, center=False))
    keep("stft_hamming_256", sp.stft(wav, n_fft=256, window="hamming"))
    keep("stft_1024_256", sp.stft(wav, n_fft=1024, hop_length=256))
    keep("stft_f32in_512_160", sp.stft(wav.astype(np.float32), n_fft=512, hop_length=160))
    ri = sp.stft(wav[:4000], n_fft=512, hop_length=160, return_complex=False)
    out["stft_ri_4000"] = ri.astype(np.float32)

    # ---- seeded synthetic batch (cfg 2 shape, smaller batch) ---------------
    rng = np.random.RandomState(1234)
    x = (0.1 * rng.randn(4, 160000)).astype(np.float32)
    out["synth_seed"] = np.int64(1234)
    Sb = sp.stft(x, n_fft=512, hop_length=160)
    assert Sb.shape == (4, 257, 1001)
    keep("stft_synth4", Sb)
    # short ragged-ish lengths around the n_fft / hop edges
    for n in (512, 513, 671, 672, 673, 1000, 1601):
        xs = (0.1 * np.random.RandomState(n).randn(2, n)).astype(np.float32)
        out["stft_short_%d" % n] = np.ascontiguousarray(sp.stft(xs, n_fft=512, hop_length=160))

    # ---- frame() ----------------------------------------------------------
    fr = sp.frame(np.arange(40, dtype=np.float64).reshape(2, 20), frame_length=8, hop_length=3)
    out["frame_2x20_8_3"] = fr

    # ---- amplitude_to_dB --------------------------------------------------
    rng = np.random.RandomState(7)
    a2 = rng.rand(41, 30) ** 6
    a3 = rng.rand(3, 41, 30) ** 6
    a3[1] *= 1e-7  # quiet utterance batched with loud ones: batch-global top_db floor
    a4 = rng.rand(2, 3, 17, 11) ** 8
    a4[1, 2] *= 1e-9
    a3[0, 0, 0] = 0.0  # hits amin
    for tag, a in (("db2", a2), ("db3", a3), ("db4", a4)):
        out[tag + "_in"] = a
        out[tag + "_power"] = sp.amplitude_to_dB(a)
        out[tag + "_mag_ref2_top60"] = sp.amplitude_to_dB(a, stype="magnitude", ref=2.0, top_db=60.0)
        out[tag + "_notop"] = sp.amplitude_to_dB(a, top_db=None)
    out["db3_f32_power"] = sp.amplitude_to_dB(a3.astype(np.float32))

    # ---- Kaldi-style fbank of examples/conformer --------------------------
    banks, centers = ds.get_mel_banks(80, 512, 16000, 20, 8000)
    assert banks.shape == (80, 257)
    nz = np.nonzero(banks)
    out["kaldi_mel_shape"] = np.array(banks.shape, dtype=np.int64)
    out["kaldi_mel_rows"] = nz[0].astype(np.int32)
    out["kaldi_mel_cols"] = nz[1].astype(np.int32)
    out["kaldi_mel_vals"] = banks[nz]
    out["kaldi_mel_centers"] = centers.reshape(-1)
    banks23, _ = ds.get_mel_banks(23, 512, 16000, 20, 8000)
    out["kaldi_mel23_dense"] = banks23

    f = ds.compute_fbank_feats(wav * (1 << 15), 16000, 25, 10, 80)
    assert f.shape == (598, 80)
    out["kaldi_wav_shape"] = np.array(f.shape, dtype=np.int64)
    rows = np.array(sorted(set([0, 1, 2, 595, 596, 597] + list(range(5, 598, 23)))), dtype=np.int64)
    out["kaldi_wav_rows"] = rows
    out["kaldi_wav_vals"] = f[rows]
    out["kaldi_wav_sum"] = np.float64(f.sum())
    out["kaldi_wav_min"] = np.float64(f.min())
    out["kaldi_wav_max"] = np.float64(f.max())
    for i, n in enumerate((16000, 12345, 400, 559, 560)):
        w = np.round(np.random.RandomState(100 + i).randn(n) * 3000.0)
        out["kaldi_synth_in_%d" % n] = w
        out["kaldi_synth_out_%d" % n] = ds.compute_fbank_feats(w, 16000, 25, 10, 80)
    out["preemph_10"] = ds.preemphasis(np.arange(10, dtype=np.float64) ** 2)

    # ---- collate helpers --------------------------------------------------
    cm, mk = ref["common"], ref["mask"]
    seqs = [np.arange(12, dtype=np.float32).reshape(4, 3), np.ones((2, 3), np.float32), 2 * np.ones((3, 3), np.float32)]
    out["pad_sequence_f32"] = cm.pad_sequence(seqs, batch_first=True, padding_value=0.0, padding_max_len=6, atype=np.float32)
    ys = [np.array([1, 2, 3, 4, 5], np.int32), np.array([4, 5, 6], np.int32), np.array([7, 8, 9, 10], np.int32)]
    out["pad_sequence_i32"] = cm.pad_sequence(ys, batch_first=True, padding_value=-1, padding_max_len=7, atype=np.int32)
    ys_in, ys_out = cm.add_sos_eos(ys, 10, 11)
    out["add_sos_eos_in"] = np.concatenate(ys_in)
    out["add_sos_eos_out"] = np.concatenate(ys_out)
    out["make_pad_mask_5_3_2"] = np.asarray(mk.make_pad_mask([5, 3, 2]))
    out["make_pad_mask_max8"] = np.asarray(mk.make_pad_mask([5, 3, 2], max_len=8))
    out["subsequent_mask_5"] = np.asarray(mk.subsequent_mask(5))

    path = os.path.join(HERE, "reference_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")
    collate_goldens(ref)
    decode_goldens(ref)
    post_goldens(ref)
    istft_goldens(ref)


def decode_goldens(ref):
    """remove_duplicates_and_blank (mindaudio/utils/common.py:116-125) and wer (mindaudio/metric/wer.py:4-57),
    both pure Python in the reference."""
    cm = ref["common"]
    wer_mod = _load("ref_wer", "mindaudio/metric/wer.py")
    rng = np.random.RandomState(77)
    out = {}
    frames = rng.randint(0, 6, (12, 40)).astype(np.int32)  # small alphabet: many repeats and blanks (0)
    frames[3] = 0
    frames[4] = 2
    out["greedy_frames"] = frames
    hyps = [cm.remove_duplicates_and_blank(row.tolist()) for row in frames]
    out["greedy_hyp_len"] = np.array([len(h) for h in hyps], np.int32)
    out["greedy_hyp_flat"] = np.array([t for h in hyps for t in h], np.int32)
    refs, hys, vals = [], [], []
    for _ in range(40):
        r = rng.randint(1, 8, rng.randint(1, 12)).tolist()
        h = rng.randint(1, 8, rng.randint(0, 12)).tolist()
        refs.append(r)
        hys.append(h)
        vals.append(wer_mod.wer(r, h))
    out["wer_ref_len"] = np.array([len(r) for r in refs], np.int32)
    out["wer_ref_flat"] = np.array([t for r in refs for t in r], np.int32)
    out["wer_hyp_len"] = np.array([len(h) for h in hys], np.int32)
    out["wer_hyp_flat"] = np.array([t for h in hys for t in h], np.int32)
    out["wer_values"] = np.array(vals, np.float64)
    path = os.path.join(HERE, "decode_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


def post_goldens(ref):
    """SURVEY 8f-4, the pieces that are pure NumPy / Python in the reference: spectrum.magphase (spectrum.py:701-735),
    load_cmvn (mindaudio/utils/load_files.py:9-36) and the per-utterance CMVN sums of compute_cmvn_stats.py:45-60."""
    import json
    import tempfile

    sp, io, ds = ref["spectrum"], ref["io"], ref["dataset"]
    out = {}
    wav, _ = io.read(os.path.join(REF, "tests/samples/ASR/BAC009S0002W0122.wav"))
    D = sp.stft(wav[:8000], n_fft=512, hop_length=160)
    D[3, 5] = 0  # exercises the zero-magnitude branch
    out["magphase_in"] = D
    for pw in (1.0, 2.0, 0.5):
        mag, ph = sp.magphase(D.copy(), power=pw, iscomplex=True)
        out["magphase_mag_%g" % pw] = mag
        out["magphase_phase_%g" % pw] = ph
    feats = ds.compute_fbank_feats(wav * (1 << 15), 16000, mel_bin=80, frame_len=25, frame_shift=10)
    out["cmvn_mean_stat"] = np.sum(feats, axis=0)
    out["cmvn_var_stat"] = np.sum(np.square(feats), axis=0)
    out["cmvn_frames"] = np.int64(feats.shape[0])
    lf = _load("ref_load_files", "mindaudio/utils/load_files.py")
    tmp = tempfile.mkdtemp(prefix="ma_cmvn_")
    path = os.path.join(tmp, "global_cmvn")
    with open(path, "w") as f:
        f.write(json.dumps({"mean_stat": out["cmvn_mean_stat"].tolist(), "var_stat": out["cmvn_var_stat"].tolist(),
                            "frame_num": int(out["cmvn_frames"])}))
    mean, istd = lf.load_cmvn(path, True)
    out["cmvn_mean"], out["cmvn_istd"] = np.asarray(mean), np.asarray(istd)
    shutil.rmtree(tmp)
    path = os.path.join(HERE, "post_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


def istft_goldens(ref):
    """spectrum.istft (spectrum.py:346-474).  The reference uses np.float_ (removed in NumPy 2; its
    requirements.txt pins numpy<2), so the alias is restored for the duration of the call — the arithmetic is untouched."""
    sp, io = ref["spectrum"], ref["io"]
    had = hasattr(np, "float_")
    if not had:
        np.float_ = np.float64
    try:
        wav, _ = io.read(os.path.join(REF, "tests/samples/ASR/BAC009S0002W0122.wav"))
        x = wav[2000:8000]
        out = {"wave": x}
        cases = {
            "default": (dict(), dict()),                                        # n_fft 512, hop 128 both ways
            "h160": (dict(n_fft=512, hop_length=160), dict(hop_length=160)),
            "nocenter": (dict(n_fft=256, hop_length=64, center=False), dict(hop_length=64, center=False)),
            "len5000": (dict(n_fft=512, hop_length=160), dict(hop_length=160, length=5000)),
            "len7000": (dict(n_fft=512, hop_length=160), dict(hop_length=160, length=7000)),
            "win400": (dict(n_fft=512, win_length=400, hop_length=100, window="hamming"),
                       dict(win_length=400, hop_length=100, window="hamming")),
            "n400": (dict(n_fft=400, hop_length=100), dict(hop_length=100)),
        }
        for tag, (kw_f, kw_i) in cases.items():
            D = sp.stft(x, **kw_f)
            out["spec_" + tag] = np.ascontiguousarray(D)
            out["y_" + tag] = sp.istft(D, **kw_i)
        rng = np.random.RandomState(7)
        xb = (0.1 * rng.randn(3, 3000)).astype(np.float32)
        D = sp.stft(xb, n_fft=512, hop_length=160)
        out["spec_batch"] = np.ascontiguousarray(D)
        out["y_batch"] = sp.istft(D, hop_length=160)
    finally:
        if not had:
            del np.float_
    path = os.path.join(HERE, "istft_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.0f KiB" % (os.path.getsize(path) / 1024), len(out), "arrays")


def _write_wav(path, pcm16):
    import wave

    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(pcm16.astype("<i2").tobytes())


def collate_goldens(ref):
    """Row a7: BucketASRDataset (dataset.py:290-381), DistributedSampler (distributed.py:4-29) and
    CollateFunc.__call__ (dataset.py:536-656) of the reference, run on small synthetic wav files written to a
    temporary directory.  `np.bool` (removed in NumPy 1.24; the reference pins numpy<2) is aliased for the run."""
    import random
    import tempfile

    if not hasattr(np, "bool"):
        np.bool = bool  # mask.py:190,233 use the removed alias
    ds, dist = ref["dataset"], ref["dist"]
    out = {}
    tmp = tempfile.mkdtemp(prefix="ma_collate_")
    rng = np.random.RandomState(4242)
    chars = list("abcdefghij")
    with open(os.path.join(tmp, "dict.txt"), "w") as f:
        f.write("<blank> 0\n<unk> 1\n")
        for i, c in enumerate(chars):
            f.write("%s %d\n" % (c, i + 2))
    n_utt = 23
    lens = rng.randint(4000, 30000, n_utt)
    lens[3] = lens[7]  # equal durations: exercises the sort ties
    rows = [["id", "duration", "wav", "transcript"]]
    pcm_all, trans = [], []
    for i in range(n_utt):
        pcm = np.round(rng.randn(lens[i]) * 2500.0).clip(-32768, 32767).astype(np.int16)
        path = os.path.join(tmp, "utt%02d.wav" % i)
        _write_wav(path, pcm)
        ntok = int(rng.randint(1, 9))
        tr = "".join(rng.choice(chars + ["z"], ntok))  # 'z' is out of vocabulary -> id 1
        rows.append([str(i), "%.4f" % (lens[i] / 16000.0), path, tr])
        pcm_all.append(pcm)
        trans.append(tr)
    import csv

    data_file = os.path.join(tmp, "train.csv")
    with open(data_file, "w", newline="") as f:
        csv.writer(f).writerows(rows)
    out["utt_lens"] = lens.astype(np.int64)
    out["utt_pcm"] = np.concatenate(pcm_all)
    out["utt_transcripts"] = np.array(trans)
    out["dict_chars"] = np.array(chars)

    kw = dict(max_length=180, min_length=30, token_max_length=7, token_min_length=1, frame_bucket_limit="60,120,200",
              batch_bucket_limit="20,15,10", batch_factor=0.2)
    for gs in (1, 2):
        dset = ds.BucketASRDataset(data_file, os.path.join(tmp, "dict.txt"), frame_factor=100, group_size=gs, **kw)
        out["bucket_g%d_nbatches" % gs] = np.int64(len(dset))
        flat, sizes, limits = [], [], []
        for b in range(len(dset)):
            data, sos, eos, max_src, max_tgt = dset[b]
            sizes.append(len(data))
            limits.append(max_src)
            flat += [int(d[0][3:5]) for d in data]  # uttid "uttNN.wav"
        out["bucket_g%d_sizes" % gs] = np.array(sizes, np.int64)
        out["bucket_g%d_limits" % gs] = np.array(limits, np.int64)
        out["bucket_g%d_utts" % gs] = np.array(flat, np.int64)
        out["bucket_g%d_sos_eos_tgt" % gs] = np.array([sos, eos, max_tgt], np.int64)
        out["bucket_g%d_labels0" % gs] = np.array([d[2] for d in dset[0][0]])

    for tag, (rank, gs, group) in {"r0g1": (0, 1, False), "r1g3": (1, 3, True)}.items():
        smp = dist.DistributedSampler(list(range(17)), rank, gs, shuffle=True, group=group)
        out["sampler_%s_epoch1" % tag] = np.array(list(iter(smp)), np.int64)
        out["sampler_%s_epoch2" % tag] = np.array(list(iter(smp)), np.int64)
    smp = dist.DistributedSampler(list(range(7)), 1, 2, shuffle=False, group=True)
    out["sampler_noshuffle"] = np.array(list(iter(smp)), np.int64)

    names = ["xs_pad", "ys_pad", "ys_in_pad", "ys_out_pad", "r_ys_in_pad", "r_ys_out_pad", "xs_masks", "ys_sub_masks",
             "ys_masks", "ys_lengths", "xs_chunk_masks"]
    dset = ds.BucketASRDataset(data_file, os.path.join(tmp, "dict.txt"), frame_factor=100, group_size=2, **kw)
    fe = {"mel_bins": 80, "frame_length": 25, "frame_shift": 10}
    cases = {
        "plain_r0g1": dict(rank=0, group_size=1),
        "plain_r1g2": dict(rank=1, group_size=2),
        "specaug_r0g2": dict(rank=0, group_size=2, use_spec_aug=True,
                             spec_aug_conf={"num_t_mask": 2, "num_f_mask": 2, "max_t": 50, "max_f": 10}),
        "static_chunk_r0g1": dict(rank=0, group_size=1, static_chunk_size=4, num_decoding_left_chunks=1),
    }
    for tag, conf in cases.items():
        cf = ds.CollateFunc(feature_extraction_conf=fe, **conf)
        for bi in (0, len(dset) - 1):
            data, sos, eos, max_src, max_tgt = dset[bi]
            random.seed(99 + bi)
            res = cf(data, sos, eos, max_src, max_tgt)
            for nme, arr in zip(names, res):
                out["collate_%s_b%d_%s" % (tag, bi, nme)] = np.asarray(arr)
        cf.pool.close()
        cf.pool.join()
    path = os.path.join(HERE, "collate_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
