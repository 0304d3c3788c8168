"""GPU parity of the device collate (mindaudio_amd.conformer.dataset.CollateFunc) against the outputs of the
reference's own CollateFunc (tests/golden/collate_goldens.npz): integer / mask columns bit-exact, features within the
Kaldi-fbank tolerance of test_features_gpu.py."""
import os
import random
import wave

import numpy as np
import pytest

from test_collate_oracle import BUCKET_KW, NAMES, write_corpus

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def cg():
    return np.load(os.path.join(HERE, "golden", "collate_goldens.npz"))


@pytest.fixture(scope="module")
def corpus(cg, tmp_path_factory):
    """The 23 wav files of the golden run, rebuilt from the stored PCM."""
    tmp = str(tmp_path_factory.mktemp("corpus"))
    off = np.concatenate(([0], np.cumsum(cg["utt_lens"])))
    paths = []
    for i in range(len(cg["utt_lens"])):
        p = os.path.join(tmp, "utt%02d.wav" % i)
        with wave.open(p, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes(cg["utt_pcm"][off[i]:off[i + 1]].astype("<i2").tobytes())
        paths.append(p)
    data_file, dict_file = write_corpus(cg, tmp)
    return tmp, data_file, dict_file


CASES = {
    "plain_r0g1": dict(rank=0, group_size=1),
    "plain_r1g2": dict(rank=1, group_size=2),
    "specaug_r0g2": dict(rank=0, group_size=2, use_spec_aug=True,
                         spec_aug_conf={"num_t_mask": 2, "num_f_mask": 2, "max_t": 50, "max_f": 10}),
    "static_chunk_r0g1": dict(rank=0, group_size=1, static_chunk_size=4, num_decoding_left_chunks=1),
}


@pytest.mark.parametrize("tag", sorted(CASES))
def test_device_collate_vs_reference(cg, corpus, tag):
    import torch

    from mindaudio_amd.conformer.dataset import BucketASRDataset, CollateFunc

    tmp, data_file, dict_file = corpus
    ds = BucketASRDataset(data_file, dict_file, frame_factor=100, group_size=2, **BUCKET_KW)
    # the index file names /corpus/uttNN.wav; read them from the rebuilt directory
    from mindaudio_amd.data.io import read

    cf = CollateFunc(feature_extraction_conf={"mel_bins": 80, "frame_length": 25, "frame_shift": 10},
                     reader=lambda p: read(os.path.join(tmp, os.path.basename(p))), **CASES[tag])
    for bi in (0, len(ds) - 1):
        data, sos, eos, max_src, max_tgt = ds[bi]
        random.seed(99 + bi)
        got = cf(data, sos, eos, max_src, max_tgt)
        assert len(got) == 11
        for name, g in zip(NAMES, got):
            want = cg["collate_%s_b%d_%s" % (tag, bi, name)]
            assert g.is_cuda and tuple(g.shape) == want.shape, name
            g = g.cpu().numpy()
            assert g.dtype == want.dtype, name
            if name == "xs_pad":
                # ln(mel energy): 2e-3 where the energy is well above the float32 noise of its frame (as in
                # test_features_gpu.py); masked / padded cells are exactly zero in both
                assert np.array_equal(g == 0.0, want == 0.0) or np.abs(g - want).max() <= 2e-3
                assert np.abs(g - want).max() <= 2e-2
                assert np.mean(np.abs(g - want)) <= 2e-4
            else:
                assert np.array_equal(g, want), name


def test_create_dataset_iterates_and_shards(cg, corpus):
    import mindaudio_amd.conformer.dataset as D

    tmp, data_file, dict_file = corpus
    from mindaudio_amd.data.io import read

    outs = []
    for rank in (0, 1):
        dim, it = D.create_dataset(
            data_file, dict_file,
            dict(feature_extraction_conf={"mel_bins": 80, "frame_length": 25, "frame_shift": 10},
                 reader=lambda p: read(os.path.join(tmp, os.path.basename(p)))),
            dict(max_length=180, min_length=30, token_max_length=7, token_min_length=1,
                 frame_bucket_limit="60,120,200", batch_bucket_limit="20,15,10", batch_factor=0.2),
            rank=rank, group_size=2)
        assert dim == 13 and len(it) == 4
        outs.append([tuple(c.cpu().numpy() for c in batch) for batch in it])
    # both ranks walk the same batch order and hold complementary halves of each batch
    for b0, b1 in zip(outs[0], outs[1]):
        assert b0[0].shape == b1[0].shape
        assert b0[0].shape[0] + b1[0].shape[0] in (4, 8, 6)


@pytest.mark.parametrize("perturb", [False, True])
def test_int16_fast_path_equals_the_host_path_bit_for_bit(cg, corpus, perturb):
    """Round 6: with the default reader the collate uploads the files' int16 samples once and finishes the waves on the device
    (speed perturbation included: the lengths, hence the sort order, follow from the draws alone; x 2^15 commutes with the resampler).
    Every one of the 11 columns - the features too - must equal, bit for bit, what the host path (any other reader: float64 waves,
    device resample, back to the host, padded float32 matrix up again) gives under the same random draws."""
    import torch

    import mindaudio_amd.conformer.dataset as D
    from mindaudio_amd.data import io as _io

    tmp, data_file, dict_file = corpus
    ds = D.BucketASRDataset(data_file, dict_file, frame_factor=100, group_size=1, **BUCKET_KW)
    kw = dict(feature_extraction_conf={"mel_bins": 80, "frame_length": 25, "frame_shift": 10}, rank=0, group_size=1,
              use_speed_perturb=perturb, use_spec_aug=True, spec_aug_conf={"num_t_mask": 2, "num_f_mask": 2, "max_t": 50, "max_f": 10})
    fast = D.CollateFunc(**kw)                                   # reader is data.io.read: the int16 path
    slow = D.CollateFunc(reader=lambda p: _io.read(p), **kw)     # the same function behind a lambda: the host path
    for bi in range(len(ds)):
        data, sos, eos, max_src, max_tgt = ds[bi]
        data = [(u[0], os.path.join(tmp, os.path.basename(u[1]))) + tuple(u[2:]) for u in data]
        random.seed(5 + bi)
        np.random.seed(5 + bi)
        a = fast(data, sos, eos, max_src, max_tgt)
        random.seed(5 + bi)
        np.random.seed(5 + bi)
        b = slow(data, sos, eos, max_src, max_tgt)
        for name, x, y in zip(NAMES, a, b):
            assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y), (bi, name)
