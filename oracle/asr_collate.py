"""TEST INFRASTRUCTURE ONLY — CPU restatement (NumPy) of the batch-assembly row of the hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(mindaudio_amd/) never does.

Restates, function by function:
  examples/conformer/dataset.py:178-209  load_samples      -> load_samples
  examples/conformer/dataset.py:233-381  BucketDatasetBase / BucketASRDataset -> bucket_batches
  mindaudio/utils/distributed.py:4-29    DistributedSampler -> sampler_epoch
  examples/conformer/dataset.py:493-534  CollateFunc.spec_aug -> spec_aug
  examples/conformer/dataset.py:536-656  CollateFunc.__call__ -> collate
  mindaudio/utils/mask.py:154-199        subsequent_chunk_mask -> subsequent_chunk_mask
  mindaudio/utils/mask.py:201-271        add_optional_chunk_mask (static / decoding chunk branches) -> chunk_mask

Pinned: tests/golden/collate_goldens.npz holds outputs of the reference classes themselves, run in the build
container by tests/golden/gen_goldens.py::collate_goldens on 23 synthetic wav files.
"""
import csv
import math
import random

import numpy as np

from . import speech_features as F

IGNORE_ID = -1  # common.py:7


def load_samples(data_file, dict_file, frame_factor=100):
    """[(uttid, wav_path, duration_frames, 'id id id ', output_dim)] — dataset.py:178-209."""
    with open(dict_file) as fh:
        symbols = [line.split()[0] for line in fh]
    table = {}
    for pos, sym in enumerate(symbols):
        table.setdefault(sym, pos)  # list.index semantics: first occurrence
    items = []
    with open(data_file) as fh:
        for n, row in enumerate(csv.reader(fh)):
            if n == 0:
                continue  # header
            ids = "".join("%d " % table.get(ch, 1) for ch in row[3].replace(" ", ""))
            items.append((row[2].split("/")[-1], row[2], int(float(row[1]) * frame_factor), ids, len(symbols) + 1))
    return items


def bucket_batches(items, max_length=10240, min_length=0, token_max_length=200, token_min_length=1,
                   frame_bucket_limit="200,300", batch_bucket_limit="220,200", batch_factor=0.2, group_size=1):
    """Length-bucketed batches: list of ([(uttid, path, ids)], frame_limit) — dataset.py:254-381.

    Items are sorted by duration (stable sort, dataset.py:270); a bucket is flushed when it reaches its batch
    size; leftovers are repeated up to the batch size (dataset.py:360-368)."""
    frame_limits = [int(v) for v in frame_bucket_limit.split(",")]
    batch_limits = [int(int(v) * batch_factor * group_size) for v in batch_bucket_limit.split(",")]
    assert len(frame_limits) == len(batch_limits)

    def bucket_of(length):  # dataset.py:275-283: first bucket whose limit is >= length
        for i, lim in enumerate(frame_limits):
            if length <= lim:
                return i
        raise KeyError(length)

    pending = [[] for _ in frame_limits]
    batches = []
    for utt in sorted(items, key=lambda it: it[2]):
        ntok = len(utt[3].split())
        if utt[2] > max_length or utt[2] < min_length or ntok > token_max_length or ntok < token_min_length:
            continue
        b = bucket_of(utt[2])
        pending[b].append((utt[0], utt[1], utt[3]))
        if len(pending[b]) >= batch_limits[b]:
            batches.append((pending[b], frame_limits[b]))
            pending[b] = []
    for b, rest in enumerate(pending):
        if rest:
            rep = math.ceil(batch_limits[b] / len(rest))
            batches.append(((rest * rep)[:batch_limits[b]], frame_limits[b]))
    return batches


class Sampler:
    """DistributedSampler (distributed.py:4-29): the seed advances by one per epoch *before* use."""

    def __init__(self, n, rank, group_size, shuffle=True, seed=0, group=True):
        self.n, self.rank, self.group_size, self.shuffle, self.seed, self.group = n, rank, group_size, shuffle, seed, group

    def epoch(self):
        if self.shuffle:
            self.seed = (self.seed + 1) & 0xFFFFFFFF
            np.random.seed(self.seed)
            idx = np.random.permutation(self.n)
        else:
            idx = np.arange(self.n)
        return idx[self.rank::self.group_size] if self.group else idx


def spec_aug(xs, num_t_mask=0, num_f_mask=0, max_t=0, max_f=0):
    """In-place SpecAugment with the reference's `random` call order (dataset.py:493-534)."""
    for x in xs:
        frames, freqs = x.shape
        for _ in range(num_t_mask):
            start = random.randint(0, frames - 1)
            end = min(frames, start + random.randint(1, max_t))
            if random.randint(1, 100) > 20:
                x[start:end, :] = 0
        for _ in range(num_f_mask):
            start = random.randint(0, freqs - 1)
            end = min(freqs, start + random.randint(1, max_f))
            if random.randint(1, 100) > 20:
                x[:, start:end] = 0
    return xs


def subsequent_chunk_mask(size, chunk_size, num_left_chunks=-1):
    """mask.py:154-199."""
    i = np.arange(size)[:, None]
    j = np.arange(size)[None, :]
    hi = np.minimum((i // chunk_size + 1) * chunk_size, size)
    lo = np.zeros_like(i) if num_left_chunks < 0 else np.maximum((i // chunk_size - num_left_chunks) * chunk_size, 0)
    return (j >= lo) & (j < hi)


def chunk_mask(xs_len, masks, static_chunk_size=0, num_decoding_left_chunks=-1, decoding_chunk_size=0,
               use_dynamic_chunk=False):
    """add_optional_chunk_mask (mask.py:201-271) without the random training-time chunk draw."""
    masks = masks.astype(bool)
    if use_dynamic_chunk:
        if decoding_chunk_size < 0:
            return masks & subsequent_chunk_mask(xs_len, xs_len, -1)[None]
        if decoding_chunk_size > 0:
            return masks & subsequent_chunk_mask(xs_len, decoding_chunk_size, num_decoding_left_chunks)[None]
        raise NotImplementedError("random dynamic chunk draw")
    if static_chunk_size > 0:
        return masks & subsequent_chunk_mask(xs_len, static_chunk_size, num_decoding_left_chunks)[None]
    return masks


def extract_and_sort(wavs_f64, labels, frame_len=25, frame_shift=10, mel_bin=80):
    """CollateFunc.extract_feature (dataset.py:452-491): features, then sort by frame count, longest first, with
    NumPy's default argsort reversed (ties therefore come out in NumPy's order)."""
    feats = [F.compute_fbank_feats(w * (1 << 15), 16000, frame_len, frame_shift, mel_bin) for w in wavs_f64]
    order = np.argsort([f.shape[0] for f in feats])[::-1]
    return [feats[i] for i in order], [np.fromiter(map(int, labels[i].split()), dtype=np.int32) for i in order], order


def collate(xs, ys, sos, eos, max_src_len, max_tgt_len, **chunk_kw):
    """The 11 columns of CollateFunc.__call__ (dataset.py:563-656) from sorted features `xs` and labels `ys`."""
    xs_pad = F.pad_sequence(xs, True, 0.0, max_src_len, np.float32)
    ys_pad = F.pad_sequence(ys, True, IGNORE_ID, max_tgt_len, np.int32)
    ys_in, ys_out = F.add_sos_eos(ys, sos, eos)
    ys_in_pad = F.pad_sequence(ys_in, True, eos, max_tgt_len + 1, np.int32)
    ys_out_pad = F.pad_sequence(ys_out, True, IGNORE_ID, max_tgt_len + 1, np.int32)
    r_in, r_out = F.add_sos_eos([y[::-1] for y in ys], sos, eos)
    r_ys_in_pad = F.pad_sequence(r_in, True, eos, max_tgt_len + 1, np.int32)
    r_ys_out_pad = F.pad_sequence(r_out, True, IGNORE_ID, max_tgt_len + 1, np.int32)
    xs_lengths = np.array([x.shape[0] for x in xs], np.int32)
    ys_lengths = np.array([len(y) for y in ys], np.int32)
    xs_masks = (~F.make_pad_mask(xs_lengths, max_src_len))[:, None, :].astype(np.float32)
    ys_masks = (~F.make_pad_mask(ys_lengths + 1, max_tgt_len + 1))[:, None, :]
    ys_sub_masks = (ys_masks & F.subsequent_mask(max_tgt_len + 1)[None]).astype(np.float32)
    ys_masks = ys_masks.astype(np.float32)
    xs_masks = xs_masks[:, :, :-2:2][:, :, :-2:2]
    xs_chunk_masks = chunk_mask((xs_pad.shape[1] - 3) // 4, xs_masks, **chunk_kw)
    return (xs_pad, ys_pad, ys_in_pad, ys_out_pad, r_ys_in_pad, r_ys_out_pad, xs_masks, ys_sub_masks, ys_masks,
            ys_lengths, xs_chunk_masks)
