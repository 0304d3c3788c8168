"""Mirror of the feature half of examples/conformer/dataset.py (compute_fbank_feats, :159-168),
batched on the device instead of the reference's multiprocessing.Pool(8) (:449, :479)."""
import numpy as np

from .. import _host, _lib


def compute_fbank_feats_batch(wavs, lengths, sample_rate=16000, frame_len=25, frame_shift=10, mel_bin=80):
    """wavs: (B, max_n) padded waves scaled by 2^15 (dataset.py:390); lengths: (B,) valid samples.
    Returns a device tensor (B, max_frames, mel_bin) float32 with zero rows past each utterance's end
    (what pad_sequence produces, dataset.py:563-569) and the per-utterance frame counts."""
    t = _host.require_gpu()
    lib = _lib.load()
    x, _, _ = _host.to_device_2d(wavs)
    flen = sample_rate * frame_len // 1000
    fshift = sample_rate * frame_shift // 1000
    lens = t.as_tensor(np.asarray(lengths) if not isinstance(lengths, t.Tensor) else lengths).to(
        device=x.device, dtype=t.int64)
    if x.shape[-1] < flen:
        raise ValueError("signal shorter than one frame ({} < {})".format(x.shape[-1], flen))
    max_frames = (x.shape[-1] - flen) // fshift + 1
    win = _host.device_kaldi_window(flen, x.device)
    # dataset.py:152,167: bank for fs*2 = sample_rate, 20..8000 Hz, 512-point FFT
    bank = _host.device_kaldi_bank(mel_bin, 512, float(sample_rate), 20.0, 8000.0, x.device)
    out = t.empty((x.shape[0], max_frames, mel_bin), dtype=t.float32, device=x.device)
    ws_bytes = lib.ma_fbank_workspace_bytes(x.shape[0], max_frames)
    ws = _host.workspace(ws_bytes, x.device)
    frames = t.empty((x.shape[0],), dtype=t.int64, device=x.device)
    rc = lib.ma_fbank_kaldi_f32(_host.ptr(x), _host.ptr(lens), x.shape[0], x.shape[-1], x.stride(0), flen, fshift,
                                512, _host.ptr(win), bank.ref(), 0.97, _host.ptr(out), _host.ptr(frames),
                                _host.ptr(ws), ws.numel(), _host.current_stream_ptr())
    _lib.check(rc, "compute_fbank_feats")
    return out, frames


def compute_fbank_feats(wav, sample_rate, frame_len, frame_shift, mel_bin):
    """Single-utterance signature of dataset.py:159: (N,) -> (num_frames, mel_bin)."""
    wav = np.asarray(wav)
    out, frames = compute_fbank_feats_batch(wav[None, :], [wav.shape[0]], sample_rate, frame_len, frame_shift,
                                            mel_bin)
    return out[0, :int(frames[0])].cpu().numpy().astype(np.float64)


# ---- batch assembly (dataset.py:170-726): host bookkeeping + device collate --------------------------------
import csv  # noqa: E402
import math  # noqa: E402
import random  # noqa: E402

from ..data import io as _io  # noqa: E402
from ..utils.distributed import DistributedSampler  # noqa: E402

IGNORE_ID = -1  # mindaudio/utils/common.py:7

COLUMNS = ("xs_pad", "ys_pad", "ys_in_pad", "ys_out_pad", "r_ys_in_pad", "r_ys_out_pad", "xs_masks", "ys_sub_masks",
           "ys_masks", "ys_lengths", "xs_chunk_masks")  # return order of CollateFunc.__call__ (dataset.py:644-656)


def load_samples(data_file, dict_file, frame_factor=100):
    """Index file -> [(uttid, wav_path, duration_in_frames, "id id ... ", output_dim)] (dataset.py:178-209).
    csv columns: _, seconds, wav path, transcript; characters outside the dictionary map to id 1."""
    with open(dict_file, "r") as fh:
        vocab = [ln.split()[0] for ln in fh]
    first_pos = {}
    for i, sym in enumerate(vocab):
        first_pos.setdefault(sym, i)
    out_dim = len(vocab) + 1
    samples = []
    with open(data_file, "r") as fh:
        rows = csv.reader(fh)
        next(rows, None)  # header line
        for row in rows:
            ids = "".join("{} ".format(first_pos.get(ch, 1)) for ch in row[3].replace(" ", ""))
            samples.append((row[2].split("/")[-1], row[2], int(float(row[1]) * frame_factor), ids, out_dim))
    return samples


class BucketASRDataset:
    """Length-bucketed batches of (uttid, wav_path, label ids) — constructor and item layout of dataset.py:290-381
    (`data_file`/`dict_file` may be replaced by a pre-parsed `samples=` list)."""

    def __init__(self, data_file=None, dict_file=None, max_length=10240, min_length=0, token_max_length=200,
                 token_min_length=1, frame_bucket_limit="200,300", batch_bucket_limit="220,200", batch_factor=0.2,
                 frame_factor=100, group_size=1, samples=None):
        self.group_size = group_size
        self.frame_bucket_limit = [int(v) for v in frame_bucket_limit.split(",")]
        self.batch_bucket_limit = [int(int(v) * batch_factor * group_size) for v in batch_bucket_limit.split(",")]
        if len(self.frame_bucket_limit) != len(self.batch_bucket_limit):
            raise AssertionError("frame_bucket_limit and batch_bucket_limit differ in length")
        if samples is None:
            samples = load_samples(data_file, dict_file, frame_factor)
        self.data = sorted(samples, key=lambda s: s[2])  # stable, by duration (dataset.py:270)
        self.token_max_length = token_max_length
        self.output_dim = self.data[-1][4] if self.data else 0
        open_buckets = [[] for _ in self.frame_bucket_limit]
        self.batches = []
        kept = 0
        for uttid, path, frames, ids, _ in self.data:
            n_tok = len(ids.split())
            if not (min_length <= frames <= max_length and token_min_length <= n_tok <= token_max_length):
                continue
            kept += 1
            b = self._bucket(frames)
            open_buckets[b].append((uttid, path, ids))
            if len(open_buckets[b]) >= self.batch_bucket_limit[b]:
                self.batches.append((open_buckets[b], self.frame_bucket_limit[b]))
                open_buckets[b] = []
        for b, rest in enumerate(open_buckets):  # leftovers are repeated to a full batch (dataset.py:360-368)
            if rest:
                want = self.batch_bucket_limit[b]
                self.batches.append(((rest * math.ceil(want / len(rest)))[:want], self.frame_bucket_limit[b]))
        self.num_kept, self.num_dropped = kept, len(self.data) - kept
        self.sos = self.eos = self.output_dim - 1

    def _bucket(self, frames):
        for i, limit in enumerate(self.frame_bucket_limit):
            if frames <= limit:
                return i
        raise KeyError(frames)  # longer than the last bucket: the reference's dict lookup fails the same way

    def __len__(self):
        return len(self.batches)

    def __getitem__(self, index):
        data, max_src_len = self.batches[index]
        return data, self.sos, self.eos, max_src_len, self.token_max_length


SPEEDS = (0.9, 1.0, 1.1)  # dataset.py:400


def speed_perturb_batch(waves, sample_rate, device, speeds=None):
    """speed_perturb (dataset.py:398-406) over a list of host waves: draws random.choice(SPEEDS) per utterance (or takes
    `speeds`), resamples the perturbed ones in ONE batched device call and returns the list of host float64 waves."""
    from ..data import processing

    t = _host.require_gpu()
    if speeds is None:
        speeds = [random.choice(SPEEDS) for _ in waves]
    todo = [i for i, s in enumerate(speeds) if s != 1.0]
    if not todo:
        return list(waves)
    n_in = [waves[i].shape[0] for i in todo]
    n_out = [processing.resampled_length(waves[i].shape[0], sample_rate * speeds[i], sample_rate) for i in todo]
    host = np.zeros((len(todo), max(n_in)), np.float32)
    for row, i in enumerate(todo):
        host[row, :n_in[row]] = waves[i]
    y = processing.resample_batch(t.from_numpy(host).to(device), n_in, n_out).cpu().numpy()
    out = list(waves)
    for row, i in enumerate(todo):
        out[i] = y[row, :n_out[row]].astype(np.float64)
    return out


def speed_perturb(waveform, sample_rate, speed=None):
    """speed_perturb of dataset.py:398-406 for one utterance."""
    t = _host.require_gpu()
    return speed_perturb_batch([np.asarray(waveform)], sample_rate, t.device("cuda", t.cuda.current_device()),
                               None if speed is None else [speed])[0]


class _Uploader:
    """The loader's host-to-device copies: from pinned staging buffers (two per kind, in rotation) on a stream of their own, so that
    they do not queue behind the training step whose launches are already on the compute stream when the next batch is collated
    (conformer/train.py enqueues step n, collates batch n + 1, then reads step n's results).  A blocking copy on the compute stream -
    and every torch.tensor(list, device=...) is one - made the host wait for the whole step first (tools/loader_bench.py --step:
    13.1 ms per step against 9.5 for the step alone).  Copies only: no kernel ever runs on this stream."""

    def __init__(self, dev):
        t = _host.torch()
        self.dev, self.stream, self.slots, self.turn = dev, t.cuda.Stream(device=dev), {}, {}

    def stage(self, kind, nbytes):
        """A pinned uint8 buffer of >= nbytes whose previous upload (two batches ago) has completed."""
        t = _host.torch()
        ring = self.slots.setdefault(kind, [None, None])
        i = self.turn[kind] = 1 - self.turn.get(kind, 1)
        slot = ring[i]
        if slot is None or slot[0].numel() < nbytes:
            slot = ring[i] = [t.empty(int(nbytes * 1.25) + 64, dtype=t.uint8, pin_memory=True), None]
        if slot[1] is not None:
            slot[1].synchronize()
        self._cur = slot
        return slot[0]

    def upload(self, pinned, nbytes):
        """-> device uint8 tensor holding pinned[:nbytes]; asynchronous (join() before a kernel reads it)."""
        t = _host.torch()
        main = t.cuda.current_stream(self.dev)
        with t.cuda.stream(self.stream):
            out = pinned[:nbytes].to(self.dev, non_blocking=True)
            ev = t.cuda.Event()
            ev.record(self.stream)
        out.record_stream(main)
        self._cur[1] = ev
        return out

    def join(self):
        t = _host.torch()
        t.cuda.current_stream(self.dev).wait_stream(self.stream)


class CollateFunc:
    """CollateFunc of dataset.py:409-656 with the feature extraction, padding, SpecAugment masking and every
    label/mask column computed on the device.  `__call__` returns the reference's 11 columns (COLUMNS) as device
    tensors: float32 / int32 as in the reference, xs_chunk_masks as torch.bool.

    Host-side work that stays: reading the wav files, the length sort (NumPy's argsort, so ties fall as in the
    reference) and the `random` / `np.random` draws of SpecAugment and dynamic chunks, issued in the reference's call
    order so that a seeded run selects the same masks."""

    def __init__(self, rank, group_size, feature_extraction_conf=None, feature_dither=0.0, use_speed_perturb=False,
                 use_spec_aug=False, spec_aug_conf=None, use_dynamic_chunk=False, use_dynamic_left_chunk=False,
                 decoding_chunk_size=0, static_chunk_size=0, num_decoding_left_chunks=-1, reader=_io.read):
        self.rank, self.group_size = rank, group_size
        self.feature_extraction_conf = feature_extraction_conf
        self.feature_dither = feature_dither
        self.use_speed_perturb = use_speed_perturb
        self.use_spec_aug, self.spec_aug_conf = use_spec_aug, spec_aug_conf
        self.use_dynamic_chunk, self.use_dynamic_left_chunk = use_dynamic_chunk, use_dynamic_left_chunk
        self.decoding_chunk_size, self.static_chunk_size = decoding_chunk_size, static_chunk_size
        self.num_decoding_left_chunks = num_decoding_left_chunks
        self.reader = reader

    # -- host draws ------------------------------------------------------------------------------
    def _spec_aug_intervals(self, frames_sorted, n_freq):
        conf = self.spec_aug_conf
        n_t, n_f = int(conf.get("num_t_mask", 0)), int(conf.get("num_f_mask", 0))
        max_t, max_f = conf.get("max_t", 0), conf.get("max_f", 0)
        t_iv = np.zeros((len(frames_sorted), max(n_t, 1), 2), np.int32)
        f_iv = np.zeros((len(frames_sorted), max(n_f, 1), 2), np.int32)
        for b, frames in enumerate(frames_sorted):  # same draw order as dataset.py:516-533
            for k in range(n_t):
                start = random.randint(0, frames - 1)
                end = min(frames, start + random.randint(1, max_t))
                if random.randint(1, 100) > 20:
                    t_iv[b, k] = (start, end)
            for k in range(n_f):
                start = random.randint(0, n_freq - 1)
                end = min(n_freq, start + random.randint(1, max_f))
                if random.randint(1, 100) > 20:
                    f_iv[b, k] = (start, end)
        return t_iv, n_t, f_iv, n_f

    def _chunk_draw(self, max_len):
        """(chunk_size, num_left_chunks) of add_optional_chunk_mask (mask.py:232-268); chunk_size 0 = no chunk mask."""
        if self.use_dynamic_chunk:
            if self.decoding_chunk_size < 0:
                return max_len, -1
            if self.decoding_chunk_size > 0:
                return self.decoding_chunk_size, self.num_decoding_left_chunks
            chunk = np.random.randint(1, max_len, (1,)).tolist()[0]
            left = -1
            if chunk > max_len // 2:
                chunk = max_len
            else:
                chunk = chunk % 25 + 1
                if self.use_dynamic_left_chunk:
                    left = np.random.randint(0, (max_len - 1) // chunk, (1,)).tolist()[0]
            return chunk, left
        if self.static_chunk_size > 0:
            return self.static_chunk_size, self.num_decoding_left_chunks
        return 0, -1

    # -- the batch from int16 files: one upload of samples, one of metadata, everything else on the device ------------------
    def _collate_pcm(self, pcm, labels_of, dev, flen, fshift, sos, eos, max_src_len, max_tgt_len, mel_bin, frame_len, frame_shift):
        """The whole collate for waves that came as the files' 16-bit samples (round 6).  read -> [speed_perturb] -> waveform * 2^15 ->
        rows sorted by frame count, zero-padded (dataset.py:386-406, 484, 563-569): the lengths after speed perturbation follow from
        the draws alone, so the sort order - and every other host decision: SpecAugment intervals, the chunk draw, the label offsets -
        is known before any sample moves.  All random draws first, in the reference's order; then TWO uploads on the loader's own
        stream (samples, metadata); then the launches: perturbed rows resampled on the device (x 2^15 commutes with the resampler
        bit for bit: a power of two), ma_wave_rows_f32 writes the padded float32 matrix, Kaldi fbank, SpecAugment, the label / mask
        columns."""
        from ..data import processing

        t = _host.torch()
        lib = _lib.load()
        n_in = [int(w.shape[0]) for w in pcm]
        speeds = [random.choice(SPEEDS) for _ in pcm] if self.use_speed_perturb else [1.0] * len(pcm)
        n_fin = [n if s == 1.0 else processing.resampled_length(n, 16000 * s, 16000) for n, s in zip(n_in, speeds)]
        frames = [int(math.floor((n - flen) / fshift) + 1) for n in n_fin]
        order = np.argsort(frames)[::-1]  # dataset.py:484
        n_b = len(order)
        frames_sorted = [frames[i] for i in order]
        labels = [labels_of(i) for i in order]
        # ---- the remaining host draws, in the order of the reference's collate (SpecAugment, then the chunk sizes)
        n_t = n_f = 0
        t_iv = f_iv = np.zeros(0, np.int32)
        if self.use_spec_aug:
            t_iv, n_t, f_iv, n_f = self._spec_aug_intervals(frames_sorted, mel_bin)
        chunk, left = self._chunk_draw((max_src_len - 3) // 4)
        # ---- samples: rows in sorted order into a pinned int16 matrix
        up = self.__dict__.get("_up")
        if up is None or up.dev != dev:
            up = self._up = _Uploader(dev)
        ld = (max(n_in) + 7) // 8 * 8
        pin = up.stage("pcm", n_b * ld * 2)
        host = pin.numpy()[:n_b * ld * 2].view(np.int16).reshape(n_b, ld)
        for row, i in enumerate(order):
            host[row, :n_in[i]] = pcm[i]
        pcm_dev = up.upload(pin, n_b * ld * 2).view(t.int16).view(n_b, ld)
        # ---- metadata: one int32 array (the int64 lengths of the fbank launch first: 8-byte aligned)
        todo = [row for row, i in enumerate(order) if speeds[i] != 1.0]
        kind, src = [0] * n_b, list(range(n_b))
        for k, row in enumerate(todo):
            kind[row], src[row] = 1, k
        k_in = [n_in[order[row]] for row in todo]
        k_out = [n_fin[order[row]] for row in todo]
        tok_off = np.zeros(n_b + 1, np.int32)
        tok_off[1:] = np.cumsum([len(y) for y in labels])
        tokens = np.concatenate(labels) if tok_off[-1] else np.zeros(1, np.int32)
        fin_sorted = [n_fin[i] for i in order]
        parts = [("len64", np.asarray(fin_sorted, np.int64).view(np.int32)), ("xs_len", frames_sorted), ("src", src), ("kind", kind),
                 ("fin", fin_sorted), ("todo", todo), ("zero", [0] * len(todo)), ("k_in", k_in), ("k_out", k_out),
                 ("t_iv", np.asarray(t_iv, np.int32).reshape(-1)), ("f_iv", np.asarray(f_iv, np.int32).reshape(-1)), ("tok_off", tok_off),
                 ("tokens", tokens)]
        off, total = {}, 0
        for name, v in parts:
            n = len(v)
            off[name] = (total, n)
            total += (n + 1) // 2 * 2  # (every part starts on an 8-byte boundary)
        pin_m = up.stage("meta", total * 4)
        words = pin_m.numpy()[:total * 4].view(np.int32)
        for name, v in parts:
            o, n = off[name]
            words[o:o + n] = v
        meta = up.upload(pin_m, total * 4).view(t.int32)
        up.join()
        M = lambda name: meta[off[name][0]:off[name][0] + off[name][1]]  # noqa: E731
        # ---- launches
        max_n = max(max(n_fin), (max_src_len - 1) * fshift + flen)
        ld_dst = (max_n + 3) // 4 * 4
        stream = _host.current_stream_ptr()
        y = None
        if todo:
            x_sub = t.empty((len(todo), (max(k_in) + 3) // 4 * 4), dtype=t.float32, device=dev)
            _lib.check(lib.ma_wave_rows_f32(_host.ptr(pcm_dev), ld, None, 0, _host.ptr(M("todo")), _host.ptr(M("zero")), _host.ptr(M("k_in")),
                                            len(todo), _host.ptr(x_sub), x_sub.stride(0), x_sub.shape[1], stream), "wave_rows")
            y = processing.resample_batch(x_sub, k_in, k_out, ni_dev=M("k_in"), no_dev=M("k_out"))
        wave_dev = t.empty((n_b, ld_dst), dtype=t.float32, device=dev)
        _lib.check(lib.ma_wave_rows_f32(_host.ptr(pcm_dev), ld, _host.ptr(y) if y is not None else None, y.stride(0) if y is not None else 0,
                                        _host.ptr(M("src")), _host.ptr(M("kind")), _host.ptr(M("fin")), n_b, _host.ptr(wave_dev), ld_dst,
                                        ld_dst, stream), "wave_rows")
        lengths = meta[off["len64"][0]:off["len64"][0] + 2 * n_b].view(t.int64)
        xs_all, _ = compute_fbank_feats_batch(wave_dev[:, :max_n], lengths, 16000, frame_len, frame_shift, mel_bin)
        xs_pad = xs_all if xs_all.shape[1] == max_src_len else xs_all[:, :max_src_len].contiguous()
        xs_len_dev = M("xs_len")
        if self.use_spec_aug:
            _lib.check(lib.ma_spec_aug_f32(_host.ptr(xs_pad), n_b, max_src_len, mel_bin, _host.ptr(xs_len_dev), _host.ptr(M("t_iv")), n_t,
                                           _host.ptr(M("f_iv")), n_f, stream), "spec_aug")
        return self._label_columns(M("tokens"), M("tok_off"), xs_len_dev, n_b, sos, eos, max_src_len, max_tgt_len, chunk, left, xs_pad, dev,
                                   keep=(meta, pcm_dev))

    def _label_columns(self, tok_dev, off_dev, xs_len_dev, n_b, sos, eos, max_src_len, max_tgt_len, chunk, left, xs_pad, dev, keep=None):
        t = _host.torch()
        lib = _lib.load()
        stream = _host.current_stream_ptr()
        t2 = lib.ma_subsampled_mask_len(max_src_len)
        l1 = max_tgt_len + 1
        i32 = dict(dtype=t.int32, device=dev)
        f32 = dict(dtype=t.float32, device=dev)
        ys_pad, ys_lengths = t.empty((n_b, max_tgt_len), **i32), t.empty((n_b,), **i32)
        ys_in, ys_out, r_in, r_out = (t.empty((n_b, l1), **i32) for _ in range(4))
        xs_masks, ys_masks, ys_sub = t.empty((n_b, 1, t2), **f32), t.empty((n_b, 1, l1), **f32), t.empty((n_b, l1, l1), **f32)
        chunk_masks = t.empty((n_b, t2 if chunk else 1, t2), dtype=t.bool, device=dev)
        rc = lib.ma_collate_asr_i32(_host.ptr(tok_dev), _host.ptr(off_dev), _host.ptr(xs_len_dev), n_b, int(sos),
                                    int(eos), max_tgt_len, max_src_len, chunk, left, _host.ptr(ys_pad),
                                    _host.ptr(ys_in), _host.ptr(ys_out), _host.ptr(r_in), _host.ptr(r_out),
                                    _host.ptr(xs_masks), _host.ptr(ys_sub), _host.ptr(ys_masks),
                                    _host.ptr(ys_lengths), _host.ptr(chunk_masks), stream)
        _lib.check(rc, "collate")
        del keep  # (the uploaded buffers were alive while the launches were issued; the stream orders their reuse)
        return xs_pad, ys_pad, ys_in, ys_out, r_in, r_out, xs_masks, ys_sub, ys_masks, ys_lengths, chunk_masks

    # -- the collate call --------------------------------------------------------------------------
    def read_batch(self, batch):
        """This rank's files of `batch` as int16 sample arrays (the fast path's input), or None when the reader is not the default
        one or a file is not mono 16-bit PCM (the collate then reads them itself, through `reader`).  No random draws, no device."""
        if self.reader is not _io.read:
            return None
        pcm = []
        for utt in batch[self.rank::self.group_size]:
            got = _io.read_pcm16(utt[1])
            if got is None:
                return None
            if got[1] != 16000:
                raise ValueError("the loader expects 16 kHz audio (dataset.py:390-396)")
            pcm.append(got[0])
        return pcm

    def __call__(self, batch, sos=0, eos=0, max_src_len=2000, max_tgt_len=30, pcm="read"):
        if self.feature_dither != 0.0:
            raise NotImplementedError  # as the reference (dataset.py:559-560)
        t = _host.require_gpu()
        lib = _lib.load()
        conf = self.feature_extraction_conf
        mel_bin, frame_len, frame_shift = int(conf["mel_bins"]), int(conf["frame_length"]), int(conf["frame_shift"])
        mine = batch[self.rank::self.group_size]
        flen, fshift = 16000 * frame_len // 1000, 16000 * frame_shift // 1000
        dev = t.device("cuda", t.cuda.current_device())
        if isinstance(pcm, str):  # (not read ahead by the iterator's helper thread)
            pcm = self.read_batch(batch)
        # pcm: the files' 16-bit samples as they lie (round 6): one upload of int16, every later stage on the device
        if pcm is not None:
            return self._collate_pcm(pcm, lambda i: np.fromiter(map(int, mine[i][2].split()), dtype=np.int32), dev, flen, fshift, sos, eos,
                                     max_src_len, max_tgt_len, mel_bin, frame_len, frame_shift)
        if True:  # the host path (a custom reader, or files that are not mono 16-bit PCM)
            waves = []
            for utt in mine:
                wav, rate = self.reader(utt[1])
                if rate != 16000:
                    raise ValueError("the loader expects 16 kHz audio (dataset.py:390-396)")
                waves.append(wav)
            if self.use_speed_perturb:
                # speed_perturb (dataset.py:398-406): one random.choice per utterance (the reference draws inside its worker
                # processes, so no cross-process draw order exists to reproduce), then resample(wave, 16000 * speed, 16000) =
                # scipy.signal.resample, here batched on the device
                waves = speed_perturb_batch(waves, 16000, dev)
            frames = [int(math.floor((w.shape[0] - flen) / fshift) + 1) for w in waves]
            order = np.argsort(frames)[::-1]  # dataset.py:484
            n_fin = [w.shape[0] for w in waves]
            # padded wave matrix: long enough for max_src_len frames so that the kernel's output *is* xs_pad
            max_n = max(max(n_fin), (max_src_len - 1) * fshift + flen)
            host = np.zeros((len(order), (max_n + 3) // 4 * 4), np.float32)
            for row, i in enumerate(order):
                host[row, :waves[i].shape[0]] = waves[i] * 32768.0  # waveform * (1 << 15), dataset.py:390 (exact in f32)
            wave_dev = t.from_numpy(host).to(dev)
        frames_sorted = [frames[i] for i in order]
        labels = [np.fromiter(map(int, mine[i][2].split()), dtype=np.int32) for i in order]
        n_b = len(order)
        lengths = np.array([n_fin[i] for i in order], np.int64)
        xs_all, _ = compute_fbank_feats_batch(wave_dev[:, :max_n], lengths, 16000, frame_len, frame_shift, mel_bin)
        xs_pad = xs_all if xs_all.shape[1] == max_src_len else xs_all[:, :max_src_len].contiguous()
        stream = _host.current_stream_ptr()
        xs_len_dev = t.tensor(frames_sorted, dtype=t.int32, device=dev)
        if self.use_spec_aug:
            t_iv, n_t, f_iv, n_f = self._spec_aug_intervals(frames_sorted, mel_bin)
            t_dev, f_dev = t.from_numpy(t_iv).to(dev), t.from_numpy(f_iv).to(dev)
            _lib.check(lib.ma_spec_aug_f32(_host.ptr(xs_pad), n_b, max_src_len, mel_bin, _host.ptr(xs_len_dev),
                                           _host.ptr(t_dev), n_t, _host.ptr(f_dev), n_f, stream), "spec_aug")
        tok_off = np.zeros(n_b + 1, np.int32)
        tok_off[1:] = np.cumsum([len(y) for y in labels])
        tokens = np.concatenate(labels) if tok_off[-1] else np.zeros(1, np.int32)
        tok_dev, off_dev = t.from_numpy(tokens).to(dev), t.from_numpy(tok_off).to(dev)
        chunk, left = self._chunk_draw((max_src_len - 3) // 4)
        return self._label_columns(tok_dev, off_dev, xs_len_dev, n_b, sos, eos, max_src_len, max_tgt_len, chunk, left, xs_pad, dev)


class _BatchIterable:
    """What the reference builds with GeneratorDataset + map + project (dataset.py:697-743): one collated batch per
    sampler index, re-iterable per epoch."""

    def __init__(self, dataset, sampler, collate):
        self.dataset, self.sampler, self.collate = dataset, sampler, collate

    def __len__(self):
        return len(self.sampler)

    def get_dataset_size(self):
        return len(self)

    def __iter__(self):
        # The files of batch n + 1 are read by a helper thread while batch n is collated and stepped on (round 6): reading is the
        # host's largest share of a batch (2.5 of 4.6 ms at the yaml's buckets), needs no random draw and no device, and releases
        # the GIL.  Everything that draws random numbers or touches the device stays on the calling thread, in the reference's order.
        read = getattr(self.collate, "read_batch", None)
        if read is None:
            for idx in self.sampler:
                yield self.collate(*self.dataset[int(idx)])
            return
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(1) as pool:
            pending = None
            for idx in self.sampler:
                item = self.dataset[int(idx)]
                fut = pool.submit(read, item[0])
                if pending is not None:
                    yield self.collate(*pending[0], pcm=pending[1].result())
                pending = (item, fut)
            if pending is not None:
                yield self.collate(*pending[0], pcm=pending[1].result())

    create_tuple_iterator = __iter__


def create_dataset(data_file, dict_file, collate_conf, dataset_conf, rank=0, group_size=1, number_workers=8):
    """(output_dim, iterable of collated batches) — dataset.py:659-743.  Every rank walks the same shuffled batch
    order (sampler group=False, dataset.py:693) and keeps batch[rank::group_size] inside the collate."""
    collate = CollateFunc(rank=rank, group_size=group_size, **collate_conf)
    dataset = BucketASRDataset(data_file, dict_file, max_length=dataset_conf["max_length"],
                               min_length=dataset_conf["min_length"],
                               token_max_length=dataset_conf["token_max_length"],
                               token_min_length=dataset_conf["token_min_length"],
                               frame_bucket_limit=dataset_conf["frame_bucket_limit"],
                               batch_bucket_limit=dataset_conf["batch_bucket_limit"],
                               batch_factor=dataset_conf["batch_factor"], frame_factor=100, group_size=group_size)
    sampler = DistributedSampler(dataset, rank, group_size, shuffle=True, group=False)
    return dataset.output_dim, _BatchIterable(dataset, sampler, collate)
