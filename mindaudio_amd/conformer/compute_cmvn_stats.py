"""Global CMVN statistics of a corpus — examples/conformer/compute_cmvn_stats.py:45-128 with the features and the
per-feature sums computed on the device in batches (the reference walks the files one by one on the host)."""
import csv
import json

import numpy as np

from .. import _host, _lib
from ..data import io as _io
from .dataset import compute_fbank_feats_batch


def accumulate(feats, frames, stats=None):
    """feats (B, T, F) float32 device tensor with frames[b] valid rows -> stats (2, F) float64 device tensor
    (+= sum, sum of squares)."""
    t = _host.require_gpu()
    b, tlen, f = feats.shape
    if stats is None:
        stats = t.zeros((2, f), dtype=t.float64, device=feats.device)
    fr = frames.to(device=feats.device, dtype=t.int32).contiguous()
    _lib.check(_lib.load().ma_cmvn_stats_f64(_host.ptr(feats.contiguous()), _host.ptr(fr), b, tlen, f, _host.ptr(stats),
                                             _host.current_stream_ptr()), "cmvn_stats")
    return stats


def compute_cmvn_stats(wav_paths, mel_bins=80, frame_len=25, frame_shift=10, batch_size=32, reader=_io.read):
    """-> {"mean_stat", "var_stat", "frame_num"} exactly as the reference writes to its json (compute_cmvn_stats.py:120-126)."""
    t = _host.require_gpu()
    stats, total = None, 0
    for lo in range(0, len(wav_paths), batch_size):
        waves = [reader(p)[0] * (1 << 15) for p in wav_paths[lo:lo + batch_size]]
        n = max(w.shape[0] for w in waves)
        host = np.zeros((len(waves), n), np.float32)
        for i, w in enumerate(waves):
            host[i, :w.shape[0]] = w
        lengths = np.array([w.shape[0] for w in waves], np.int64)
        feats, frames = compute_fbank_feats_batch(t.from_numpy(host).cuda(), lengths, 16000, frame_len, frame_shift, mel_bins)
        stats = accumulate(feats, frames, stats)
        total += int(frames.sum())
    s = stats.cpu().numpy()
    return {"mean_stat": s[0].tolist(), "var_stat": s[1].tolist(), "frame_num": total}


def main(in_scp, out_cmvn, mel_bins=80, frame_len=25, frame_shift=10):
    with open(in_scp) as fh:
        rows = list(csv.reader(fh))[1:]
    with open(out_cmvn, "w") as fout:
        fout.write(json.dumps(compute_cmvn_stats([r[2] for r in rows], mel_bins, frame_len, frame_shift)))
