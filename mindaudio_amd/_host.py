"""Host-side plumbing shared by the Python mirrors: tensor marshalling, cached device tables
(windows, mel banks), workspaces.  PyTorch is used for device memory and streams only."""
import ctypes
import math
import threading

import numpy as np
from scipy.signal import get_window

from . import _lib

_torch = None


def torch():
    global _torch
    if _torch is None:
        import torch as _t

        _torch = _t
    rec = _lib.recording()
    if rec is not None:  # the block table this thread is filling keeps every tensor the wrappers allocate (train/block_table.py)
        return rec.torch(_torch)
    return _torch


_gpu_ready = False


def require_gpu():
    """torch, after checking that a HIP device exists; the first call also runs the library's one-time kernel set-up (ma_init)."""
    global _gpu_ready
    t = torch()
    if not _gpu_ready:
        if not t.cuda.is_available():
            raise _lib.MindaudioAmdError("mindaudio_amd needs a HIP device (torch.cuda.is_available() is False); "
                                         "there is no CPU fallback")
        _lib.check(_lib.load().ma_init(), "ma_init")
        _gpu_ready = True
    return t


# The pin is per THREAD: a loader thread that runs device fbank (or ECAPA inference) while the training thread is inside
# forward_backward() must keep launching on ITS torch stream, not on the training thread's pinned one.
_tls = threading.local()


def current_stream_ptr():
    p = getattr(_tls, "pin", None)
    if p is not None:
        return p
    return ctypes.c_void_p(torch().cuda.current_stream().cuda_stream)


def swap_pinned(ptr):
    """Set this thread's pinned stream pointer (None: follow torch's current stream); returns the previous one."""
    prev = getattr(_tls, "pin", None)
    _tls.pin = ptr
    return prev


class pinned_stream:
    """`with pinned_stream():` - resolve torch's current stream ONCE for the block (torch.cuda.current_stream() costs ~1.5 us of
    host time per call, and a training step makes several hundred launches; the step is host-bound without this).  Thread-local."""

    def __enter__(self):
        self.prev = swap_pinned(ctypes.c_void_p(torch().cuda.current_stream().cuda_stream))
        return self

    def __exit__(self, *exc):
        swap_pinned(self.prev)
        return False


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def to_device_2d(x, dtype=None):
    """(array-like | tensor) with time on the last axis -> (cuda float32 (B, N) tensor, leading shape, was_numpy)."""
    t = require_gpu()
    was_numpy = not isinstance(x, t.Tensor)
    if was_numpy:
        arr = np.asarray(x)
        if arr.dtype not in (np.float32, np.float64, np.int16, np.int32):
            arr = arr.astype(np.float64)
        x = t.from_numpy(np.ascontiguousarray(arr))
    if not x.is_cuda:
        x = x.cuda()
    lead = tuple(x.shape[:-1])
    x = x.reshape(-1, x.shape[-1]).to(t.float32)
    if x.stride(-1) != 1:
        x = x.contiguous()
    return x, lead, was_numpy


# ---- windows ---------------------------------------------------------------------------------
def centred_window_f64(window, win_length, n_fft):
    """scipy get_window(window, win_length, fftbins=True) padded centred to n_fft (spectrum.py:173-175)."""
    w = get_window(window, win_length, fftbins=True)
    if win_length > n_fft:
        raise ValueError("Target size ({:d}) must be at least input size ({:d})".format(n_fft, win_length))
    left = (n_fft - win_length) // 2
    return np.pad(w, (left, n_fft - win_length - left))


_window_cache = {}


def device_window(window, win_length, n_fft, device):
    key = ("centred", str(window), int(win_length), int(n_fft), str(device))
    if key not in _window_cache:
        w = centred_window_f64(window, win_length, n_fft).astype(np.float32)
        _window_cache[key] = torch().from_numpy(w).to(device)
    return _window_cache[key]


def device_kaldi_window(frame_len, device):
    key = ("kaldi", int(frame_len), str(device))
    if key not in _window_cache:
        w = np.power(np.hanning(frame_len), 0.85).astype(np.float32)  # dataset.py:126
        _window_cache[key] = torch().from_numpy(w).to(device)
    return _window_cache[key]


# ---- mel banks ------------------------------------------------------------------------------
def _hz_to_mel(f, mel_type):
    """MelScale's two frequency warps (torchaudio `_hz_to_mel`, which MindSpore's audio ops mirror): HTK 2595 log10(1 + f/700);
    Slaney = linear below 1 kHz (200/3 Hz per mel), logarithmic above (27 steps per factor 6.4)."""
    f = np.asarray(f, dtype=np.float64)
    if mel_type == "htk":
        return 2595.0 * np.log10(1.0 + f / 700.0)
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m, mel_type):
    m = np.asarray(m, dtype=np.float64)
    if mel_type == "htk":
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_fbanks_f64(n_freqs, f_min, f_max, n_mels, sample_rate, norm="none", mel_type="htk"):
    """MelScale(mel_type, norm) triangles, (n_freqs, n_mels) float64 (spectrum.py:625-626, 686-694).  norm="slaney" divides every
    triangle by half its band width in Hz (area normalisation): fb[:, m] *= 2 / (f_{m+2} - f_m)."""
    all_freqs = np.linspace(0.0, float(sample_rate // 2), n_freqs)
    m_lo, m_hi = float(_hz_to_mel(f_min, mel_type)), float(_hz_to_mel(f_max, mel_type))
    f_pts = _mel_to_hz(np.linspace(m_lo, m_hi, n_mels + 2), mel_type)
    f_diff = np.diff(f_pts)
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    if norm == "slaney":
        fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels]))[None, :]
    return fb


def htk_fbanks_f64(n_freqs, f_min, f_max, n_mels, sample_rate):
    """MelScale(mel_type=HTK, norm=NONE) triangles, (n_freqs, n_mels) float64 (spectrum.py:686-694)."""
    return mel_fbanks_f64(n_freqs, f_min, f_max, n_mels, sample_rate, "none", "htk")


def mel_enum(value, what):
    """'slaney' / 'htk' / 'none' from a string or a MindSpore-style enum (`NormType.SLANEY`, `MelType.HTK`; spectrum.py:668-669)."""
    name = str(getattr(value, "name", value)).lower().rsplit(".", 1)[-1]
    allowed = ("none", "slaney") if what == "norm" else ("htk", "slaney")
    if name not in allowed:
        raise ValueError("%s must be one of %s, got %r" % (what, allowed, value))
    return name


def kaldi_banks_f64(num_bins, n_fft_padded, sample_freq, low_freq, high_freq):
    """(num_bins, n_fft_padded//2+1) Kaldi-style triangles in the mel domain (dataset.py:68-113)."""
    nb = n_fft_padded // 2
    width = sample_freq / n_fft_padded
    lo = 1127.0 * math.log(1.0 + low_freq / 700.0)
    hi = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (hi - lo) / (num_bins + 1)
    idx = np.arange(num_bins).reshape(-1, 1)
    left, centre, right = lo + idx * delta, lo + (idx + 1.0) * delta, lo + (idx + 2.0) * delta
    mel = (1127.0 * np.log(1.0 + width * np.arange(nb) / 700.0))[None, :]
    up = (mel - left) / (centre - left)
    down = (right - mel) / (right - centre)
    tri = np.where(up > down, down, up)
    tri = np.where(tri < 0, 0, tri)
    return np.pad(tri, ((0, 0), (0, 1)), "constant")


def group_melbank(dense_mels_by_freqs, row_limit=260):
    """(n_mels, n_freqs) float64 bank -> grouped band form of include/mindaudio_amd.h (struct ma_melbank):
    steps[n_rows], row_off[n_rows], start[n_rows*8], weights[total_steps, 8, 4] (float32)."""
    dense = np.asarray(dense_mels_by_freqs, dtype=np.float64)
    n_mels, n_freqs = dense.shape
    n_rows = (n_mels + 7) // 8
    lo4 = np.zeros(n_rows * 8, np.int64)
    hi4 = np.zeros(n_rows * 8, np.int64)
    for m in range(n_mels):
        nz = np.nonzero(dense[m])[0]
        if nz.size:
            lo4[m] = (int(nz[0]) // 4) * 4
            hi4[m] = -(-(int(nz[-1]) + 1) // 4) * 4
    steps = np.zeros(n_rows, np.int32)
    for i in range(n_rows):
        steps[i] = max(1, int(((hi4[8 * i:8 * i + 8] - lo4[8 * i:8 * i + 8]) // 4).max()))
    row_off = np.concatenate(([0], np.cumsum(steps)[:-1])).astype(np.int32)
    total = int(steps.sum())
    start = np.zeros(n_rows * 8, np.int32)
    weights = np.zeros((total, 8, 4), np.float64)
    for i in range(n_rows):
        span = 4 * int(steps[i])
        for g in range(8):
            m = 8 * i + g
            st = int(min(lo4[m], row_limit - span))  # keep start + span inside the zero-padded row
            st = max(st - st % 4, 0)
            start[m] = st
            if m < n_mels:
                seg = np.zeros(span)
                hi = min(st + span, n_freqs)
                seg[:hi - st] = dense[m, st:hi]
                assert np.count_nonzero(dense[m]) == np.count_nonzero(seg), "band does not fit its row"
                weights[row_off[i]:row_off[i] + steps[i], g, :] = seg.reshape(-1, 4)
    return steps, row_off, start, weights.astype(np.float32)


class DeviceMelBank:
    """Grouped band form of a (n_mels, n_freqs) filterbank on the device + its ma_melbank struct."""

    def __init__(self, dense_mels_by_freqs, device):
        t = torch()
        dense = np.asarray(dense_mels_by_freqs, dtype=np.float64)
        self.n_mels, self.n_freqs = dense.shape
        n_fft = 2 * (self.n_freqs - 1)
        row_limit = int(_lib.load().ma_mel_row_stride(n_fft))
        if row_limit < 0:
            _lib.check(row_limit, "mel bank for n_fft=%d" % n_fft)
        steps, row_off, start, weights = group_melbank(dense, row_limit)
        self.steps = t.from_numpy(steps).to(device)
        self.row_off = t.from_numpy(row_off).to(device)
        self.start = t.from_numpy(start).to(device)
        self.weights = t.from_numpy(np.ascontiguousarray(weights)).to(device)
        self.struct = _lib.MelBank(self.n_mels, self.n_freqs, len(steps), int(steps.sum()), self.steps.data_ptr(),
                                   self.row_off.data_ptr(), self.start.data_ptr(), self.weights.data_ptr())

    def ref(self):
        return ctypes.byref(self.struct)


_mel_cache = {}


def device_htk_bank(n_fft, f_min, f_max, n_mels, sample_rate, device, norm="none", mel_type="htk"):
    key = ("mel", n_fft, float(f_min), float(f_max), n_mels, sample_rate, str(device), norm, mel_type)
    if key not in _mel_cache:
        fb = mel_fbanks_f64(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate, norm, mel_type)
        _mel_cache[key] = DeviceMelBank(fb.T, device)
    return _mel_cache[key]


def device_kaldi_bank(mel_bin, n_fft, sample_freq, low, high, device):
    key = ("kaldi", mel_bin, n_fft, float(sample_freq), float(low), float(high), str(device))
    if key not in _mel_cache:
        _mel_cache[key] = DeviceMelBank(kaldi_banks_f64(mel_bin, n_fft, sample_freq, low, high), device)
    return _mel_cache[key]


# ---- workspaces ------------------------------------------------------------------------------
_ws = {}


def workspace(nbytes, device):
    key = str(device)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch().empty(max(int(nbytes), 1 << 16), dtype=torch().uint8, device=device)
        _ws[key] = buf
    rec = _lib.recording()
    if rec is not None:  # a block table being filled names this buffer: it must outlive a later, larger workspace
        rec.keep.append(buf)
    return buf
