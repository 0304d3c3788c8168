"""CPU ORACLE — test infrastructure, NOT a product path.

NumPy (float64) restatement of the speech-feature half of the hot path of
mindspore-lab/mindaudio.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; mindaudio_amd never does (the product
path raises when the HIP library is missing).

Pinning status
--------------
* stft / frame / amplitude_to_dB / Kaldi fbank (get_mel_banks, preemphasis,
  enframe, compute_fbank_feats) / collate helpers: PINNED against outputs of the
  imported reference (tests/golden/reference_goldens.npz, produced by
  tests/golden/gen_goldens.py in the build container).
* spectrogram / melscale_fbanks / melspectrogram / fbank: **parity unpinned**.
  The reference delegates to MindSpore 2.3.0 C++ ops
  (mindspore.dataset.audio.Spectrogram / MelScale, requirements.txt:1), absent
  from /root/reference and not installable here.  Restated from the call sites
  (mindaudio/data/spectrum.py:609-698, mindaudio/data/features.py:252-263) and the
  documented semantics of those ops (torchaudio-equivalent: centre reflect pad,
  periodic Hann, |X|^power, HTK triangular filters with norm=None); cross-checked
  against torch.stft in tests/test_oracle.py.

Each function cites the reference lines it follows.  Two flavours exist where the
CPU baseline needs them: `*_ref` keeps the reference's loop structure (what
"mindaudio's own NumPy path" costs), `*_vec` is an honest vectorised NumPy version.
"""
from __future__ import annotations

import math
import struct

import numpy as np
from scipy.signal import get_window

MAX_BLOCK_BYTES = 256 * 1024  # spectrum.py:22


# --------------------------------------------------------------------------
# wav reading (plumbing for cfg 1; mindaudio/data/io.py:552-745 restated for the
# PCM16/PCM32 mono/multi-channel RIFF case only)
# --------------------------------------------------------------------------
def read_wav(path):
    with open(path, "rb") as fh:
        blob = fh.read()
    if blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file: %s" % path)
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(blob):
        cid, size = blob[pos:pos + 4], struct.unpack("<I", blob[pos + 4:pos + 8])[0]
        body = blob[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError("missing fmt/data chunk")
    tag, nch, rate, _, _, bits = fmt
    if tag != 1 or bits not in (16, 32):
        raise ValueError("only integer PCM16/PCM32 supported by the oracle reader")
    raw = np.frombuffer(data, dtype="<i2" if bits == 16 else "<i4")
    if nch > 1:
        raw = raw.reshape(-1, nch)
    # io.py:741-745: int16 -> /32768, int32 -> /2147483648, float64 result
    return raw / (32768.0 if bits == 16 else 2147483648.0), rate


# --------------------------------------------------------------------------
# frame  (spectrum.py:281-304)
# --------------------------------------------------------------------------
def frame_ref(x, frame_length=2048, hop_length=64):
    """Reference loop structure: one strided gather per in-frame position."""
    if hop_length < 1:
        raise ValueError("Invalid hop_length: {:d}".format(hop_length))
    n = (x.shape[-1] - frame_length) // hop_length + 1
    out = np.zeros(x.shape[:-1] + (frame_length, n))  # always float64 (spectrum.py:299)
    span = n * hop_length
    for pos in range(frame_length):
        out[..., pos, :] = x[..., pos:pos + span:hop_length]
    return out


def frame_vec(x, frame_length, hop_length):
    if hop_length < 1:
        raise ValueError("Invalid hop_length: {:d}".format(hop_length))
    n = (x.shape[-1] - frame_length) // hop_length + 1
    v = np.lib.stride_tricks.sliding_window_view(x, frame_length, axis=-1)[..., ::hop_length, :][..., :n, :]
    return np.swapaxes(v, -1, -2)  # (..., frame_length, n)


# --------------------------------------------------------------------------
# stft  (spectrum.py:125-278)
# --------------------------------------------------------------------------
def _centered_window(window, win_length, n_fft):
    w = get_window(window, win_length, fftbins=True)  # spectrum.py:173 (periodic)
    if win_length > n_fft:  # spectrum.py:331-334
        raise ValueError("Target size ({:d}) must be at least input size ({:d})".format(n_fft, win_length))
    left = (n_fft - win_length) // 2  # spectrum.py:326-329
    return np.pad(w, (left, n_fft - win_length - left))


def stft_vec(waveforms, n_fft=512, win_length=None, hop_length=None, window="hann",
             center=True, pad_mode="constant", return_complex=True):
    """Net semantics of spectrum.stft: pad n_fft//2 each side with `pad_mode`
    (default zeros: signature spectrum.py:132), frames every hop, periodic window
    centred in n_fft, rFFT along the frame axis; float64 compute, complex64 store."""
    waveforms = np.asarray(waveforms)
    win_length = n_fft if win_length is None else win_length  # :167-168
    hop_length = win_length // 4 if hop_length is None else hop_length  # :170-171
    w = _centered_window(window, win_length, n_fft)
    if n_fft > waveforms.shape[-1]:  # :182-187 / :243-246
        raise ValueError("n_fft={} is too large for input signal of length={}".format(n_fft, waveforms.shape[-1]))
    x = waveforms
    if center:
        pads = [(0, 0)] * (x.ndim - 1) + [(n_fft // 2, n_fft // 2)]
        x = np.pad(x, pads, mode=pad_mode)
    fr = frame_vec(x.astype(np.float64, copy=False), n_fft, hop_length)  # (..., n_fft, T)
    spec = np.fft.rfft(fr * w[:, None], axis=-2).astype(np.complex64)
    if return_complex:
        return spec
    return np.stack((spec.real, spec.imag), -1)  # :278


def stft_ref(waveforms, n_fft=512, win_length=None, hop_length=None, window="hann",
             center=True, pad_mode="constant", return_complex=True):
    """Same result as stft_vec, but keeping the reference's cost structure
    (spectrum.py:181-273): head/tail frames padded separately, 512-iteration
    framing loop (frame_ref), per-column-block rFFT with the 256 KiB block rule,
    Fortran-ordered complex64 result."""
    waveforms = np.asarray(waveforms)
    win_length = n_fft if win_length is None else win_length
    hop_length = win_length // 4 if hop_length is None else hop_length
    w = _centered_window(window, win_length, n_fft)
    wcol = w.reshape([1] * (waveforms.ndim - 1) + [n_fft, 1])
    half = n_fft // 2
    n = waveforms.shape[-1]
    if n_fft > n:
        raise ValueError("n_fft={} is too large for input signal of length={}".format(n_fft, n))
    head = tail = None
    start = 0
    if center:
        lead = [(0, 0)] * (waveforms.ndim - 1)
        n_head = int(math.ceil(half / hop_length))  # frames touching the left pad (:193)
        first_tail = (n + half - n_fft) // hop_length + 1  # first frame touching the right pad (:196)
        if first_tail <= n_head:  # :198-203
            waveforms = np.pad(waveforms, lead + [(half, half)], mode=pad_mode)
        else:
            start = n_head * hop_length - half  # :209
            pre = np.pad(waveforms[..., :(n_head - 1) * hop_length - half + n_fft + 1],
                         lead + [(half, 0)], mode=pad_mode)
            head = frame_ref(pre, n_fft, hop_length)[..., :n_head]
            if first_tail * hop_length - half + n_fft <= n + half:  # :223-226
                post = np.pad(waveforms[..., first_tail * hop_length - half:],
                              lead + [(0, half)], mode=pad_mode)
                tail = frame_ref(post, n_fft, hop_length)
            else:  # reference's else-branch (:235-239) is broken (tuple.shape); unreachable for n >= n_fft
                tail = np.zeros(head.shape[:-1] + (0,))
    body = frame_ref(waveforms[..., start:], n_fft, hop_length)
    n_extra = (head.shape[-1] if head is not None else 0) + (tail.shape[-1] if tail is not None else 0)
    shape = list(body.shape)
    shape[-2] = 1 + half
    shape[-1] += n_extra
    out = np.empty(shape, order="F", dtype=np.complex64)  # :252
    off = 0
    if head is not None:
        off = head.shape[-1]
        out[..., :off] = np.fft.rfft(wcol * head, axis=-2)
        if tail.shape[-1] > 0:
            out[..., -tail.shape[-1]:] = np.fft.rfft(wcol * tail, axis=-2)
    per_col = int(np.prod(body.shape[:-1])) * body.itemsize
    cols = max(MAX_BLOCK_BYTES // per_col, 1)  # :265-267
    for a in range(0, body.shape[-1], cols):  # :269-273
        b = min(a + cols, body.shape[-1])
        out[..., a + off:b + off] = np.fft.rfft(wcol * body[..., a:b], axis=-2)
    if return_complex:
        return out
    return np.stack((out.real, out.imag), -1)


# --------------------------------------------------------------------------
# amplitude_to_dB  (spectrum.py:25-90)
# --------------------------------------------------------------------------
def amplitude_to_dB(spec, stype="power", ref=1.0, amin=1e-10, top_db=80.0):
    spec = np.asarray(spec)
    if np.issubdtype(spec.dtype, np.complexfloating):  # :59-64 — raises, does not warn
        raise UserWarning("amplitude_to_db was called on complex input so phase information will be discarded.")
    ref_value = ref(spec) if callable(ref) else np.abs(ref)
    mult = 10.0 if stype == "power" else 20.0  # :74
    db = mult * np.log10(np.clip(spec, a_min=amin, a_max=None))
    db -= mult * np.log10(max(amin, ref_value))  # in place: keeps the input dtype (:77)
    if top_db is not None:
        shp = db.shape
        ch = shp[-3] if len(shp) > 2 else 1  # :82 — for (B, F, T) this is B: batch-global floor
        g = db.reshape((-1, ch, shp[-2], shp[-1]))
        floor = g.max(axis=(-3, -2, -1)) - top_db
        db = np.maximum(g, floor.reshape((-1, 1, 1, 1))).reshape(shp)
    return db


# --------------------------------------------------------------------------
# Spectrogram / MelScale / melspectrogram / fbank   (PARITY UNPINNED, see header)
# --------------------------------------------------------------------------
def hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def hz_to_mel_slaney(f):
    """Slaney (Auditory Toolbox) warp, as torchaudio `_hz_to_mel(mel_scale="slaney")` which MindSpore's MelScale mirrors:
    linear below 1 kHz at 200/3 Hz per mel, logarithmic above with 27 mels per factor 6.4."""
    f = np.asarray(f, dtype=np.float64)
    lin = 3.0 * f / 200.0
    log = 15.0 + 27.0 * np.log(np.maximum(f, 1e-300) / 1000.0) / np.log(6.4)
    return np.where(f >= 1000.0, log, lin)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * 6.4 ** ((m - 15.0) / 27.0), 200.0 * m / 3.0)


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm="none", mel_type="htk"):
    """Triangular filterbank, shape (n_freqs, n_mels); defaults HTK / norm=None (spectrum.py:625-626).
    fb[f, m] = max(0, min((f - f_m)/(f_{m+1}-f_m), (f_{m+2} - f)/(f_{m+2}-f_{m+1})))
    with all_freqs = linspace(0, sample_rate // 2, n_freqs) (SURVEY row a3;
    call site spectrum.py:686-694).  norm="slaney": column m times 2 / (f_{m+2} - f_m)."""
    to_mel, to_hz = (hz_to_mel_htk, mel_to_hz_htk) if mel_type == "htk" else (hz_to_mel_slaney, mel_to_hz_slaney)
    all_freqs = np.linspace(0.0, float(sample_rate // 2), n_freqs)
    m_pts = np.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)
    f_pts = to_hz(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]  # (n_freqs, n_mels+2)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    if norm == "slaney":
        for m in range(n_mels):
            fb[:, m] *= 2.0 / (f_pts[m + 2] - f_pts[m])
    return fb


def spectrogram(waveforms, n_fft=400, win_length=None, hop_length=None, pad=0, window="hann",
                power=2.0, normalized=False, center=True, pad_mode="reflect", onesided=True):
    """spectrum.py:547-606 → msaudio.Spectrogram semantics."""
    x = np.asarray(waveforms, dtype=np.float64)
    win_length = win_length if win_length else n_fft  # :590
    hop_length = hop_length if hop_length else win_length // 2  # :591
    if not onesided:
        raise NotImplementedError("oracle covers onesided=True only (the fbank path)")
    if pad > 0:
        x = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)])
    w = _centered_window(window, win_length, n_fft)
    if center:
        x = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(n_fft // 2, n_fft // 2)], mode=pad_mode)
    fr = frame_vec(x, n_fft, hop_length)
    spec = np.fft.rfft(fr * w[:, None], axis=-2)
    if normalized:
        spec = spec / np.sqrt(np.sum(w ** 2))
    mag = np.abs(spec)
    return mag ** power if power != 1.0 else mag


def melspectrogram(waveforms, n_fft=400, win_length=None, hop_length=None, pad=0, window="hann",
                   power=2.0, normalized=False, center=True, pad_mode="reflect", onesided=True,
                   n_mels=128, sample_rate=16000, f_min=0, f_max=None, norm="none", mel_type="htk"):
    """spectrum.py:609-698 (hop default win_length // 2, :666)."""
    win_length = win_length if win_length is not None else n_fft
    hop_length = hop_length if hop_length is not None else win_length // 2
    f_max = f_max if f_max is not None else sample_rate // 2
    spec = spectrogram(waveforms, n_fft, win_length, hop_length, pad, window, power, normalized,
                       center, pad_mode, onesided)
    fb = melscale_fbanks(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate, norm, mel_type)
    return np.einsum("fm,...ft->...mt", fb, spec)


def fbank(waveforms, deltas=False, context=False, n_mels=40, n_fft=400, sample_rate=16000,
          f_min=0.0, f_max=None, left_frames=5, right_frames=5, win_length=None,
          hop_length=None, window="hann"):
    """features.py:196-270 with deltas=False, context=False (the in-tree usage)."""
    if deltas or context:
        raise NotImplementedError("deltas/context are outside the hot path (SURVEY §8 row a5)")
    mel = melspectrogram(waveforms, n_fft=n_fft, win_length=win_length, hop_length=hop_length,
                         window=window, n_mels=n_mels, sample_rate=sample_rate, f_min=f_min, f_max=f_max)
    return amplitude_to_dB(mel, stype="power", ref=1.0, top_db=80.0)  # features.py:263


def fbank_ref_cost(waveforms, n_mels=80, n_fft=512, sample_rate=16000, hop_length=160):
    """CPU-baseline flavour (R): reference-faithful cost structure for cfg 2 —
    stft_ref (float64 framing loop + per-column rFFT) on the reflect-padded wave,
    |X|^2, dense mel matmul, amplitude_to_dB.  Same numbers as fbank()."""
    x = np.asarray(waveforms)
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(n_fft // 2, n_fft // 2)], mode="reflect")
    spec = stft_ref(xp, n_fft=n_fft, hop_length=hop_length, center=False)
    power = spec.real.astype(np.float64) ** 2 + spec.imag.astype(np.float64) ** 2
    fb = melscale_fbanks(n_fft // 2 + 1, 0.0, sample_rate // 2, n_mels, sample_rate)
    mel = np.einsum("fm,...ft->...mt", fb, power)
    return amplitude_to_dB(mel, stype="power", ref=1.0, top_db=80.0)


# --------------------------------------------------------------------------
# Kaldi-style fbank of examples/conformer/dataset.py:56-168
# --------------------------------------------------------------------------
def kaldi_mel_banks(num_bins, n_fft_padded, sample_freq, low_freq, high_freq):
    """dataset.py:68-113: triangles in the mel domain, mel(f)=1127 ln(1+f/700);
    returns (num_bins, n_fft_padded//2 + 1) with a zero last column, and centre freqs."""
    n_bins_fft = n_fft_padded // 2
    width = sample_freq / n_fft_padded
    lo = 1127.0 * math.log(1.0 + low_freq / 700.0)
    hi = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (hi - lo) / (num_bins + 1)
    b = np.arange(num_bins).reshape(-1, 1)
    left, centre, right = lo + b * delta, lo + (b + 1.0) * delta, lo + (b + 2.0) * delta
    mel = (1127.0 * np.log(1.0 + width * np.arange(n_bins_fft) / 700.0))[None, :]
    up = (mel - left) / (centre - left)
    down = (right - mel) / (right - centre)
    tri = np.where(up > down, down, up)
    tri = np.where(tri < 0, 0, tri)
    tri = np.pad(tri, ((0, 0), (0, 1)), "constant")
    return tri, (700.0 * (np.exp(centre / 1127.0) - 1.0)).reshape(-1)


def preemphasis(signal, coeff=0.97):
    """dataset.py:117-119 (whole-signal first-order high-pass; y[0] = x[0])."""
    signal = np.asarray(signal, dtype=np.float64)
    out = signal.copy()
    out[1:] -= coeff * signal[:-1]
    return out


def kaldi_window(frame_len):
    return np.power(np.hanning(frame_len), 0.85)  # dataset.py:126


def kaldi_num_frames(num_samples, frame_len, frame_shift):
    return int(np.floor((num_samples - frame_len) / frame_shift) + 1)  # dataset.py:127


def compute_fbank_feats(wav, sample_rate=16000, frame_len=25, frame_shift=10, mel_bin=80,
                        n_fft=512, per_frame_loop=False):
    """dataset.py:159-168.  `per_frame_loop=True` keeps the reference's Python
    per-frame enframe loop (:129-131) for the CPU baseline flavour (R)."""
    y = preemphasis(wav)
    flen = sample_rate * frame_len // 1000
    fshift = sample_rate * frame_shift // 1000
    nfr = kaldi_num_frames(y.size, flen, fshift)
    win = kaldi_window(flen)
    if per_frame_loop:
        frames = np.zeros((nfr, flen))
        for t in range(nfr):
            frames[t, :] = y[t * fshift:t * fshift + flen]
            frames[t, :] = frames[t, :] * win
    else:
        frames = np.lib.stride_tricks.sliding_window_view(y, flen)[::fshift][:nfr] * win
    frames = frames - np.mean(frames)  # ONE scalar over all windowed frames (:165)
    power = np.abs(np.fft.rfft(frames, n=n_fft)) ** 2  # :137-138
    # dataset.py:152,167: fs passed is sample_rate/2, the bank is built for fs*2 with 20..8000 Hz
    banks, _ = kaldi_mel_banks(mel_bin, 512, (sample_rate / 2) * 2, 20, 8000)
    feats = power @ banks.T
    feats = np.where(feats == 0, np.finfo(float).eps, feats)  # :154
    return np.log(feats)


# --------------------------------------------------------------------------
# collate helpers (mindaudio/utils/common.py:10-88, mindaudio/utils/mask.py:19-67)
# --------------------------------------------------------------------------
def pad_sequence(sequences, batch_first=True, padding_value=0, padding_max_len=None, atype=np.int32):
    trailing = sequences[0].shape[1:]
    longest = max(s.shape[0] for s in sequences)
    if padding_max_len is not None:
        longest = padding_max_len
    shape = (len(sequences), longest) + trailing if batch_first else (longest, len(sequences)) + trailing
    out = np.full(shape, padding_value, dtype=atype)
    for i, s in enumerate(sequences):
        keep = min(s.shape[0], longest)  # over-long sequences are truncated (common.py:44)
        if batch_first:
            out[i, :keep, ...] = s[:keep]
        else:
            out[:keep, i, ...] = s[:keep]
    return out


def add_sos_eos(ys, sos=0, eos=0):
    return ([np.concatenate(([sos], y)) for y in ys], [np.concatenate((y, [eos])) for y in ys])


def make_pad_mask(lengths, max_len=0):
    lengths = np.asarray(lengths)
    width = int(max_len) if max_len > 0 else int(lengths.max())  # mask.py:62
    return np.arange(width)[None, :] >= lengths[:, None]


def subsequent_mask(size):
    r = np.arange(size)
    return r[None, :] <= r[:, None]


# --------------------------------------------------------------------------
# post-processing of features.py / spectrum.py.  magphase and load_cmvn are pinned by
# tests/golden/post_goldens.npz (reference NumPy code); compute_deltas / context_window / mfcc go through MindSpore
# operators in the reference and are restated from their documented torchaudio-equivalent semantics: parity unpinned.
# --------------------------------------------------------------------------
def magphase(D, power):
    """spectrum.py:722-732 (iscomplex=True)."""
    mag = np.abs(D)
    zero = mag == 0
    nz = mag + zero
    phase = np.empty(D.shape, dtype=np.complex64)
    phase.real = D.real / nz + zero
    phase.imag = D.imag / nz
    return mag ** power, phase


def compute_deltas(x, win_length=5, pad_mode="edge"):
    """features.py:158-193 -> ComputeDeltas: sum_j j x[t + j] / (n (n + 1) (2n + 1) / 3) over the padded time axis."""
    n = (win_length - 1) // 2
    denom = n * (n + 1) * (2 * n + 1) / 3.0
    xp = np.pad(np.asarray(x, np.float64), [(0, 0)] * (x.ndim - 1) + [(n, n)], mode=pad_mode)
    t = x.shape[-1]
    out = np.zeros(x.shape, np.float64)
    for j in range(-n, n + 1):
        out += j * xp[..., n + j:n + j + t]
    return out / denom


def context_window(x, left_frames=0, right_frames=0):
    """features.py:64-155: grouped Conv1d with an identity kernel — channel f*cs + k of the output is channel f of the
    input shifted by k + max(R - L, 0) - max(L, R) frames, zero padded."""
    x = np.asarray(x)
    cs = left_frames + right_frames + 1
    off = max(right_frames - left_frames, 0) - max(left_frames, right_frames)
    f, t = x.shape[-2], x.shape[-1]
    out = np.zeros(x.shape[:-2] + (f * cs, t), x.dtype)
    for k in range(cs):
        sh = k + off
        lo, hi = max(0, -sh), min(t, t - sh)
        if hi > lo:
            out[..., k::cs, lo:hi] = x[..., :, lo + sh:hi + sh]
    return out


def create_dct(n_mfcc, n_mels, norm="ortho"):
    n = np.arange(n_mels, dtype=np.float64)
    k = np.arange(n_mfcc, dtype=np.float64)[:, None]
    dct = np.cos(np.pi / n_mels * (n + 0.5) * k)
    if norm in (None, "none"):
        dct *= 2.0
    else:
        dct[0] *= 1.0 / np.sqrt(2.0)
        dct *= np.sqrt(2.0 / n_mels)
    return dct.T


def mfcc(waveforms, deltas=True, context=True, n_mels=23, n_mfcc=20, n_fft=400, sample_rate=16000, f_min=0.0, f_max=None,
         left_frames=5, right_frames=5, win_length=None, hop_length=None, norm="ortho"):
    """features.py:273-373 (log_mels=False)."""
    mel = melspectrogram(waveforms, n_fft=n_fft, win_length=win_length, hop_length=hop_length, n_mels=n_mels,
                         sample_rate=sample_rate, f_min=f_min, f_max=f_max)
    db = amplitude_to_dB(mel, stype="power", ref=1.0, top_db=80.0)
    out = np.swapaxes(np.matmul(np.swapaxes(db, -1, -2), create_dct(n_mfcc, n_mels, norm)), -1, -2)
    if deltas:
        d1 = compute_deltas(out)
        d2 = compute_deltas(d1)
        out = np.concatenate((out, d1, d2), axis=-2)
    if context:
        out = context_window(out, left_frames, right_frames)
    return out


def load_cmvn_stats(mean_stat, var_stat, frame_num):
    """mindaudio/utils/load_files.py:9-36 on the three json fields."""
    mean = np.asarray(mean_stat, np.float64) / frame_num
    var = np.maximum(np.asarray(var_stat, np.float64) / frame_num - mean * mean, 1.0e-20)
    return mean, 1.0 / np.sqrt(var)


def istft(stft_matrix, n_fft=None, win_length=None, hop_length=None, window="hann", center=True, length=None,
          return_wss=False):
    """spectrum.py:346-474 (pinned by tests/golden/istft_goldens.npz): irfft of every frame, synthesis window, overlap-add,
    division by the window sum-square where it exceeds 1e-9, centre trimming / `length` handling."""
    D = np.asarray(stft_matrix)
    if n_fft is None:
        n_fft = 2 * (D.shape[-2] - 1)  # :400-401
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = int(win_length // 4)  # :408-409
    w = _centered_window(window, win_length, n_fft)  # get_window(fftbins=True) + _pad_center (:411-415)
    if length:
        padded = length + int(n_fft) if center else length
        n_frames = min(D.shape[-1], int(np.ceil(padded / hop_length)))  # :418-423
    else:
        n_frames = D.shape[-1]
    exp_len = n_fft + hop_length * (n_frames - 1)
    y = np.zeros(D.shape[:-2] + (exp_len,), np.float64)
    fr = np.fft.irfft(D[..., :n_frames], n=n_fft, axis=-2) * w[:, None]
    wss = np.zeros(exp_len, np.float64)
    for t in range(n_frames):
        y[..., t * hop_length:t * hop_length + n_fft] += fr[..., t]
        wss[t * hop_length:t * hop_length + n_fft] += w * w  # _window_sumsquare (:476-493)
    nz = wss > 1e-9
    y[..., nz] /= wss[nz]

    def crop(v):
        if length is None:
            return v[..., n_fft // 2:-(n_fft // 2)] if center else v
        v = v[..., (n_fft // 2 if center else 0):]
        if v.shape[-1] > length:
            return v[..., :length]
        return np.pad(v, [(0, 0)] * (v.ndim - 1) + [(0, length - v.shape[-1])])

    if return_wss:  # the divisor of every output sample: tells a test where float32 round-off is amplified
        return crop(y), crop(wss)
    return crop(y)


def resample_fft(x, num):
    """scipy.signal.resample for real input (the call behind mindaudio.data.processing.resample, processing.py:168-170),
    restated with numpy FFTs; pinned against scipy itself in tests/test_resample.py."""
    x = np.asarray(x, np.float64)
    n = x.shape[-1]
    X = np.fft.rfft(x, axis=-1)
    Y = np.zeros(x.shape[:-1] + (num // 2 + 1,), np.complex128)
    nmin = min(n, num)
    nyq = nmin // 2 + 1
    Y[..., :nyq] = X[..., :nyq]
    if nmin % 2 == 0:
        if num < n:
            Y[..., nmin // 2] *= 2.0
        elif n < num:
            Y[..., nmin // 2] *= 0.5
    return np.fft.irfft(Y, num, axis=-1) * (float(num) / float(n))
