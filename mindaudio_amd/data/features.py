"""Drop-in mirror of mindaudio.data.features.fbank (features.py:196-270) on MI355X."""
import math

import numpy as np

from .. import _host, _lib
from . import spectrum as _spectrum

__all__ = ["fbank", "fbanks", "mfcc", "compute_deltas", "context_window"]


def fbank(waveforms, deltas=False, context=False, n_mels=40, n_fft=400, sample_rate=16000, f_min=0.0, f_max=None,
          left_frames=5, right_frames=5, win_length=None, hop_length=None, window="hann"):
    """Filter-bank features: melspectrogram (power 2, centred, reflect pad, HTK mel) -> amplitude_to_dB
    (stype='power', ref=1.0, top_db=80.0), fused in one kernel (+ a tile-skipping floor pass).

    Shapes as the reference: [time] / [batch, time] / [batch, channel, time] ->
    [freq, time] / [batch, freq, time] / [batch, channel, freq, time].
    deltas/context (off in every in-tree call site) are outside the hot path.
    """
    if deltas or context:  # features.py:264-270: appended after the dB features
        out = fbank(waveforms, False, False, n_mels, n_fft, sample_rate, f_min, f_max, left_frames, right_frames,
                    win_length, hop_length, window)
        t_ = _host.require_gpu()
        was_np = not isinstance(out, t_.Tensor)
        od = t_.as_tensor(out).cuda() if was_np else out
        if deltas:
            d1 = compute_deltas(od)
            d2 = compute_deltas(d1)
            od = t_.cat((od, d1, d2), dim=-2)
        if context:
            od = context_window(od, left_frames, right_frames)
        return od.cpu().numpy() if was_np else od
    t = _host.require_gpu()
    lib = _lib.load()
    x, lead, was_numpy = _host.to_device_2d(waveforms)
    if len(lead) >= 2:
        # [batch, channel, time]: the top_db floor is per batch entry (spectrum.py:82-86) -> unfused path
        mel = _spectrum.melspectrogram(x.reshape(lead + (x.shape[-1],)), n_fft=n_fft, win_length=win_length,
                                       hop_length=hop_length, window=window, n_mels=n_mels,
                                       sample_rate=sample_rate, f_min=f_min, f_max=f_max)
        out = _spectrum.amplitude_to_dB(mel, stype="power", ref=1.0, top_db=80.0)
        return out.cpu().numpy() if was_numpy else out
    n = x.shape[-1]
    win_length, hop_length, win, bank = _spectrum._mel_args(n_fft, win_length, hop_length, window, True, "reflect",
                                                            n_mels, sample_rate, f_min, f_max, x.device)
    n_frames = lib.ma_num_frames(n, n_fft, hop_length, 1)
    _lib.check(min(n_frames, 0), "fbank")
    out = t.empty((x.shape[0], n_mels, n_frames), dtype=t.float32, device=x.device)
    ws_bytes = lib.ma_fbank_workspace_bytes(x.shape[0], n_frames)
    ws = _host.workspace(ws_bytes, x.device)
    amin, ref, top_db, mult = 1e-10, 1.0, 80.0, 10.0  # features.py:263 / spectrum.py:25
    rc = lib.ma_fbank_db_f32(_host.ptr(x), x.shape[0], n, x.stride(0), n_fft, hop_length, _host.ptr(win), 1,
                             _lib.PAD_MODES["reflect"], bank.ref(), 2.0, mult, amin,
                             mult * math.log10(max(amin, ref)), top_db, _host.ptr(out), _host.ptr(ws), ws.numel(),
                             _host.current_stream_ptr())
    _lib.check(rc, "fbank")
    out = out.reshape(lead + tuple(out.shape[1:]))
    return out.cpu().numpy() if was_numpy else out


fbanks = fbank  # README.md:41 and the docstring (features.py:247) spell it `fbanks`


# ---- post-processing of features.py -----------------------------------------------------------------
def _as_device_f32(x):
    t = _host.require_gpu()
    was_numpy = not isinstance(x, t.Tensor)
    xd = t.as_tensor(np.asarray(x) if was_numpy else x).to(device="cuda", dtype=t.float32).contiguous()
    return t, xd, was_numpy


def _back(out, was_numpy):
    return out.cpu().numpy() if was_numpy else out


def compute_deltas(specgram, win_length=5, pad_mode="edge"):
    """features.compute_deltas (features.py:158-193): delta coefficients along the last (time) axis."""
    if win_length < 3:
        raise ValueError("win_length must be no less than 3")
    t, x, was_numpy = _as_device_f32(specgram)
    out = t.empty_like(x)
    tlen = x.shape[-1]
    _lib.check(_lib.load().ma_compute_deltas_f32(_host.ptr(x), x.numel() // tlen, tlen, int(win_length),
                                                 _lib.PAD_MODES[pad_mode], _host.ptr(out), _host.current_stream_ptr()),
               "compute_deltas")
    return _back(out, was_numpy)


def context_window(waveforms, left_frames=0, right_frames=0):
    """features.context_window (features.py:64-155): [freq, time] / [batch, freq, time] / [batch, channel, freq, time]
    -> the same with freq * (left + right + 1) rows."""
    t, x, was_numpy = _as_device_f32(waveforms)
    shape = tuple(x.shape)
    if len(shape) not in (2, 3, 4):
        raise TypeError("Input dimension must be 2, 3 or 4, but got {}".format(len(shape)))
    f, tlen = shape[-2], shape[-1]
    batch = x.numel() // (f * tlen)
    cs = left_frames + right_frames + 1
    out = t.empty(shape[:-2] + (f * cs, tlen), dtype=t.float32, device=x.device)
    _lib.check(_lib.load().ma_context_window_f32(_host.ptr(x), batch, f, tlen, int(left_frames), int(right_frames),
                                                 _host.ptr(out), _host.current_stream_ptr()), "context_window")
    return _back(out, was_numpy)


def _create_dct(n_mfcc, n_mels, norm):
    """create_dct of mindspore.dataset.audio.utils (= torchaudio.functional.create_dct): (n_mels, n_mfcc) DCT-II."""
    n = np.arange(n_mels, dtype=np.float64)
    k = np.arange(n_mfcc, dtype=np.float64)[:, None]
    dct = np.cos(math.pi / n_mels * (n + 0.5) * k)
    if norm in (None, "none"):
        dct *= 2.0
    else:
        dct[0] *= 1.0 / math.sqrt(2.0)
        dct *= math.sqrt(2.0 / n_mels)
    return np.ascontiguousarray(dct.T).astype(np.float32)


def mfcc(waveforms, deltas=True, context=True, n_mels=23, n_mfcc=20, n_fft=400, sample_rate=16000, f_min=0.0, f_max=None,
         left_frames=5, right_frames=5, win_length=None, hop_length=None, norm="ortho", log_mels=False):
    """features.mfcc (features.py:273-373): melspectrogram -> dB (or log) -> DCT [-> deltas, delta-deltas] [-> context]."""
    if n_mfcc > n_mels:
        raise ValueError("The number of MFCC coefficients must be no more than # mel bins.")
    t = _host.require_gpu()
    was_numpy = not isinstance(waveforms, t.Tensor)
    mel = _spectrum.melspectrogram(waveforms if not was_numpy else t.as_tensor(np.asarray(waveforms)).cuda(),
                                   sample_rate=sample_rate, n_fft=n_fft, n_mels=n_mels, f_min=f_min, f_max=f_max,
                                   win_length=win_length, hop_length=hop_length)
    if log_mels:  # np.log(melspec + 1e-6), features.py:343-344
        mel = mel.contiguous()
        _lib.check(_lib.load().ma_pointwise_f32(_host.ptr(mel), mel.numel(), 1, 1e-6, 1.0, _host.ptr(mel),
                                                _host.current_stream_ptr()), "log mel")
    else:
        mel = _spectrum.amplitude_to_dB(mel, stype="power", ref=1.0, top_db=80.0)
    shape = tuple(mel.shape)
    tlen = shape[-1]
    batch = mel.numel() // (n_mels * tlen)
    dct = t.from_numpy(_create_dct(n_mfcc, n_mels, norm)).to(mel.device)
    out = t.empty(shape[:-2] + (n_mfcc, tlen), dtype=t.float32, device=mel.device)
    _lib.check(_lib.load().ma_dct_f32(_host.ptr(mel.contiguous()), batch, n_mels, tlen, _host.ptr(dct), n_mfcc, _host.ptr(out),
                                      _host.current_stream_ptr()), "mfcc dct")
    if deltas:
        d1 = compute_deltas(out)
        d2 = compute_deltas(d1)
        out = t.cat((out, d1, d2), dim=-2)
    if context:
        out = context_window(out, left_frames, right_frames)
    return _back(out, was_numpy)
