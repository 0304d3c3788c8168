"""Drop-in mirror of mindaudio.data.features.fbank (features.py:196-270) on MI355X."""
import math

from .. import _host, _lib
from . import spectrum as _spectrum

__all__ = ["fbank", "fbanks"]


def fbank(waveforms, deltas=False, context=False, n_mels=40, n_fft=400, sample_rate=16000, f_min=0.0, f_max=None,
          left_frames=5, right_frames=5, win_length=None, hop_length=None, window="hann"):
    """Filter-bank features: melspectrogram (power 2, centred, reflect pad, HTK mel) -> amplitude_to_dB
    (stype='power', ref=1.0, top_db=80.0), fused in one kernel (+ a tile-skipping floor pass).

    Shapes as the reference: [time] / [batch, time] / [batch, channel, time] ->
    [freq, time] / [batch, freq, time] / [batch, channel, freq, time].
    deltas/context (off in every in-tree call site) are outside the hot path.
    """
    if deltas or context:
        raise NotImplementedError("deltas/context are not on the fbank->Conformer hot path (SURVEY §8 row a5)")
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
