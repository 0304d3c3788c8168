"""Drop-in mirrors of mindaudio.data.spectrum for the functions on the hot path
(stft, melspectrogram, amplitude_to_dB), executing on MI355X through the C-ABI.

Inputs may be NumPy arrays (the reference's container; results come back as NumPy) or CUDA/HIP
torch tensors (results stay on the device).  Arithmetic is float32 on the device: the reference's
stft computes in float64 and rounds to complex64 (spectrum.py:252), so results agree to float32
round-off of the frame norm, not bit for bit.
"""
import numpy as np

from .. import _host, _lib

__all__ = ["stft", "istft", "frame", "melspectrogram", "amplitude_to_dB", "spectrogram", "magphase"]


def _finish(out, lead, was_numpy):
    out = out.reshape(lead + tuple(out.shape[1:]))
    return out.cpu().numpy() if was_numpy else out


def stft(waveforms, n_fft=512, win_length=None, hop_length=None, window="hann", center=True,
         pad_mode="constant", return_complex=True):
    """Short-time Fourier transform — same signature and defaults as spectrum.py:125-134.

    Returns (..., 1 + n_fft/2, 1 + N // hop) complex64 (or a trailing (…, 2) real/imag stack when
    return_complex is False, spectrum.py:275-278).  For a 1-D wave the memory layout is the
    reference's Fortran order (spectrum.py:252).
    """
    t = _host.require_gpu()
    lib = _lib.load()
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = win_length // 4
    x, lead, was_numpy = _host.to_device_2d(waveforms)
    n = x.shape[-1]
    win = _host.device_window(window, win_length, n_fft, x.device)  # ValueError if win_length > n_fft
    if n_fft > n:
        raise ValueError("n_fft={} is too large for input signal of length={}".format(n_fft, n))
    if hop_length < 1:
        raise ValueError("Invalid hop_length: {:d}".format(hop_length))
    if pad_mode not in _lib.PAD_MODES:
        raise ValueError("unsupported pad_mode %r" % (pad_mode,))
    n_frames = lib.ma_num_frames(n, n_fft, hop_length, int(bool(center)))
    _lib.check(min(n_frames, 0), "stft")
    n_freq = n_fft // 2 + 1
    out = t.empty((x.shape[0], n_frames, n_freq, 2), dtype=t.float32, device=x.device)
    rc = lib.ma_stft_f32(_host.ptr(x), x.shape[0], n, x.stride(0), n_fft, hop_length, _host.ptr(win),
                         int(bool(center)), _lib.PAD_MODES[pad_mode], _lib.STFT_FRAME_MAJOR, _host.ptr(out),
                         _host.current_stream_ptr())
    _lib.check(rc, "stft")
    spec = t.view_as_complex(out).transpose(1, 2)  # (B, n_freq, n_frames) view of frame-major memory
    if not return_complex:
        spec = t.stack((spec.real, spec.imag), -1)
    return _finish(spec, lead, was_numpy)


def frame(x, frame_length=2048, hop_length=64):
    """spectrum.frame (spectrum.py:281-304): overlapping frames of the last axis, frame axis = -2, time axis = -1,
    num_frame = (N - frame_length) // hop_length + 1.  Always float64, like the reference's np.zeros buffer; NumPy in -> NumPy
    out, device tensor in -> device tensor out."""
    if hop_length < 1:
        raise ValueError("Invalid hop_length: {:d}".format(hop_length))
    t = _host.require_gpu()
    was_numpy = not isinstance(x, t.Tensor)
    xt = t.as_tensor(np.ascontiguousarray(x) if was_numpy else x)
    if xt.dtype not in (t.float32, t.float64):
        xt = xt.to(t.float64)
    xt = xt.cuda()
    lead, n = tuple(xt.shape[:-1]), xt.shape[-1]
    x2 = xt.reshape(-1, n)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    num_frame = (n - frame_length) // hop_length + 1
    if num_frame < 1:
        # the reference's np.zeros raises "negative dimensions are not allowed" here
        raise ValueError("frame_length={} is too large for input signal of length={}".format(frame_length, n))
    out = t.empty((x2.shape[0], frame_length, num_frame), dtype=t.float64, device=x2.device)
    _lib.check(_lib.load().ma_frame_f64(_host.ptr(x2), 1 if x2.dtype == t.float64 else 0, x2.shape[0], n, x2.stride(0),
                                        frame_length, hop_length, _host.ptr(out), _host.current_stream_ptr()), "frame")
    out = out.reshape(lead + (frame_length, num_frame))
    return out.cpu().numpy() if was_numpy else out


def istft(stft_matrix, n_fft=None, win_length=None, hop_length=None, window="hann", center=True, length=None):
    """Inverse STFT — same signature and defaults as spectrum.py:346-354: irfft of every frame, synthesis window, overlap-add,
    division by the window sum-square, centre trimming or `length`.  stft_matrix (..., 1 + n_fft/2, frames) complex64, NumPy
    (returns float64 like the reference's np.float_ buffer, :429) or a device tensor (returns float32).  Arithmetic is float32."""
    import math

    t = _host.require_gpu()
    lib = _lib.load()
    was_numpy = not isinstance(stft_matrix, t.Tensor)
    D = t.as_tensor(np.ascontiguousarray(stft_matrix) if was_numpy else stft_matrix).to(device="cuda", dtype=t.complex64)
    if D.dim() < 2:
        raise ValueError("stft_matrix must be (..., 1 + n_fft/2, frames)")
    lead = tuple(D.shape[:-2])
    n_freq, frames_total = D.shape[-2], D.shape[-1]
    D = D.reshape((-1, n_freq, frames_total)).contiguous()
    if n_fft is None:
        n_fft = 2 * (n_freq - 1)  # :400-401
    if n_freq != n_fft // 2 + 1:
        raise ValueError("stft_matrix has %d rows, n_fft=%d needs %d" % (n_freq, n_fft, n_fft // 2 + 1))
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = int(win_length // 4)  # :408-409
    if hop_length < 1:
        raise ValueError("Invalid hop_length: {:d}".format(hop_length))
    win = _host.device_window(window, win_length, n_fft, D.device)  # get_window(fftbins=True) + _pad_center (:411-415)
    if length:
        padded = length + int(n_fft) if center else length
        n_frames = min(frames_total, int(math.ceil(padded / hop_length)))  # :418-423
    else:
        n_frames = frames_total
    exp_len = n_fft + hop_length * (n_frames - 1)
    start = n_fft // 2 if center else 0
    out_len = int(length) if length else (exp_len - 2 * (n_fft // 2) if center else exp_len)
    if out_len < 1:
        raise ValueError("istft output would be empty")
    b = D.shape[0]
    out = t.empty((b, out_len), dtype=t.float32, device=D.device)
    ws_bytes = lib.ma_istft_workspace_bytes(b, n_frames, n_fft)
    _lib.check(min(ws_bytes, 0), "istft")
    ws = _host.workspace(ws_bytes, D.device)
    rc = lib.ma_istft_f32(_host.ptr(t.view_as_real(D)), b, n_fft, frames_total, n_frames, hop_length, _host.ptr(win), start,
                          _host.ptr(out), out_len, _host.ptr(ws), ws.numel() * ws.element_size(), _host.current_stream_ptr())
    _lib.check(rc, "istft")
    out = out.reshape(lead + (out_len,))
    return out.cpu().numpy().astype(np.float64) if was_numpy else out


def _mel_args(n_fft, win_length, hop_length, window, center, pad_mode, n_mels, sample_rate, f_min, f_max, device,
              norm="none", mel_type="htk"):
    win_length = win_length if win_length is not None else n_fft  # spectrum.py:665
    hop_length = hop_length if hop_length is not None else win_length // 2  # spectrum.py:666
    f_max = f_max if f_max is not None else sample_rate // 2  # spectrum.py:770 / MelScale default
    if pad_mode not in _lib.PAD_MODES:
        raise ValueError("unsupported pad_mode %r" % (pad_mode,))
    win = _host.device_window(window, win_length, n_fft, device)
    bank = _host.device_htk_bank(n_fft, float(f_min), float(f_max), int(n_mels), int(sample_rate), device,
                                 _host.mel_enum(norm, "norm"), _host.mel_enum(mel_type, "mel_type"))
    return win_length, hop_length, win, bank


def melspectrogram(waveforms, n_fft=400, win_length=None, hop_length=None, pad=0, window="hann", power=2.0,
                   normalized=False, center=True, pad_mode="reflect", onesided=True, n_mels=128,
                   sample_rate=16000, f_min=0, f_max=None, norm="none", mel_type="htk"):
    """Mel-scaled spectrogram — signature of spectrum.py:609-627 (norm/mel_type as strings)."""
    t = _host.require_gpu()
    lib = _lib.load()
    if normalized or not onesided:
        raise NotImplementedError("only normalized=False, onesided=True are on the hot path (SURVEY §8 row a3)")
    x, lead, was_numpy = _host.to_device_2d(waveforms)
    if pad > 0:
        x = t.nn.functional.pad(x, (pad, pad))
    n = x.shape[-1]
    win_length, hop_length, win, bank = _mel_args(n_fft, win_length, hop_length, window, center, pad_mode, n_mels,
                                                  sample_rate, f_min, f_max, x.device, norm, mel_type)
    n_frames = lib.ma_num_frames(n, n_fft, hop_length, int(bool(center)))
    _lib.check(min(n_frames, 0), "melspectrogram")
    out = t.empty((x.shape[0], n_mels, n_frames), dtype=t.float32, device=x.device)
    rc = lib.ma_melspectrogram_f32(_host.ptr(x), x.shape[0], n, x.stride(0), n_fft, hop_length, _host.ptr(win),
                                   int(bool(center)), _lib.PAD_MODES[pad_mode], bank.ref(), float(power),
                                   _host.ptr(out), _host.current_stream_ptr())
    _lib.check(rc, "melspectrogram")
    return _finish(out, lead, was_numpy)


def amplitude_to_dB(wavform, stype="power", ref=1.0, amin=1e-10, top_db=80.0):
    """spectrum.py:25-90: 10/20*log10(clip(x, amin)) - mult*log10(max(amin, |ref|)), then the top_db floor
    relative to the maximum over the last three axes after the reference's reshape — for a (B, F, T) input
    that is ONE floor for the whole batch (spectrum.py:79-89)."""
    t = _host.require_gpu()
    lib = _lib.load()
    was_numpy = not isinstance(wavform, t.Tensor)
    if was_numpy:
        wavform = np.asarray(wavform)
        is_complex = np.issubdtype(wavform.dtype, np.complexfloating)
    else:
        is_complex = wavform.is_complex()
    if is_complex:  # the reference raises (not warns): spectrum.py:59-64
        raise UserWarning("amplitude_to_db was called on complex input so phase information will be discarded. "
                          "To suppress this warning, call amplitude_to_db(np.abs(D)**2) instead.")
    ref_value = float(ref(wavform)) if callable(ref) else abs(float(ref))
    mult = 10.0 if stype == "power" else 20.0
    in_dtype = wavform.dtype
    x = t.from_numpy(np.ascontiguousarray(wavform)) if was_numpy else wavform
    shape = tuple(x.shape)
    x = x.to(device="cuda", dtype=t.float32).contiguous()
    channels = shape[-3] if len(shape) > 2 else 1
    elems = channels * shape[-2] * shape[-1]
    groups = x.numel() // elems
    out = t.empty_like(x)
    ws_bytes = lib.ma_db_workspace_bytes(groups, elems)
    ws = _host.workspace(ws_bytes, x.device)
    import math

    rc = lib.ma_amplitude_to_db_f32(_host.ptr(x), groups, elems, mult, float(amin),
                                    mult * math.log10(max(amin, ref_value)),
                                    -1.0 if top_db is None else float(top_db), _host.ptr(out), _host.ptr(ws),
                                    ws.numel(), _host.current_stream_ptr())
    _lib.check(rc, "amplitude_to_dB")
    if was_numpy:
        return out.cpu().numpy().astype(in_dtype, copy=False)
    return out.to(in_dtype)


def spectrogram(waveforms, n_fft=400, win_length=None, hop_length=None, pad=0, window="hann", power=2.0,
                normalized=False, center=True, pad_mode="reflect", onesided=True):
    """spectrum.spectrogram (spectrum.py:560-606 -> MindSpore Spectrogram, torchaudio semantics): |STFT| ** power with reflect
    padding by default, hop = win_length // 2; (..., n_fft // 2 + 1, frames) float32.  normalized=True divides the STFT by
    sqrt(sum(window ** 2)) before the power.  Two-sided output is not built."""
    if not onesided:
        raise NotImplementedError("two-sided spectrograms are not built")
    t = _host.require_gpu()
    lib = _lib.load()
    was_numpy = not isinstance(waveforms, t.Tensor)
    x = t.as_tensor(np.asarray(waveforms)).cuda() if was_numpy else waveforms
    if pad > 0:
        x = t.nn.functional.pad(x, (pad, pad))
    win_length = win_length or n_fft
    hop_length = hop_length or win_length // 2
    S = stft(x, n_fft=n_fft, win_length=win_length, hop_length=hop_length, window=window, center=center, pad_mode=pad_mode)
    mag = t.empty(S.shape, dtype=t.float32, device=S.device)
    Sc = S.contiguous()
    _lib.check(lib.ma_magphase_f32(_host.ptr(t.view_as_real(Sc)), Sc.numel(), float(power), _host.ptr(mag), None,
                                   _host.current_stream_ptr()), "spectrogram")
    if normalized:
        w = _host.centred_window_f64(window, win_length, n_fft)
        scale = float(np.sum(w * w)) ** (-0.5 * float(power))
        _lib.check(lib.ma_pointwise_f32(_host.ptr(mag), mag.numel(), 0, scale, 0.0, _host.ptr(mag), _host.current_stream_ptr()),
                   "spectrogram normalisation")
    return mag.cpu().numpy() if was_numpy else mag


def magphase(waveform, power, iscomplex=True):
    """spectrum.magphase (spectrum.py:701-735).  Complex input: (|D| ** power, D / |D|), phase 1+0j where D == 0; real (..., 2)
    input (iscomplex=False): (|D| ** power, atan2(im, re))."""
    t = _host.require_gpu()
    lib = _lib.load()
    was_numpy = not isinstance(waveform, t.Tensor)
    if not iscomplex:
        # real (..., 2) input -> MindSpore Magphase (spectrum.py:732-735; torchaudio semantics: magnitude ** power and the ANGLE)
        z = t.as_tensor(np.ascontiguousarray(waveform) if was_numpy else waveform).to(device="cuda", dtype=t.float32).contiguous()
        if z.shape[-1] != 2:
            raise ValueError("magphase(iscomplex=False) takes a (..., 2) real/imaginary stack")
        mag = t.empty(z.shape[:-1], dtype=t.float32, device=z.device)
        ang = t.empty(z.shape[:-1], dtype=t.float32, device=z.device)
        _lib.check(lib.ma_magphase_angle_f32(_host.ptr(z), mag.numel(), float(power), _host.ptr(mag), _host.ptr(ang),
                                             _host.current_stream_ptr()), "magphase")
        return (mag.cpu().numpy(), ang.cpu().numpy()) if was_numpy else (mag, ang)
    z = t.as_tensor(np.ascontiguousarray(waveform) if was_numpy else waveform).to(device="cuda", dtype=t.complex64).contiguous()
    mag = t.empty(z.shape, dtype=t.float32, device=z.device)
    phase = t.empty_like(z)
    zr, pr = t.view_as_real(z), t.view_as_real(phase)
    _lib.check(lib.ma_magphase_f32(_host.ptr(zr), z.numel(), float(power), _host.ptr(mag), _host.ptr(pr),
                                   _host.current_stream_ptr()), "magphase")
    if was_numpy:
        return mag.cpu().numpy(), phase.cpu().numpy()
    return mag, phase
