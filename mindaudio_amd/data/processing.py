"""Drop-in mirror of mindaudio.data.processing.resample (processing.py:132-176) for the FFT method — the one
examples/conformer/dataset.py:398-406 (speed_perturb) uses — on MI355X through the C-ABI (resample.hip)."""
import math

import numpy as np

from .. import _host, _lib

__all__ = ["resample", "resample_batch"]


def resampled_length(n, orig_freq, new_freq):
    """n_samples of processing.py:164-166, with the reference's float arithmetic."""
    ratio = float(new_freq) / orig_freq
    return int(np.ceil(n * ratio))


def resample_batch(x, n_in, n_out, ni_dev=None, no_dev=None):
    """x (B, >= max(n_in)) float32 device tensor, n_in / n_out per-row lengths -> (B, max(n_out)) float32 with row b =
    scipy.signal.resample(x[b, :n_in[b]], n_out[b]) and zeros behind it.  ni_dev / no_dev: the same lengths as int32 device tensors
    when the caller has uploaded them already (the loader: no copy of its own behind a busy stream)."""
    t = _host.require_gpu()
    lib = _lib.load()
    assert x.is_cuda and x.dtype == t.float32 and x.dim() == 2 and x.stride(1) == 1
    n_in = np.asarray(n_in, np.int64)
    n_out = np.asarray(n_out, np.int64)
    b = x.shape[0]
    assert n_in.shape == (b,) and n_out.shape == (b,) and n_in.min() >= 1 and n_out.min() >= 1 and n_in.max() <= x.shape[1]
    max_in, max_out = int(n_in.max()), int(n_out.max())
    ws_bytes = lib.ma_resample_fft_workspace_bytes(b, max_in, max_out)
    _lib.check(min(ws_bytes, 0), "resample")
    ws = _host.workspace(ws_bytes, x.device)
    out = t.empty((b, max_out), dtype=t.float32, device=x.device)
    ni = ni_dev if ni_dev is not None else t.from_numpy(n_in.astype(np.int32)).to(x.device)
    no = no_dev if no_dev is not None else t.from_numpy(n_out.astype(np.int32)).to(x.device)
    rc = lib.ma_resample_fft_f32(_host.ptr(x), x.stride(0), _host.ptr(ni), _host.ptr(no), b, max_in, max_out, _host.ptr(out),
                                 out.stride(0), _host.ptr(ws), ws.numel(), _host.current_stream_ptr())
    _lib.check(rc, "resample")
    return out


def resample(waveform, orig_freq=16000, new_freq=16000, res_type="fft", lowpass_filter_width=6, rolloff=0.99, beta=None):
    """processing.resample: `[time]` or `[batch, time]`; res_type "fft" / "scipy" (scipy.signal.resample, processing.py:168-170).
    The "minddata" method goes through MindSpore's Resample in the reference and is not built."""
    if orig_freq == new_freq:
        return waveform  # processing.py:161-162
    if res_type not in ("scipy", "fft"):
        raise NotImplementedError("res_type %r uses MindSpore's Resample in the reference" % (res_type,))
    t = _host.require_gpu()
    was_numpy = not isinstance(waveform, t.Tensor)
    x = t.as_tensor(np.asarray(waveform) if was_numpy else waveform).to(device="cuda", dtype=t.float32)
    if x.dim() > 2:
        raise NotImplementedError("resample over [batch, time, channel] is not built")
    lead = tuple(x.shape[:-1])
    x2 = x.reshape((-1, x.shape[-1])).contiguous()
    n = x2.shape[-1]
    m = resampled_length(n, orig_freq, new_freq)
    y = resample_batch(x2, [n] * x2.shape[0], [m] * x2.shape[0]).reshape(lead + (m,))
    if was_numpy:
        return y.cpu().numpy().astype(np.asarray(waveform).dtype)  # np.asarray(y_hat, dtype=waveform.dtype), :170
    return y
