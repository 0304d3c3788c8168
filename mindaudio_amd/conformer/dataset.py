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
    rc = lib.ma_fbank_kaldi_f32(_host.ptr(x), _host.ptr(lens), x.shape[0], x.shape[-1], x.stride(0), flen, fshift,
                                512, _host.ptr(win), bank.ref(), 0.97, _host.ptr(out), _host.ptr(ws), ws.numel(),
                                _host.current_stream_ptr())
    _lib.check(rc, "compute_fbank_feats")
    frames = t.clamp((lens - flen) // fshift + 1, min=0)
    return out, frames


def compute_fbank_feats(wav, sample_rate, frame_len, frame_shift, mel_bin):
    """Single-utterance signature of dataset.py:159: (N,) -> (num_frames, mel_bin)."""
    wav = np.asarray(wav)
    out, frames = compute_fbank_feats_batch(wav[None, :], [wav.shape[0]], sample_rate, frame_len, frame_shift,
                                            mel_bin)
    return out[0, :int(frames[0])].cpu().numpy().astype(np.float64)
