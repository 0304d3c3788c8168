"""mindaudio_amd — MI355X-native hot path of mindspore-lab/mindaudio.

Mirrors the reference's flat namespace for the functions on the hot path
(mindaudio/__init__.py:1-7 re-exports mindaudio.data.*): stft, frame, batched fbank (framing fused into the kernel),
amplitude_to_dB, melspectrogram, and the Conformer loader's Kaldi-style fbank.
Every function runs hand-written HIP kernels through the C-ABI in include/mindaudio_amd.h;
nothing here falls back to NumPy/PyTorch arithmetic.
"""
from .data.spectrum import amplitude_to_dB, frame, istft, magphase, melspectrogram, spectrogram, stft  # noqa: F401
from .data.features import compute_deltas, context_window, fbank, fbanks, mfcc  # noqa: F401

__version__ = "0.1.0"
