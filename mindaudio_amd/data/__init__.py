from .features import fbank, fbanks  # noqa: F401
from .spectrum import amplitude_to_dB, frame, melspectrogram, stft  # noqa: F401
