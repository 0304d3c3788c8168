"""Host-side metrics next to the hot path (mirror of mindaudio/metric)."""
from .wer import wer  # noqa: F401
