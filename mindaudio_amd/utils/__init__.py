"""Host-side helpers that sit either side of the hot path (mirrors of mindaudio/utils/*)."""
from .distributed import DistributedSampler  # noqa: F401
