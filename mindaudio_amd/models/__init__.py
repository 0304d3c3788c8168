from .conformer import ConformerEncoder  # noqa: F401
from .ecapatdnn import EcapaTDNN  # noqa: F401
