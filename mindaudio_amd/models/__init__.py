from .conformer import ConformerEncoder  # noqa: F401
