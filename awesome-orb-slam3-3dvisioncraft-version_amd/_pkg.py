from ._lib import load, OrbHipError, KP_DTYPE  # noqa: F401
from .extractor import ORBextractor  # noqa: F401

__all__ = ["load", "OrbHipError", "KP_DTYPE", "ORBextractor"]
