from ._lib import load, OrbHipError, KP_DTYPE  # noqa: F401
from .extractor import ORBextractor, stereo_matches  # noqa: F401
from .matcher import ORBmatcher, QUERY_DTYPE  # noqa: F401
from .lba import LbaWindows, synth_window  # noqa: F401
from .frame import FrameOps, Camera  # noqa: F401

__all__ = ["load", "OrbHipError", "KP_DTYPE", "ORBextractor", "stereo_matches", "ORBmatcher", "QUERY_DTYPE", "LbaWindows", "synth_window", "FrameOps", "Camera"]
