"""ctypes binding of liborbhip.so (C ABI in include/orbhip.h).

There is NO fallback: if the hipcc-built library is missing or fails to load, importing callers get an
OrbHipError.  (Build it with tools/build_lib.sh or __graft_entry__.build().)"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORBHIP_LIB", os.path.join(_HERE, "liborbhip.so"))   # explicit override for kernel experiments

ORB_OK, ORB_E_EMPTY_IMAGE, ORB_E_CAPACITY, ORB_E_INVALID, ORB_E_HIP, ORB_E_NOMEM, ORB_E_ABORTED = 0, -1, -2, -3, -4, -5, -6

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])  # cv::KeyPoint, 28 B


class OrbHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("orbhip error %d: %s" % (code, msg))
        self.code = code


class OrbxConfig(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


def bind(lib):
    """Declare prototypes on a loaded CDLL (include/orbhip.h)."""
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    protos = {
        "orbx_create": (i32, [C.POINTER(OrbxConfig), i32, i32, i32, i32, C.POINTER(vp)]),
        "orbx_destroy": (None, [vp]),
        "orbx_last_error": (C.c_char_p, [vp]),
        "orbx_get_tables": (i32, [vp, vp, vp, vp, vp, vp]),
        "orbx_max_keypoints": (i32, [vp]),
        "orbx_extract": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, C.POINTER(i32), C.POINTER(i32)]),
        "orbx_extract_batch_dev": (i32, [vp, vp, i32, sz, i32, i32, i32, vp, vp, i32, vp, vp]),
        "orbx_pyramid_level": (i32, [vp, i32, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "orbx_copy_level": (i32, [vp, i32, i32, i32, vp]),
        "orbx_debug_candidates": (i32, [vp, i32, i32, vp, i32, C.POINTER(i32)]),
        "orbx_debug_selected": (i32, [vp, i32, i32, vp, i32, C.POINTER(i32)]),
        "orbx_enable_timing": (i32, [vp, i32]),
        "orbx_last_timing": (i32, [vp, vp]),
        "orbx_last_fast_passes": (i32, [vp, vp, vp, vp]),
        "orbx_stereo_matches": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, C.c_float, C.c_float, vp, vp, vp, vp]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    lib._orbhip_bound = True
    return lib


_lib = None


def load():
    """Load the product library.  Raises OrbHipError if it is absent — there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OrbHipError(ORB_E_INVALID, "%s not found: build it with tools/build_lib.sh (hipcc, gfx950); "
                              "there is no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64; if liborbhip.so is dlopen'ed first it binds the system copy
        # and torch later loads a second runtime whose device pointers the first one rejects (every launch fails with ORB_E_HIP).  Importing torch
        # first makes the library's libamdhip64 dependency resolve to the copy that is already mapped.
        try:
            import torch  # noqa: F401
        except Exception:   # noqa: BLE001  (the library itself does not need torch)
            pass
        _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)
