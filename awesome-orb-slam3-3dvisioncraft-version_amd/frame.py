"""Host-side mirror of the ORB_SLAM3::Frame constructor steps between the extractor and the matcher (reference src/Frame.cc:
UndistortKeyPoints :874, ComputeImageBounds :926 + grid scalars :394-397, ComputeStereoFromRGBD :1136) above the C ABI.
Arrays are torch CUDA tensors (product path) or numpy arrays (emulated test build only)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import OrbHipError
from .matcher import GridParams, _like, _ptr, _stream


class Camera(C.Structure):
    """Pinhole::toK() + mDistCoef (k1, k2, p1, p2, k3)"""
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("dist", C.c_float * 5)]

    @classmethod
    def make(cls, fx, fy, cx, cy, dist=()):
        d = list(dist) + [0.0] * (5 - len(dist))
        return cls(fx, fy, cx, cy, (C.c_float * 5)(*d))

    def as_array(self):
        return np.array([self.fx, self.fy, self.cx, self.cy] + list(self.dist), np.float32)


def bind(lib):
    vp, i32, sz, f32 = C.c_void_p, C.c_int, C.c_size_t, C.c_float
    protos = {
        "orbf_undistort_keypoints": (i32, [vp, vp, i32, i32, i32, C.POINTER(Camera), vp, vp]),
        "orbf_image_bounds": (i32, [C.POINTER(Camera), i32, i32, C.POINTER(f32 * 4), C.POINTER(GridParams)]),
        "orbf_stereo_from_rgbd": (i32, [vp, vp, vp, i32, i32, i32, vp, sz, i32, i32, i32, f32, vp, vp, vp]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


class FrameOps:
    def __init__(self, camera, width, height, *, lib=None):
        self._L = bind(lib if lib is not None else _lib.load())
        self.camera, self.width, self.height = camera, int(width), int(height)
        b, gp = (C.c_float * 4)(), GridParams()
        self._check(self._L.orbf_image_bounds(C.byref(camera), self.width, self.height, C.byref(b), C.byref(gp)))
        self.mnMinX, self.mnMaxX, self.mnMinY, self.mnMaxY = (float(v) for v in b)
        self.grid = (gp.min_x, gp.min_y, gp.grid_w_inv, gp.grid_h_inv)   # argument of ORBmatcher.grid_build

    def _check(self, rc):
        if rc != 0:
            raise OrbHipError(rc, "orbf call failed")

    def UndistortKeyPoints(self, kps, counts, count_stride=1, out=None):
        """kps [B, cap, 7] float32 (orb_keypoint), counts int32 -> mvKeysUn, same shape"""
        B, cap = kps.shape[0], kps.shape[1]
        out = _like(kps, tuple(kps.shape), np.float32) if out is None else out
        self._check(self._L.orbf_undistort_keypoints(_ptr(kps), _ptr(counts), count_stride, cap, B, C.byref(self.camera), _ptr(out), _stream(kps)))
        return out

    def ComputeStereoFromRGBD(self, kps, kps_un, counts, depth, mbf, count_stride=1):
        """depth [B, H, W] float32 -> (mvuRight, mvDepth) [B, cap] float32"""
        B, cap = kps.shape[0], kps.shape[1]
        H, W = depth.shape[1], depth.shape[2]
        ur, dz = _like(kps, (B, cap), np.float32), _like(kps, (B, cap), np.float32)
        self._check(self._L.orbf_stereo_from_rgbd(_ptr(kps), _ptr(kps_un), _ptr(counts), count_stride, cap, B, _ptr(depth), H * W, W, W, H,
                                                  float(mbf), _ptr(ur), _ptr(dz), _stream(kps)))
        return ur, dz
