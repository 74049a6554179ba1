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
        "orbm_undistort_and_grid_build": (i32, [vp, vp, i32, i32, i32, C.POINTER(Camera), C.POINTER(GridParams), vp, vp, vp, vp]),
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

    def UndistortAndGrid(self, kps, counts, count_stride=1, out=None):
        """UndistortKeyPoints + AssignFeaturesToGrid in one launch (orbm_undistort_and_grid_build).
        -> (mvKeysUn [B, cap, 7] f32, grid_start [B, 64*48+1] i32, grid_idx [B, cap] i32), the same arrays as the two separate calls"""
        from .matcher import GRID_COLS, GRID_ROWS
        B, cap = kps.shape[0], kps.shape[1]
        un, gs, gi = out if out is not None else (_like(kps, tuple(kps.shape), np.float32), _like(kps, (B, GRID_COLS * GRID_ROWS + 1), np.int32),
                                                  _like(kps, (B, cap), np.int32))
        gp = GridParams(*self.grid)
        self._check(self._L.orbm_undistort_and_grid_build(_ptr(kps), _ptr(counts), count_stride, cap, B, C.byref(self.camera), C.byref(gp), _ptr(un),
                                                          _ptr(gs), _ptr(gi), _stream(kps)))
        return un, gs, gi

    def ComputeStereoFromRGBD(self, kps, kps_un, counts, depth, mbf, count_stride=1):
        """depth [B, H, W] float32 -> (mvuRight, mvDepth) [B, cap] float32"""
        B, cap = kps.shape[0], kps.shape[1]
        H, W = depth.shape[1], depth.shape[2]
        ur, dz = _like(kps, (B, cap), np.float32), _like(kps, (B, cap), np.float32)
        self._check(self._L.orbf_stereo_from_rgbd(_ptr(kps), _ptr(kps_un), _ptr(counts), count_stride, cap, B, _ptr(depth), H * W, W, W, H,
                                                  float(mbf), _ptr(ur), _ptr(dz), _stream(kps)))
        return ur, dz


class FisheyeRig(C.Structure):
    """KannalaBrandt8 parameters of mpCamera / mpCamera2, mRlr / mtlr (Frame.cc:1242-1243), mvLevelSigma2"""
    _fields_ = [("k_left", C.c_float * 8), ("k_right", C.c_float * 8), ("R_lr", C.c_float * 9), ("t_lr", C.c_float * 3), ("level_sigma2", C.c_float * 16)]

    @classmethod
    def make(cls, k_left, k_right, R_lr, t_lr, level_sigma2):
        ls = list(level_sigma2) + [0.0] * (16 - len(level_sigma2))
        return cls((C.c_float * 8)(*[float(v) for v in k_left]), (C.c_float * 8)(*[float(v) for v in k_right]),
                   (C.c_float * 9)(*[float(v) for v in np.asarray(R_lr).reshape(-1)]), (C.c_float * 3)(*[float(v) for v in t_lr]), (C.c_float * 16)(*ls))

    def as_array(self):
        return np.array(list(self.k_left) + list(self.k_right) + list(self.R_lr) + list(self.t_lr), np.float32)


def ComputeStereoFishEyeMatches(kps_l, desc_l, n_l, mono_l, kps_r, desc_r, n_r, mono_r, rig, *, lib=None, count_stride=1):
    """Frame::ComputeStereoFishEyeMatches for a batch of fisheye stereo frames -> (mvLeftToRightMatch [B,capL], mvRightToLeftMatch [B,capR],
    mvDepth [B,capL], mvStereo3Dpoints [B,capL,3], nMatches [B])"""
    L = lib if lib is not None else _lib.load()
    fn = L.orbf_stereo_fisheye_matches
    vp, i32 = C.c_void_p, C.c_int
    fn.restype = i32
    fn.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, C.POINTER(FisheyeRig), vp, vp, vp, vp, vp, vp]
    B, capL, capR = kps_l.shape[0], kps_l.shape[1], kps_r.shape[1]
    l2r, r2l = _like(kps_l, (B, capL), np.int32), _like(kps_l, (B, capR), np.int32)
    depth, p3d, nm = _like(kps_l, (B, capL), np.float32), _like(kps_l, (B, capL, 3), np.float32), _like(kps_l, (B,), np.int32)
    rc = fn(_ptr(kps_l), _ptr(desc_l), _ptr(n_l), _ptr(mono_l), _ptr(kps_r), _ptr(desc_r), _ptr(n_r), _ptr(mono_r), capL, capR, count_stride, B,
            C.byref(rig), _ptr(l2r), _ptr(r2l), _ptr(depth), _ptr(p3d), _ptr(nm), _stream(kps_l))
    if rc != 0:
        raise OrbHipError(rc, "orbf_stereo_fisheye_matches failed")
    return l2r, r2l, depth, p3d, nm
