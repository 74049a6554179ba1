"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE (tests/, smoke(), bench cpu_baseline only)."""
import ctypes as C
import os
import subprocess
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "liboracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build_oracle(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = C.CDLL(_SO)
        _lib.oro_create.restype = C.c_void_p
        _lib.oro_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        _lib.oro_destroy.argtypes = [C.c_void_p]
        _lib.oro_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        _lib.oro_extract.restype = C.c_int
        _lib.oro_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.oro_level_size.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.oro_level_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib.oro_level_blurred.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib.oro_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        _lib.oro_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        _lib.oro_resize_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        _lib.oro_gaussian7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.oro_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        _lib.oro_fast_atan2.restype = C.c_float
        _lib.oro_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.oro_sincos.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        _lib.oro_cvround.argtypes = [C.c_float]
        _lib.oro_bordered_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib.oro_bench_extract_mt.restype = C.c_double
        _lib.oro_bench_extract_mt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OrbOracle:
    """CPU restatement of ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:49-59)."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.L.oro_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oro_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        f = [np.zeros(n, np.float32) for _ in range(4)]
        nf = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        self.L.oro_tables(self.h, _p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]), _p(nf), _p(um))
        return dict(scale=f[0], inv_scale=f[1], sigma2=f[2], inv_sigma2=f[3], nfeat=nf, umax=um)

    def extract(self, img, lap0=0, lap1=0, cap=None):
        """-> (monoIndex, keypoints[KP_DTYPE], descriptors[n,32])"""
        img = np.ascontiguousarray(img, np.uint8)
        H, W = img.shape
        cap = cap or (self.nfeatures + 64)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        mono = self.L.oro_extract(self.h, _p(img), W, H, W, lap0, lap1, _p(kps), _p(desc), cap, C.byref(n))
        if mono == -2:
            return self.extract(img, lap0, lap1, cap=n.value + 8)
        if mono < 0:
            return mono, kps[:0], desc[:0]
        return mono, kps[:n.value].copy(), desc[:n.value].copy()

    def level_size(self, l):
        w, h = C.c_int(), C.c_int()
        self.L.oro_level_size(self.h, l, C.byref(w), C.byref(h))
        return w.value, h.value

    def level_image(self, l):
        w, h = self.level_size(l)
        out = np.zeros((h, w), np.uint8)
        self.L.oro_level_image(self.h, l, _p(out))
        return out

    def level_blurred(self, l):
        w, h = self.level_size(l)
        out = np.zeros((h, w), np.uint8)
        ok = self.L.oro_level_blurred(self.h, l, _p(out))
        return out if ok else None

    def level_bordered(self, l):
        w, h = self.level_size(l)
        out = np.zeros((h + 38, w + 38), np.uint8)
        self.L.oro_bordered_level(self.h, l, _p(out))
        return out

    def level_candidates(self, l):
        n = self.L.oro_level_candidates(self.h, l, None, 0)
        out = np.zeros((max(n, 1), 3), np.int32)
        self.L.oro_level_candidates(self.h, l, _p(out), n)
        return out[:n]

    def level_keypoints(self, l):
        n = self.L.oro_level_keypoints(self.h, l, None, None, 0)
        kps = np.zeros(max(n, 1), KP_DTYPE)
        desc = np.zeros((max(n, 1), 32), np.uint8)
        self.L.oro_level_keypoints(self.h, l, _p(kps), _p(desc), n)
        return kps[:n], desc[:n]


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    out = np.zeros((dh, dw), np.uint8)
    lib().oro_resize_linear(_p(src), src.shape[1], src.shape[0], _p(out), dw, dh)
    return out


def gaussian7(src):
    src = np.ascontiguousarray(src, np.uint8)
    out = np.zeros_like(src)
    lib().oro_gaussian7(_p(src), src.shape[1], src.shape[0], _p(out))
    return out


def fast(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros((cap, 3), np.int32)
    n = lib().oro_fast(_p(img), img.shape[1], img.shape[0], threshold, _p(out), cap)
    return out[:n]


def fast_atan2(y, x):
    return lib().oro_fast_atan2(float(y), float(x))


def sincos(a):
    s, c = C.c_float(), C.c_float()
    lib().oro_sincos(float(a), C.byref(s), C.byref(c))
    return s.value, c.value


def bench_extract_mt(frames, nthreads, frames_per_thread, nfeatures=1000, lap=(0, 1000)):
    """-> (seconds, total keypoints) for nthreads*frames_per_thread extractions in native threads."""
    frames = np.ascontiguousarray(frames, np.uint8)
    B, H, W = frames.shape
    tot = C.c_long(0)
    s = lib().oro_bench_extract_mt(_p(frames), B, W, H, nfeatures, 1.2, 8, 20, 7, lap[0], lap[1], nthreads, frames_per_thread, C.byref(tot))
    return s, tot.value
