"""Host-side mirror of ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:49-83) above the C ABI.

Same constructor arguments, same call semantics (returns monoIndex / -1 for an empty image, keypoints in the
reference's output order, N x 32 descriptor bytes), same getters.  `lib` lets tests inject a differently built
copy of the same C ABI; by default the hipcc-built liborbhip.so is loaded and its absence is an error."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, OrbHipError, OrbxConfig, ptr


class ORBextractor:
    HARRIS_SCORE, FAST_SCORE = 0, 1  # ORBextractor.h:53

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, *, device=0, max_batch=1, lib=None):
        self._L = lib if lib is not None else _lib.load()
        self.nfeatures, self.scaleFactor, self.nlevels = int(nfeatures), float(scaleFactor), int(nlevels)
        self.iniThFAST, self.minThFAST = int(iniThFAST), int(minThFAST)
        self.device, self.max_batch = device, max_batch
        self._h = None
        self._size = None

    # -- handle management: the C ABI binds a handle to one image size; re-create lazily on change
    def _handle(self, W, H, max_batch=None):
        mb = max(self.max_batch, max_batch or 1)
        if self._h is None or self._size != (W, H) or mb > self.max_batch:
            self.close()
            cfg = OrbxConfig(self.nfeatures, self.scaleFactor, self.nlevels, self.iniThFAST, self.minThFAST)
            h = C.c_void_p()
            rc = self._L.orbx_create(C.byref(cfg), W, H, mb, self.device, C.byref(h))
            if rc != 0:
                raise OrbHipError(rc, (self._L.orbx_last_error(None) or b"").decode())
            self._h, self._size, self.max_batch = h, (W, H), mb
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            self._L.orbx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise OrbHipError(rc, (self._L.orbx_last_error(self._h) or b"").decode())

    # -- ORBextractor::operator()  (ORBextractor.cc:1074-1156)
    def __call__(self, image, mask=None, vLappingArea=(0, 0)):
        """-> (monoIndex, keypoints[KP_DTYPE], descriptors uint8[n,32]); monoIndex == -1 and empty outputs for an
        empty image (ORBextractor.cc:1078-1079).  `mask` is ignored, as in the reference."""
        if image is None or getattr(image, "size", 0) == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        image = np.asarray(image)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected (ORBextractor.cc:1082)"
        if not image.flags["C_CONTIGUOUS"]:
            image = np.ascontiguousarray(image)
        H, W = image.shape
        h = self._handle(W, H)
        cap = self._L.orbx_max_keypoints(h)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n, mono = C.c_int(0), C.c_int(0)
        rc = self._L.orbx_extract(h, ptr(image), W, H, image.strides[0], int(vLappingArea[0]), int(vLappingArea[1]),
                                  ptr(kps), ptr(desc), cap, C.byref(n), C.byref(mono))
        self._check(rc)
        return mono.value, kps[:n.value].copy(), desc[:n.value].copy()

    # -- batched device-resident form (torch uint8 CUDA tensor [B,H,W]); outputs are torch tensors on the device
    def extract_batch(self, images, vLappingArea=(0, 0), out=None, stream=None):
        import torch
        assert images.is_cuda and images.dtype == torch.uint8 and images.dim() == 3 and images.is_contiguous()
        B, H, W = images.shape
        h = self._handle(W, H, max_batch=B)
        cap = self._L.orbx_max_keypoints(h)
        if out is None:
            out = (torch.empty((B, cap, 7), dtype=torch.float32, device=images.device),
                   torch.empty((B, cap, 32), dtype=torch.uint8, device=images.device),
                   torch.empty((B, 2), dtype=torch.int32, device=images.device))
        kps, desc, counts = out
        st = stream if stream is not None else torch.cuda.current_stream(images.device).cuda_stream
        rc = self._L.orbx_extract_batch_dev(h, images.data_ptr(), B, H * W, W, int(vLappingArea[0]), int(vLappingArea[1]),
                                            kps.data_ptr(), desc.data_ptr(), cap, counts.data_ptr(), C.c_void_p(st))
        self._check(rc)
        return out

    def enable_timing(self, on=True):
        """Stage events of the batch calls that follow (orbhip.h: orbx_enable_timing; off by default), for last_timing()."""
        self._check(self._L.orbx_enable_timing(self._h, 1 if on else 0))

    def last_timing(self):
        ms = np.zeros(5, np.float32)
        self._check(self._L.orbx_last_timing(self._h, ptr(ms)))
        return dict(pyramid=float(ms[0]), fast=float(ms[1]), octree=float(ms[2]), describe=float(ms[3]), total=float(ms[4]))

    def last_fast_passes(self):
        """How the last batch call ran FAST (orbhip.h: orbx_last_fast_passes): dict(two_pass, listed, tiles) — a scheduling detail, never the result."""
        v = np.zeros(3, np.uint32)
        a = v.ctypes.data
        self._check(self._L.orbx_last_fast_passes(self._h, C.c_void_p(a), C.c_void_p(a + 4), C.c_void_p(a + 8)))
        return dict(two_pass=int(v[0]), listed=int(v[1]), tiles=int(v[2]))

    # -- getters, ORBextractor.h:61-81
    def _tables(self):
        W, H = self._size if self._size else (752, 480)
        h = self._handle(W, H)
        n = self.nlevels
        t = [np.zeros(n, np.float32) for _ in range(4)] + [np.zeros(n, np.int32)]
        self._check(self._L.orbx_get_tables(h, *[ptr(a) for a in t]))
        return t

    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return np.float32(self.scaleFactor)

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        return self._tables()[4]

    # -- mvImagePyramid (public member, ORBextractor.h:83): level planes of the last call, host copies
    def pyramid_level(self, level, frame=0, border=0):
        w, hh = C.c_int(), C.c_int()
        self._check(self._L.orbx_pyramid_level(self._h, frame, level, None, C.byref(w), C.byref(hh), None))
        out = np.zeros((hh.value + 2 * border, w.value + 2 * border), np.uint8)
        self._check(self._L.orbx_copy_level(self._h, frame, level, border, ptr(out)))
        return out

    @property
    def mvImagePyramid(self):
        return [self.pyramid_level(l) for l in range(self.nlevels)]

    # -- stage taps for parity tests
    def debug_candidates(self, level, frame=0):
        n = C.c_int(0)
        self._check(self._L.orbx_debug_candidates(self._h, frame, level, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.int32)
        self._check(self._L.orbx_debug_candidates(self._h, frame, level, ptr(out), n.value, C.byref(n)))
        return out[:n.value]

    def debug_selected(self, level, frame=0):
        n = C.c_int(0)
        self._check(self._L.orbx_debug_selected(self._h, frame, level, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.int32)
        self._check(self._L.orbx_debug_selected(self._h, frame, level, ptr(out), n.value, C.byref(n)))
        return out[:n.value]


def stereo_matches(left, right, out_left, out_right, mb, mbf, stream=None):
    """Frame::ComputeStereoMatches (reference src/Frame.cc:955-1133) for the batch both extractors just processed.
    out_left / out_right: the (kps, desc, counts) tuples returned by ORBextractor.extract_batch.  -> (mvuRight, mvDepth) float32 [B, cap]"""
    import torch
    kl, dl, cl = out_left
    kr, dr, cr = out_right
    B, cap = kl.shape[0], kl.shape[1]
    ur = torch.empty((B, cap), dtype=torch.float32, device=kl.device)
    dp = torch.empty((B, cap), dtype=torch.float32, device=kl.device)
    work = torch.empty((B, cap), dtype=torch.int32, device=kl.device)
    st = stream if stream is not None else torch.cuda.current_stream(kl.device).cuda_stream
    rc = left._L.orbx_stereo_matches(left._h, right._h, kl.data_ptr(), dl.data_ptr(), cl.data_ptr(), kr.data_ptr(), dr.data_ptr(), cr.data_ptr(),
                                     cap, B, float(mb), float(mbf), ur.data_ptr(), dp.data_ptr(), work.data_ptr(), C.c_void_p(st))
    left._check(rc)
    return ur, dp


def stereo_matches_host(left, right, kl, dl, kr, dr, mb, mbf):
    """Single-pair host-array form (numpy in/out); with the emulated test build device memory == host memory."""
    cap = max(len(kl), len(kr), 1)
    K = [np.zeros((1, cap, 7), np.float32) for _ in range(2)]
    D = [np.zeros((1, cap, 32), np.uint8) for _ in range(2)]
    for i, (k, d) in enumerate(((kl, dl), (kr, dr))):
        K[i][0, :len(k)] = k.view(np.float32).reshape(-1, 7)
        D[i][0, :len(k)] = d
    cnt = [np.array([[len(kl), 0]], np.int32), np.array([[len(kr), 0]], np.int32)]
    ur = np.zeros((1, cap), np.float32); dp = np.zeros((1, cap), np.float32); work = np.zeros((1, cap), np.int32)
    rc = left._L.orbx_stereo_matches(left._h, right._h, ptr(K[0]), ptr(D[0]), ptr(cnt[0]), ptr(K[1]), ptr(D[1]), ptr(cnt[1]), cap, 1,
                                     float(mb), float(mbf), ptr(ur), ptr(dp), ptr(work), None)
    left._check(rc)
    return ur[0, :len(kl)], dp[0, :len(kl)]
