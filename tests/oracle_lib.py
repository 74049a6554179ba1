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
    # make is incremental: a no-op when liboracle.so is newer than every source
    subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
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
        if mono == -3:
            raise ValueError("geometry on which the reference has undefined behaviour (no FAST cell or no octree root at some level)")
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
    lib().oro_bench_tune_allocator(nthreads)
    s = lib().oro_bench_extract_mt(_p(frames), B, W, H, nfeatures, 1.2, 8, 20, 7, lap[0], lap[1], nthreads, frames_per_thread, C.byref(tot))
    return s, tot.value


def bench_extract_match_mt(frames, nthreads, frames_per_thread, cam9, grid4, queries, qdesc, nq, th_dist=100, nnratio=0.9, check_ori=True,
                           do_match=True, nfeatures=1000, lap=(0, 1000)):
    """oracle/bench_oracle.cpp: extract (+ UndistortKeyPoints + grid + motion-model SearchByProjection) per frame in native threads.
    -> (seconds, total keypoints, total matches)"""
    frames = np.ascontiguousarray(frames, np.uint8)
    B, H, W = frames.shape
    L = lib()
    L.oro_bench_extract_match_mt.restype = C.c_double
    L.oro_bench_extract_match_mt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    cam9 = np.ascontiguousarray(cam9, np.float32); grid4 = np.ascontiguousarray(grid4, np.float32)
    queries = np.ascontiguousarray(queries); qdesc = np.ascontiguousarray(qdesc, np.uint8); nq = np.ascontiguousarray(nq, np.int32)
    cap_q = qdesc.shape[1]
    assert queries.nbytes == B * cap_q * 28 and qdesc.shape == (B, cap_q, 32) and nq.shape == (B,)
    kp, mt = C.c_long(0), C.c_long(0)
    s = L.oro_bench_extract_match_mt(_p(frames), B, W, H, nfeatures, 1.2, 8, 20, 7, lap[0], lap[1], _p(cam9), _p(grid4), _p(queries), _p(qdesc), _p(nq),
                                     cap_q, th_dist, nnratio, int(check_ori), int(do_match), nthreads, frames_per_thread, C.byref(kp), C.byref(mt))
    return s, kp.value, mt.value


def extract_match_frames(frames, sel, cam9, grid4, queries, qdesc, nq, mode=1, th_dist=100, nnratio=0.9, check_ori=True, do_match=True,
                         nfeatures=1000, lap=(0, 1000), nthreads=None, cap=None, cfg=(1.2, 8, 20, 7)):
    """oracle/bench_oracle.cpp oro_extract_match_frames_mt: the benchmark step's per-frame unit (ORBextractor -> UndistortKeyPoints -> grid ->
    SearchByProjection with prepared projection records) for the frames `sel` of the batch, in native threads, every output kept.
    -> dict(kps [n,cap,7] f32, desc [n,cap,32] u8, counts [n,2] i32 = {N, monoIndex}, un [n,cap,7], q_match [n,cap_q], kp_match [n,cap], nm [n])"""
    frames = np.ascontiguousarray(frames, np.uint8)
    B, H, W = frames.shape
    sel = np.ascontiguousarray(sel, np.int32)
    n = len(sel)
    cap = int(cap or (4 * nfeatures + 64))
    L = lib()
    L.oro_extract_match_frames_mt.restype = C.c_int
    L.oro_extract_match_frames_mt.argtypes = ([C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] +
                                              [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int] +
                                              [C.c_void_p] * 7)
    if do_match:
        cam9 = np.ascontiguousarray(cam9, np.float32); grid4 = np.ascontiguousarray(grid4, np.float32)
        queries = np.ascontiguousarray(queries); qdesc = np.ascontiguousarray(qdesc, np.uint8); nq = np.ascontiguousarray(nq, np.int32)
        cap_q = qdesc.shape[1]
        assert queries.nbytes == B * cap_q * 28 and qdesc.shape == (B, cap_q, 32) and nq.shape == (B,)
    else:
        cam9 = np.zeros(9, np.float32); grid4 = np.zeros(4, np.float32)
        queries = np.zeros(1, np.uint8); qdesc = np.zeros((1, 1, 32), np.uint8); nq = np.zeros(B, np.int32); cap_q = 1
    if nthreads is None:
        nthreads = max(1, min(len(os.sched_getaffinity(0)), 32, n))
    out = dict(kps=np.zeros((n, cap, 7), np.float32), desc=np.zeros((n, cap, 32), np.uint8), counts=np.zeros((n, 2), np.int32),
               un=np.zeros((n, cap, 7), np.float32), q_match=np.zeros((n, cap_q), np.int32), kp_match=np.zeros((n, cap), np.int32),
               nm=np.zeros(n, np.int32))
    rc = L.oro_extract_match_frames_mt(_p(frames), B, W, H, nfeatures, cfg[0], cfg[1], cfg[2], cfg[3], lap[0], lap[1], _p(cam9), _p(grid4), _p(queries),
                                       _p(qdesc), _p(nq), cap_q, mode, th_dist, nnratio, int(check_ori), int(do_match), int(nthreads), _p(sel), n, cap,
                                       _p(out["kps"]), _p(out["desc"]), _p(out["counts"]), _p(out["un"]), _p(out["q_match"]), _p(out["kp_match"]),
                                       _p(out["nm"]))
    if rc != 0:
        raise RuntimeError("oro_extract_match_frames_mt: frame %d exceeds the output capacity %d" % (int(sel[-rc - 1]), cap))
    return out


# ---- stage 2 (oracle/match_oracle.cpp) ------------------------------------------------------------------
QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("radius", "<f4"), ("u_right", "<f4"), ("angle", "<f4"),
                        ("min_level", "<i2"), ("max_level", "<i2"), ("flags", "<u4")])


def hamming(a, b):
    return lib().omo_hamming(_p(np.ascontiguousarray(a, np.uint8)), _p(np.ascontiguousarray(b, np.uint8)))


def grid_build(kps, grid):
    kps = np.ascontiguousarray(kps)
    gs = np.zeros(64 * 48 + 1, np.int32)
    gi = np.zeros(max(len(kps), 1), np.int32)
    L = lib()
    L.omo_grid_build.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.omo_grid_build(_p(kps), len(kps), *[float(g) for g in grid], _p(gs), _p(gi))
    return gs, gi


def search_by_projection(kps, desc, queries, qdesc, grid, mode, th_dist, nnratio, check_ori, u_right=None, occupied0=None):
    L = lib()
    L.omo_search_by_projection.restype = C.c_int
    L.omo_search_by_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float,
                                           C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                           C.c_void_p, C.c_void_p]
    kps = np.ascontiguousarray(kps); desc = np.ascontiguousarray(desc); queries = np.ascontiguousarray(queries); qdesc = np.ascontiguousarray(qdesc)
    q_match = np.zeros(max(len(queries), 1), np.int32)
    kp_match = np.zeros(max(len(kps), 1), np.int32)
    n = L.omo_search_by_projection(_p(kps), _p(desc), _p(u_right) if u_right is not None else None,
                                   _p(occupied0) if occupied0 is not None else None, len(kps), *[float(g) for g in grid],
                                   _p(queries), _p(qdesc), len(queries), mode, th_dist, nnratio, int(check_ori), _p(q_match), _p(kp_match))
    return q_match[:len(queries)], kp_match[:len(kps)], n


def search_by_bow(kf, kf_valid, f, nnratio, check_ori, f_nleft=-1):
    L = lib()
    L.omo_search_by_bow.restype = C.c_int
    L.omo_search_by_bow.argtypes = [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int]
    fN = f["desc"].shape[0]
    f_match = np.zeros(max(fN, 1), np.int32)
    n = L.omo_search_by_bow(_p(kf["desc"]), _p(kf["angle"]), _p(kf_valid), _p(kf["node_id"]), _p(kf["node_start"]), _p(kf["feat_idx"]),
                            int(kf["n_nodes"]), _p(f["desc"]), _p(f["angle"]), fN, _p(f["node_id"]), _p(f["node_start"]), _p(f["feat_idx"]),
                            int(f["n_nodes"]), nnratio, int(check_ori), _p(f_match), int(f_nleft))
    return f_match[:fN], n


def search_by_bow_kf(kf1, valid1, kf2, valid2, nnratio, check_ori):
    L = lib()
    L.omo_search_by_bow_kf.restype = C.c_int
    L.omo_search_by_bow_kf.argtypes = ([C.c_void_p] * 6 + [C.c_int, C.c_int]) * 2 + [C.c_float, C.c_int, C.c_void_p]
    n1, n2 = kf1["desc"].shape[0], kf2["desc"].shape[0]
    m12 = np.zeros(max(n1, 1), np.int32)
    args = []
    for kf, v, n in ((kf1, valid1, n1), (kf2, valid2, n2)):
        args += [_p(kf["desc"]), _p(kf["angle"]), _p(v), _p(kf["node_id"]), _p(kf["node_start"]), _p(kf["feat_idx"]), int(kf["n_nodes"]), n]
    n = L.omo_search_by_bow_kf(*args, nnratio, int(check_ori), _p(m12))
    return m12[:n1], n


def knn2(q, t):
    L = lib()
    L.omo_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    q = np.ascontiguousarray(q); t = np.ascontiguousarray(t)
    idx = np.zeros((max(len(q), 1), 2), np.int32); dist = np.zeros((max(len(q), 1), 2), np.int32)
    L.omo_knn2(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
    return idx[:len(q)], dist[:len(q)]


# ---- stage 3 (oracle/lba_oracle.cpp) ----------------------------------------------------------------------
def lba_build_system(window, cameras, huber):
    L = lib()
    L.olb_build_system.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_double,
                                   C.c_double] + [C.c_void_p] * 10
    poses = np.ascontiguousarray(window["poses"], np.float64); hidx = np.ascontiguousarray(window["pose_hidx"], np.int32)
    pts = np.ascontiguousarray(window["points"], np.float64); edges = np.ascontiguousarray(window["edges"]); cams = np.ascontiguousarray(cameras)
    npz, nl, ne = len(poses), len(pts), len(edges)
    nfree = int(hidx.max()) + 1 if (hidx >= 0).any() else 0
    o = dict(Hpp=np.zeros((max(nfree, 1), 36)), bp=np.zeros((max(nfree, 1), 6)), Hll=np.zeros((nl, 9)), bl=np.zeros((nl, 3)), Hpl=np.zeros((ne, 18)),
             err=np.zeros((ne, 3)), chi2=np.zeros(ne), rho=np.zeros((ne, 2)), depth=np.zeros(ne), robust_chi2_sum=np.zeros(1))
    L.olb_build_system(_p(poses), _p(hidx), npz, _p(pts), nl, _p(edges), ne, _p(cams), huber[0], huber[1],
                       *[_p(o[k]) for k in ("Hpp", "bp", "Hll", "bl", "Hpl", "err", "chi2", "rho", "depth", "robust_chi2_sum")])
    o["nfree"] = nfree
    return o


def lba_edge_error(pose7, point3, edge, cam, dpose, dpoint):
    L = lib()
    L.olb_edge_error.argtypes = [C.c_void_p] * 7
    e = np.zeros(3)
    L.olb_edge_error(_p(np.ascontiguousarray(pose7, np.float64)), _p(np.ascontiguousarray(point3, np.float64)), _p(np.ascontiguousarray(edge)),
                     _p(np.ascontiguousarray(cam)), _p(np.ascontiguousarray(dpose, np.float64)), _p(np.ascontiguousarray(dpoint, np.float64)), _p(e))
    return e


def quat_from_matrix(R):
    L = lib()
    L.olb_quat_from_matrix.argtypes = [C.c_void_p, C.c_void_p]
    q = np.zeros(4)
    L.olb_quat_from_matrix(_p(np.ascontiguousarray(R, np.float64)), _p(q))
    return q


def stereo_matches(oL, oR, kl, dl, kr, dr, mb, mbf):
    """Frame::ComputeStereoMatches restated (oracle/orb_oracle.cpp); oL / oR: OrbOracle objects that just extracted the pair."""
    L = lib()
    L.oro_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                     C.c_void_p, C.c_void_p]
    kl = np.ascontiguousarray(kl); kr = np.ascontiguousarray(kr); dl = np.ascontiguousarray(dl); dr = np.ascontiguousarray(dr)
    ur = np.zeros(max(len(kl), 1), np.float32); dp = np.zeros(max(len(kl), 1), np.float32)
    L.oro_stereo_matches(oL.h, oR.h, _p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), mb, mbf, _p(ur), _p(dp))
    return ur[:len(kl)], dp[:len(kl)]


def lba_optimize(window, cameras, huber, iterations):
    """SparseOptimizer::optimize(iterations) restated (oracle/lba_oracle.cpp olb_optimize) -> (poses, points, stats[4])"""
    L = lib()
    L.olb_optimize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double,
                               C.c_int, C.c_void_p]
    poses = np.ascontiguousarray(window["poses"], np.float64).copy(); pts = np.ascontiguousarray(window["points"], np.float64).copy()
    hidx = np.ascontiguousarray(window["pose_hidx"], np.int32); edges = np.ascontiguousarray(window["edges"]); cams = np.ascontiguousarray(cameras)
    stats = np.zeros(4)
    L.olb_optimize(_p(poses), _p(hidx), len(poses), _p(pts), len(pts), _p(edges), len(edges), _p(cams), huber[0], huber[1], iterations, _p(stats))
    return poses, pts, stats


def pose_optimize(pose, edges, cameras):
    """Optimizer::PoseOptimization restated (oracle/lba_oracle.cpp opo_pose_optimize) -> (pose_out[7], outlier[n] u8, n_good)"""
    L = lib()
    L.opo_pose_optimize.restype = C.c_int
    L.opo_pose_optimize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    pose = np.ascontiguousarray(pose, np.float64); edges = np.ascontiguousarray(edges); cams = np.ascontiguousarray(cameras)
    out = np.zeros(7); outl = np.zeros(max(len(edges), 1), np.uint8)
    n = L.opo_pose_optimize(_p(pose), _p(edges), len(edges), _p(cams), _p(out), _p(outl))
    return out, outl[:len(edges)], n


# ---- SURVEY N1 rows (oracle/match_oracle.cpp) -------------------------------------------------------------------------------
def search_for_initialization(kps1, desc1, kps2, desc2, grid, prev_matched, window_size, nnratio, check_ori):
    """-> (vnMatches12, nmatches, vbPrevMatched after the call)"""
    L = lib()
    L.omo_search_for_initialization.restype = C.c_int
    L.omo_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float,
                                                C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2); desc1 = np.ascontiguousarray(desc1); desc2 = np.ascontiguousarray(desc2)
    prev = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.zeros(max(len(kps1), 1), np.int32)
    n = L.omo_search_for_initialization(_p(kps1), _p(desc1), len(kps1), _p(kps2), _p(desc2), len(kps2), *[float(g) for g in grid], _p(prev),
                                        int(window_size), nnratio, int(check_ori), _p(m12))
    return m12[:len(kps1)], n, prev


def fuse(kps, desc, queries, qdesc, grid, th_dist, inv_level_sigma2=None, u_right=None):
    L = lib()
    L.omo_fuse.restype = C.c_int
    L.omo_fuse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                           C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    kps = np.ascontiguousarray(kps); desc = np.ascontiguousarray(desc); queries = np.ascontiguousarray(queries); qdesc = np.ascontiguousarray(qdesc)
    s2 = np.zeros(16, np.float32)
    if inv_level_sigma2 is not None:
        s2[:len(inv_level_sigma2)] = inv_level_sigma2
    qm = np.zeros(max(len(queries), 1), np.int32); qd = np.zeros(max(len(queries), 1), np.int32)
    n = L.omo_fuse(_p(kps), _p(desc), _p(np.ascontiguousarray(u_right, np.float32)) if u_right is not None else None, len(kps),
                   *[float(g) for g in grid], _p(queries), _p(qdesc), len(queries), th_dist, 0 if inv_level_sigma2 is None else 1, _p(s2), _p(qm), _p(qd))
    return qm[:len(queries)], qd[:len(queries)], n


class _TriSide(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("kps", "desc", "uRight", "has_mp", "node_id", "node_start", "feat")] + [("n_nodes", C.c_int), ("N", C.c_int)]


def search_for_triangulation(k1, k2, F12, ep, level_sigma2_2, scale_factors_2, only_stereo, coarse, check_ori):
    """k1/k2: dict(kps, desc, u_right|None, has_mp, node_id, node_start, feat_idx, n_nodes) of ONE key frame each -> (vMatches12, nmatches)"""
    L = lib()
    L.omo_search_for_triangulation.restype = C.c_int
    L.omo_search_for_triangulation.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p]
    keep = []

    def side(d):
        arrs = [np.ascontiguousarray(d["kps"]), np.ascontiguousarray(d["desc"]),
                np.ascontiguousarray(d["u_right"], np.float32) if d.get("u_right") is not None else None, np.ascontiguousarray(d["has_mp"], np.uint8),
                np.ascontiguousarray(d["node_id"], np.int32), np.ascontiguousarray(d["node_start"], np.int32), np.ascontiguousarray(d["feat_idx"], np.int32)]
        keep.append(arrs)
        return _TriSide(*[(a.ctypes.data if a is not None else None) for a in arrs], int(d["n_nodes"]), len(arrs[1]))
    a, b = side(k1), side(k2)
    F12 = np.ascontiguousarray(F12, np.float32).reshape(9); ep = np.ascontiguousarray(ep, np.float32)
    ls = np.zeros(16, np.float32); ls[:len(level_sigma2_2)] = level_sigma2_2
    sf = np.zeros(16, np.float32); sf[:len(scale_factors_2)] = scale_factors_2
    m12 = np.zeros(max(a.N, 1), np.int32)
    n = L.omo_search_for_triangulation(C.byref(a), C.byref(b), _p(F12), _p(ep), _p(ls), _p(sf), int(only_stereo), int(coarse), int(check_ori), _p(m12))
    return m12[:a.N], n


# ---- DBoW2 vocabulary + transform (oracle/bow_oracle.cpp) --------------------------------------------------------------
def search_for_triangulation_kb8(k1, k2, nleft1, nleft2, pair, only_stereo, coarse, check_ori):
    """KannalaBrandt8 key frames (oracle/match_oracle.cpp omo_search_for_triangulation_kb8); pair: one TRI_KB8_PAIR_DTYPE record"""
    L = lib()
    L.omo_search_for_triangulation_kb8.restype = C.c_int
    L.omo_search_for_triangulation_kb8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    keep = []

    def side(d):
        arrs = [np.ascontiguousarray(d["kps"]), np.ascontiguousarray(d["desc"]), None, np.ascontiguousarray(d["has_mp"], np.uint8),
                np.ascontiguousarray(d["node_id"], np.int32), np.ascontiguousarray(d["node_start"], np.int32), np.ascontiguousarray(d["feat_idx"], np.int32)]
        keep.append(arrs)
        return _TriSide(*[(a.ctypes.data if a is not None else None) for a in arrs], int(d["n_nodes"]), len(arrs[1]))
    a, b = side(k1), side(k2)
    pair = np.ascontiguousarray(pair)
    m12 = np.zeros(max(a.N, 1), np.int32)
    n = L.omo_search_for_triangulation_kb8(C.addressof(a), C.addressof(b), int(nleft1), int(nleft2), _p(pair), int(only_stereo), int(coarse), int(check_ori), _p(m12))
    return m12[:a.N], n


class OracleVocabulary:
    def __init__(self, file_bytes):
        L = lib()
        L.obw_load_binary.restype = C.c_void_p
        L.obw_load_binary.argtypes = [C.c_void_p, C.c_size_t]
        buf = np.frombuffer(file_bytes, np.uint8)
        self.h = L.obw_load_binary(_p(buf), buf.size)
        if not self.h:
            raise ValueError("inconsistent vocabulary file")
        info = np.zeros(6, np.int32)
        L.obw_info.argtypes = [C.c_void_p, C.c_void_p]
        L.obw_info(self.h, _p(info))
        self.k, self.L, self.scoring, self.weighting, self.n_nodes, self.n_words = [int(x) for x in info]

    def __del__(self):
        if getattr(self, "h", None):
            L = lib()
            L.obw_destroy.argtypes = [C.c_void_p]
            L.obw_destroy(self.h)
            self.h = None

    def transform(self, desc, levelsup=4):
        L = lib()
        L.obw_transform.restype = C.c_int
        L.obw_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 9
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        m = max(n, 1)
        o = dict(word_id=np.zeros(m, np.int32), node_id=np.zeros(m, np.int32), weight=np.zeros(m), fv_node_id=np.zeros(m, np.int32),
                 fv_node_start=np.zeros(m + 1, np.int32), fv_feat_idx=np.zeros(m, np.int32), fv_n_nodes=np.zeros(1, np.int32),
                 bv_word=np.zeros(m, np.int32), bv_value=np.zeros(m))
        o["bv_n"] = L.obw_transform(self.h, _p(desc), n, levelsup, *[_p(o[k]) for k in ("word_id", "node_id", "weight", "fv_node_id", "fv_node_start",
                                                                                       "fv_feat_idx", "fv_n_nodes", "bv_word", "bv_value")])
        o["fv_n_nodes"] = int(o["fv_n_nodes"][0])
        return o


def bow_score_l1(w1, v1, w2, v2):
    L = lib()
    L.obw_score_l1.restype = C.c_double
    L.obw_score_l1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    w1 = np.ascontiguousarray(w1, np.int32); w2 = np.ascontiguousarray(w2, np.int32); v1 = np.ascontiguousarray(v1); v2 = np.ascontiguousarray(v2)
    return L.obw_score_l1(_p(w1), _p(v1), len(w1), _p(w2), _p(v2), len(w2))


def grid_build_rig(kps, nleft, grid):
    kps = np.ascontiguousarray(kps)
    gs = np.zeros(2 * 64 * 48 + 1, np.int32)
    gi = np.zeros(max(len(kps), 1), np.int32)
    L = lib()
    L.omo_grid_build_rig.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.omo_grid_build_rig(_p(kps), len(kps), int(nleft), *[float(g) for g in grid], _p(gs), _p(gi))
    return gs, gi


def search_by_projection_rig(kps, desc, nleft, link, queries, qdesc, grid, mode, th_dist, nnratio, check_ori, occupied0=None):
    L = lib()
    L.omo_search_by_projection_rig.restype = C.c_int
    L.omo_search_by_projection_rig.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                               C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    kps = np.ascontiguousarray(kps); desc = np.ascontiguousarray(desc); queries = np.ascontiguousarray(queries); qdesc = np.ascontiguousarray(qdesc)
    link = None if link is None else np.ascontiguousarray(link, np.int32)
    q_match = np.zeros(max(len(queries), 1), np.int32)
    kp_match = np.zeros(max(len(kps), 1), np.int32)
    n = L.omo_search_by_projection_rig(_p(kps), _p(desc), _p(occupied0) if occupied0 is not None else None, len(kps), int(nleft),
                                       _p(link) if link is not None else None, *[float(g) for g in grid], _p(queries), _p(qdesc), len(queries),
                                       mode, th_dist, nnratio, int(check_ori), _p(q_match), _p(kp_match))
    return q_match[:len(queries)], kp_match[:len(kps)], n


# ---- Frame constructor steps (oracle/frame_oracle.cpp) ------------------------------------------------------------------------------
def undistort_keypoints(kps, cam9):
    kps = np.ascontiguousarray(kps)
    out = np.zeros_like(kps)
    L = lib()
    L.ofr_undistort_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    cam9 = np.ascontiguousarray(cam9, np.float32)
    L.ofr_undistort_keypoints(_p(kps), len(kps), _p(cam9), _p(out))
    return out


def image_bounds(cam9, cols, rows):
    out = np.zeros(6, np.float32)
    L = lib()
    L.ofr_image_bounds.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    cam9 = np.ascontiguousarray(cam9, np.float32)
    L.ofr_image_bounds(_p(cam9), cols, rows, _p(out))
    return out


def stereo_from_rgbd(kps, kps_un, depth, mbf):
    kps, kps_un, depth = np.ascontiguousarray(kps), np.ascontiguousarray(kps_un), np.ascontiguousarray(depth, np.float32)
    ur, dz = np.zeros(len(kps), np.float32), np.zeros(len(kps), np.float32)
    L = lib()
    L.ofr_stereo_from_rgbd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.ofr_stereo_from_rgbd(_p(kps), _p(kps_un), len(kps), _p(depth), depth.shape[1], float(mbf), _p(ur), _p(dz))
    return ur, dz


# ---- visual-inertial local BA (oracle/inertial_oracle.cpp) ----------------------------------------------------------------------------
def _ib(w):
    return (np.ascontiguousarray(w["kfs"]), w["rig"], np.ascontiguousarray(w["points"], np.float64), np.ascontiguousarray(w["edges"]),
            np.ascontiguousarray(w["imu"]))


def inertial_imu_error(edge, kf1, kf2, rig, d=None):
    e = np.zeros(9)
    L = lib()
    L.oib_imu_error.argtypes = [C.c_void_p] * 6
    dd = None if d is None else np.ascontiguousarray(d, np.float64)
    L.oib_imu_error(_p(np.ascontiguousarray(edge)), _p(np.ascontiguousarray(kf1)), _p(np.ascontiguousarray(kf2)), C.addressof(rig),
                    None if dd is None else _p(dd), _p(e))
    return e


def inertial_imu_jacobian(edge, kf1, kf2):
    J = np.zeros((9, 24))
    L = lib()
    L.oib_imu_jacobian.argtypes = [C.c_void_p] * 4
    L.oib_imu_jacobian(_p(np.ascontiguousarray(edge)), _p(np.ascontiguousarray(kf1)), _p(np.ascontiguousarray(kf2)), _p(J))
    return J


def inertial_vis_error(edge, kf, rig, X, dpose=None, dpoint=None, jac=False):
    e, A, B = np.zeros(3), np.zeros((3, 3)), np.zeros((3, 6))
    L = lib()
    L.oib_vis_error.argtypes = [C.c_void_p] * 9
    X = np.ascontiguousarray(X, np.float64)
    dp = None if dpose is None else np.ascontiguousarray(dpose, np.float64)
    dx = None if dpoint is None else np.ascontiguousarray(dpoint, np.float64)
    L.oib_vis_error(_p(np.ascontiguousarray(edge)), _p(np.ascontiguousarray(kf)), C.addressof(rig), _p(X), None if dp is None else _p(dp),
                    None if dx is None else _p(dx), _p(e), _p(A) if jac else None, _p(B) if jac else None)
    return (e, A, B) if jac else e


def inertial_errors(w, huber):
    kfs, rig, pts, edges, imu = _ib(w)
    vchi, vdp, ichi, rs = np.zeros(len(edges)), np.zeros(len(edges), np.uint8), np.zeros((len(imu), 3)), np.zeros(1)
    L = lib()
    L.oib_errors.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oib_errors(_p(kfs), len(kfs), C.addressof(rig), _p(pts), len(pts), _p(edges), len(edges), _p(imu), len(imu), huber[0], huber[1],
                 _p(vchi), _p(vdp), _p(ichi), _p(rs))
    return {"vis_chi2": vchi, "vis_depth_pos": vdp, "imu_chi2": ichi, "robust_chi2_sum": rs[0]}


def inertial_optimize(w, huber, lambda_init, iterations):
    kfs, rig, pts, edges, imu = _ib(w)
    kfs, pts = kfs.copy(), pts.copy()
    stats = np.zeros(5)
    L = lib()
    L.oib_optimize.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double,
                               C.c_double, C.c_int, C.c_void_p]
    L.oib_optimize(_p(kfs), len(kfs), C.addressof(rig), _p(pts), len(pts), _p(edges), len(edges), _p(imu), len(imu), huber[0], huber[1],
                   float(lambda_init), iterations, _p(stats))
    return kfs, pts, stats


def search_by_sim3(k1, d1, grid1, k2, d2, grid2, q12, q12desc, q21, q21desc):
    L = lib()
    k1, k2 = np.ascontiguousarray(k1), np.ascontiguousarray(k2)
    out = np.zeros(max(len(k1), 1), np.int32)
    f = C.c_float
    L.omo_search_by_sim3.argtypes = [C.c_void_p, C.c_void_p, C.c_int, f, f, f, f, C.c_void_p, C.c_void_p, C.c_int, f, f, f, f,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    n = L.omo_search_by_sim3(_p(k1), _p(np.ascontiguousarray(d1)), len(k1), *[float(g) for g in grid1], _p(k2), _p(np.ascontiguousarray(d2)), len(k2),
                             *[float(g) for g in grid2], _p(np.ascontiguousarray(q12)), _p(np.ascontiguousarray(q12desc)),
                             _p(np.ascontiguousarray(q21)), _p(np.ascontiguousarray(q21desc)), _p(out))
    return out[:len(k1)], n


def stereo_fisheye(kl, dl, mono_l, kr, dr, mono_r, rig_arr, level_sigma2):
    kl, kr = np.ascontiguousarray(kl), np.ascontiguousarray(kr)
    nl, nr = len(kl), len(kr)
    l2r, r2l = np.zeros(max(nl, 1), np.int32), np.zeros(max(nr, 1), np.int32)
    depth, p3d = np.zeros(max(nl, 1), np.float32), np.zeros((max(nl, 1), 3), np.float32)
    L = lib()
    L.ofr_stereo_fisheye.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rg, ls = np.ascontiguousarray(rig_arr, np.float32), np.ascontiguousarray(level_sigma2, np.float32)
    n = L.ofr_stereo_fisheye(_p(kl), _p(np.ascontiguousarray(dl)), nl, int(mono_l), _p(kr), _p(np.ascontiguousarray(dr)), nr, int(mono_r), _p(rg), _p(ls),
                             _p(l2r), _p(r2l), _p(depth), _p(p3d))
    return l2r[:nl], r2l[:nr], depth[:nl], p3d[:nl], n


POSE_EDGE_DTYPE_I = np.dtype([("xw", "<f4", (3,)), ("obs", "<f4", (3,)), ("inv_sigma2", "<f4"), ("kind", "<i2"), ("cam", "<i2")])


def pose_inertial_kf(frame, keyframe, rig, edges, imu, rec_init=False):
    """Optimizer::PoseInertialOptimizationLastKeyFrame restated -> (frame state, outlier flags, H 15x15, return value)"""
    f = np.ascontiguousarray(frame).copy()
    edges = np.ascontiguousarray(edges)
    outl = np.zeros(max(len(edges), 1), np.uint8)
    H = np.zeros((15, 15))
    L = lib()
    L.oib_pose_inertial_kf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    n = L.oib_pose_inertial_kf(_p(f), _p(np.ascontiguousarray(keyframe)), C.addressof(rig), _p(edges), len(edges), _p(np.ascontiguousarray(imu)), int(rec_init),
                               _p(outl), _p(H))
    return f, outl[:len(edges)], H, n


def pose_inertial_lastframe(frame, prev, rig, edges, imu, prior, rec_init=False):
    f, pv = np.ascontiguousarray(frame).copy(), np.ascontiguousarray(prev).copy()
    edges = np.ascontiguousarray(edges)
    outl = np.zeros(max(len(edges), 1), np.uint8)
    H = np.zeros((15, 15))
    L = lib()
    L.oib_pose_inertial_lastframe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    n = L.oib_pose_inertial_lastframe(_p(f), _p(pv), C.addressof(rig), _p(edges), len(edges), _p(np.ascontiguousarray(imu)), _p(np.ascontiguousarray(prior)),
                                      int(rec_init), _p(outl), _p(H))
    return f, pv, outl[:len(edges)], H, n
