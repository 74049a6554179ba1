"""Host-side mirror of ORB_SLAM3::ORBmatcher (reference include/ORBmatcher.h:39-94) above the C ABI.

The reference methods take Frame&/KeyFrame*/MapPoint* pointer graphs; here they take the flattened records that the
C++ adapter gathers from those objects (include/orbhip.h "Stage 2"), batched over independent problems.  Arrays may be
torch CUDA tensors (product path) or numpy arrays (only meaningful with the emulated test build, whose "device" is
host memory).  Outputs are allocated like the inputs."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import OrbHipError

TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30   # ORBmatcher.cc:36-38
GRID_COLS, GRID_ROWS = 64, 48                 # Frame.h:38-39
MODE_LOCAL_MAP, MODE_BEST_ONLY, MODE_INIT = 0, 1, 2
Q_VALID, Q_STEREO, Q_HAS_OBS, Q_RIGHT, Q_TWIN = 1, 2, 4, 8, 16

QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("radius", "<f4"), ("u_right", "<f4"), ("angle", "<f4"),
                        ("min_level", "<i2"), ("max_level", "<i2"), ("flags", "<u4")])
assert QUERY_DTYPE.itemsize == 28


class GridParams(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("grid_w_inv", C.c_float), ("grid_h_inv", C.c_float)]


class SearchParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("th_dist", C.c_int32), ("nn_ratio", C.c_float), ("check_orientation", C.c_int32),
                ("grid", GridParams)]


class BowSide(C.Structure):
    _fields_ = [("desc", C.c_void_p), ("angle", C.c_void_p), ("node_id", C.c_void_p), ("node_start", C.c_void_p),
                ("feat_idx", C.c_void_p), ("n_nodes", C.c_void_p), ("cap_f", C.c_int32), ("cap_nodes", C.c_int32), ("n_left", C.c_void_p)]


class FuseParams(C.Structure):
    _fields_ = [("th_dist", C.c_int32), ("chi2_gate", C.c_int32), ("grid", GridParams), ("inv_level_sigma2", C.c_float * 16)]


class TriSide(C.Structure):
    _fields_ = [("kps", C.c_void_p), ("desc", C.c_void_p), ("u_right", C.c_void_p), ("has_mp", C.c_void_p), ("node_id", C.c_void_p),
                ("node_start", C.c_void_p), ("feat_idx", C.c_void_p), ("n_nodes", C.c_void_p), ("cap_f", C.c_int32), ("cap_nodes", C.c_int32)]


TRI_PAIR_DTYPE = np.dtype([("F12", "<f4", (9,)), ("ep", "<f4", (2,)), ("level_sigma2_2", "<f4", (16,)), ("scale_factors_2", "<f4", (16,)),
                           ("reserved", "<f4")])
assert TRI_PAIR_DTYPE.itemsize == 176


TRI_KB8_PAIR_DTYPE = np.dtype([("n_cams", "<i4"), ("reserved", "<i4"), ("k1", "<f4", (2, 8)), ("k2", "<f4", (2, 8)), ("R12", "<f4", (4, 9)),
                               ("t12", "<f4", (4, 3)), ("ep", "<f4", (2,)), ("level_sigma2_1", "<f4", (16,)), ("level_sigma2_2", "<f4", (16,)),
                               ("scale_factors_2", "<f4", (16,))])
assert TRI_KB8_PAIR_DTYPE.itemsize == 528


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)
    assert a.is_contiguous()
    return C.c_void_p(a.data_ptr())


def _like(a, shape, dtype):
    if isinstance(a, np.ndarray):
        return np.zeros(shape, dtype)
    import torch
    tdt = {np.int32: torch.int32, np.uint16: torch.int16, np.uint8: torch.uint8, np.float32: torch.float32}[dtype]
    return torch.zeros(shape, dtype=tdt, device=a.device)


def _stream(a):
    if isinstance(a, np.ndarray):
        return None
    import torch
    return C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)


def bind(lib):
    vp, i32, sz, f32 = C.c_void_p, C.c_int, C.c_size_t, C.c_float
    protos = {
        "orbm_hamming": (i32, [vp, i32, vp, i32, i32, vp, vp]),
        "orbm_knn2": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp]),
        "orbm_grid_build": (i32, [vp, vp, i32, i32, i32, C.POINTER(GridParams), vp, vp, vp]),
        "orbm_search_workspace_bytes": (sz, [i32, i32]),
        "orbm_search_by_projection": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, i32, C.POINTER(SearchParams),
                                            vp, vp, vp, vp, vp]),
        "orbm_search_by_bow": (i32, [C.POINTER(BowSide), vp, C.POINTER(BowSide), i32, f32, i32, vp, vp, vp]),
        "orbm_search_by_bow_kf": (i32, [C.POINTER(BowSide), vp, C.POINTER(BowSide), vp, i32, f32, i32, vp, vp, vp]),
        "orbm_enable_timing": (i32, [i32]),
        "orbm_last_timing": (i32, [vp]),
        "orbm_grid_build_rig": (i32, [vp, vp, vp, i32, i32, i32, C.POINTER(GridParams), vp, vp, vp]),
        "orbm_search_by_projection_rig": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, i32, C.POINTER(SearchParams),
                                                vp, vp, vp, vp, vp]),
        "orbm_fuse": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, i32, C.POINTER(FuseParams), vp, vp, vp, vp]),
        "orbm_search_for_triangulation": (i32, [C.POINTER(TriSide), C.POINTER(TriSide), vp, i32, i32, i32, i32, vp, vp, vp]),
        "orbm_search_for_triangulation_kb8": (i32, [C.POINTER(TriSide), C.POINTER(TriSide), vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
        "orbm_mutual_matches": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


class ORBmatcher:
    TH_HIGH, TH_LOW, HISTO_LENGTH = TH_HIGH, TH_LOW, HISTO_LENGTH

    def __init__(self, nnratio=0.6, checkOri=True, *, lib=None):   # ORBmatcher.h:39
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)
        self._L = bind(lib if lib is not None else _lib.load())

    def _check(self, rc):
        if rc != 0:
            raise OrbHipError(rc, "orbm call failed")

    # -- measurement facility: device time of the last grid build / projection search kernels (HIP events on the launch stream)
    def enable_timing(self, on=True):
        self._check(self._L.orbm_enable_timing(int(bool(on))))

    def last_timing(self):
        ms = np.zeros(3, np.float32)
        self._check(self._L.orbm_last_timing(ms.ctypes.data_as(C.c_void_p)))
        return dict(grid_build=float(ms[0]), sbp_candidates=float(ms[1]), sbp_resolve=float(ms[2]))

    # -- ORBmatcher::DescriptorDistance for all pairs (ORBmatcher.cc:2700-2716): q [B,nq,32], t [B,nt,32] -> [B,nq,nt] uint16
    def DescriptorDistance(self, q, t):
        B, nq, _ = q.shape
        nt = t.shape[1]
        out = _like(q, (B, nq, nt), np.uint16)
        self._check(self._L.orbm_hamming(_ptr(q), nq, _ptr(t), nt, B, _ptr(out), _stream(q)))
        return out

    # -- BFMatcher(NORM_HAMMING).knnMatch(k=2) (Frame.cc:1300): q [B,capq,32], nq [B] int32, t [B,capt,32], nt [B]
    def knnMatch2(self, q, nq, t, nt):
        B, capq, _ = q.shape
        idx = _like(q, (B, capq, 2), np.int32)
        dist = _like(q, (B, capq, 2), np.int32)
        self._check(self._L.orbm_knn2(_ptr(q), _ptr(nq), capq, _ptr(t), _ptr(nt), t.shape[1], 1, B, _ptr(idx), _ptr(dist), _stream(q)))
        return idx, dist

    # -- Frame::AssignFeaturesToGrid (Frame.cc:444-478): kps [B,cap,7] f32 (orb_keypoint), counts int32 (stride in elements)
    def grid_build(self, kps, counts, grid, count_stride=1, out=None):
        B, cap = kps.shape[0], kps.shape[1]
        gs, gi = out if out is not None else (_like(kps, (B, GRID_COLS * GRID_ROWS + 1), np.int32), _like(kps, (B, cap), np.int32))
        gp = GridParams(*grid)
        self._check(self._L.orbm_grid_build(_ptr(kps), _ptr(counts), count_stride, cap, B, C.byref(gp), _ptr(gs), _ptr(gi), _stream(kps)))
        return gs, gi

    # -- SearchByProjection (ORBmatcher.cc:59-255 mode LOCAL_MAP / :2244-2509 mode BEST_ONLY) on flattened records
    def SearchByProjection(self, kps, desc, counts, grid_start, grid_idx, queries, qdesc, nq, grid, mode, th_dist=TH_HIGH,
                           u_right=None, occupied0=None, count_stride=1, work=None, out=None):
        """out = (q_match [B,cap_q], kp_match [B,cap_k], nmatches [B]) int32 buffers of an earlier call may be passed back in (every entry is rewritten)."""
        B, cap_k = kps.shape[0], kps.shape[1]
        cap_q = qdesc.shape[1]
        q_match, kp_match, nmatches = out if out is not None else (_like(kps, (B, cap_q), np.int32), _like(kps, (B, cap_k), np.int32), _like(kps, (B,), np.int32))
        if work is None:
            work = _like(kps, (self._L.orbm_search_workspace_bytes(B, cap_q),), np.uint8)
        prm = SearchParams(mode, th_dist, self.mfNNratio, int(self.mbCheckOrientation), GridParams(*grid))
        self._check(self._L.orbm_search_by_projection(_ptr(kps), _ptr(desc), _ptr(u_right), _ptr(occupied0), _ptr(counts), count_stride,
                                                      cap_k, _ptr(grid_start), _ptr(grid_idx), _ptr(queries), _ptr(qdesc), _ptr(nq),
                                                      cap_q, B, C.byref(prm), _ptr(q_match), _ptr(kp_match), _ptr(nmatches),
                                                      _ptr(work), _stream(kps)))
        return q_match, kp_match, nmatches

    # -- fisheye rig (Nleft != -1): kps/desc = [left | right] concatenated, n_left [B]; grid with 2*64*48 cells
    def grid_build_rig(self, kps, counts, n_left, grid, count_stride=1):
        B, cap = kps.shape[0], kps.shape[1]
        gs = _like(kps, (B, 2 * GRID_COLS * GRID_ROWS + 1), np.int32)
        gi = _like(kps, (B, cap), np.int32)
        gp = GridParams(*grid)
        self._check(self._L.orbm_grid_build_rig(_ptr(kps), _ptr(counts), _ptr(n_left), count_stride, cap, B, C.byref(gp), _ptr(gs), _ptr(gi),
                                                _stream(kps)))
        return gs, gi

    def SearchByProjectionRig(self, kps, desc, counts, grid_start, grid_idx, queries, qdesc, nq, grid, mode, th_dist=TH_HIGH, kp_link=None,
                              occupied0=None, count_stride=1):
        """Rig twins of SearchByProjection (ORBmatcher.cc:184-251 / :2403-2460): queries in map-point order, a right-camera query flagged
        Q_RIGHT | Q_TWIN directly after its left one."""
        B, cap_k = kps.shape[0], kps.shape[1]
        cap_q = qdesc.shape[1]
        q_match = _like(kps, (B, cap_q), np.int32)
        kp_match = _like(kps, (B, cap_k), np.int32)
        nmatches = _like(kps, (B,), np.int32)
        work = _like(kps, (self._L.orbm_search_workspace_bytes(B, cap_q),), np.uint8)
        prm = SearchParams(mode, th_dist, self.mfNNratio, int(self.mbCheckOrientation), GridParams(*grid))
        self._check(self._L.orbm_search_by_projection_rig(_ptr(kps), _ptr(desc), _ptr(occupied0), _ptr(kp_link), _ptr(counts), count_stride, cap_k,
                                                          _ptr(grid_start), _ptr(grid_idx), _ptr(queries), _ptr(qdesc), _ptr(nq), cap_q, B,
                                                          C.byref(prm), _ptr(q_match), _ptr(kp_match), _ptr(nmatches), _ptr(work), _stream(kps)))
        return q_match, kp_match, nmatches

    # -- SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (ORBmatcher.cc:323-587) on FeatureVector CSRs
    def SearchByBoW(self, kf, kf_valid, f):
        """kf / f: dict(desc [B,cap,32], angle [B,cap], node_id [B,capn], node_start [B,capn+1], feat_idx [B,cap], n_nodes [B])"""
        def side(d):
            return BowSide(_ptr(d["desc"]).value, _ptr(d["angle"]).value, _ptr(d["node_id"]).value, _ptr(d["node_start"]).value,
                           _ptr(d["feat_idx"]).value, _ptr(d["n_nodes"]).value, d["desc"].shape[1], d["node_id"].shape[1],
                           _ptr(d["n_left"]).value if d.get("n_left") is not None else None)
        B = kf["desc"].shape[0]
        f_match = _like(f["desc"], (B, f["desc"].shape[1]), np.int32)
        nmatches = _like(f["desc"], (B,), np.int32)
        a, b = side(kf), side(f)
        self._check(self._L.orbm_search_by_bow(C.byref(a), _ptr(kf_valid), C.byref(b), B, self.mfNNratio, int(self.mbCheckOrientation),
                                               _ptr(f_match), _ptr(nmatches), _stream(f["desc"])))
        return f_match, nmatches

    # -- SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (ORBmatcher.cc:984-1124; LoopClosing.cc:697) on FeatureVector CSRs
    def SearchByBoWKF(self, kf1, valid1, kf2, valid2):
        """kf1 / kf2 as in SearchByBoW; valid1 / valid2 [B,cap] u8 = feature holds a good map point (and is a left-camera feature on a rig).
        -> (vpMatches12 as indices into key frame 2 or -1 [B,cap1] int32, nmatches [B])"""
        def side(d):
            return BowSide(_ptr(d["desc"]).value, _ptr(d["angle"]).value, _ptr(d["node_id"]).value, _ptr(d["node_start"]).value,
                           _ptr(d["feat_idx"]).value, _ptr(d["n_nodes"]).value, d["desc"].shape[1], d["node_id"].shape[1], None)
        B = kf1["desc"].shape[0]
        m12 = _like(kf1["desc"], (B, kf1["desc"].shape[1]), np.int32)
        nmatches = _like(kf1["desc"], (B,), np.int32)
        a, b = side(kf1), side(kf2)
        self._check(self._L.orbm_search_by_bow_kf(C.byref(a), _ptr(valid1), C.byref(b), _ptr(valid2), B, self.mfNNratio,
                                                  int(self.mbCheckOrientation), _ptr(m12), _ptr(nmatches), _stream(kf1["desc"])))
        return m12, nmatches

    # -- SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:838-979)
    def SearchForInitialization(self, kps1, desc1, n1, kps2, desc2, n2, grid_start2, grid_idx2, prev_matched, grid, windowSize=10):
        """kps1/kps2 [B,cap,7] f32 (mvKeysUn), prev_matched [B,cap1,2] f32 (vbPrevMatched, updated in place like the reference).
        -> (vnMatches12 [B,cap1] int32, nmatches [B])"""
        B, cap1 = kps1.shape[0], kps1.shape[1]
        if isinstance(kps1, np.ndarray):
            q = np.zeros((B, cap1), QUERY_DTYPE)
            q["u"], q["v"], q["radius"], q["angle"] = prev_matched[..., 0], prev_matched[..., 1], float(windowSize), kps1[..., 3]
            q["flags"] = np.where(kps1.view(np.int32)[..., 5] == 0, Q_VALID, 0)
            queries = q.view(np.uint8).reshape(B, -1)
        else:
            import torch
            q = torch.zeros((B, cap1, 7), dtype=torch.float32, device=kps1.device)
            q[..., 0], q[..., 1], q[..., 2], q[..., 4] = prev_matched[..., 0], prev_matched[..., 1], float(windowSize), kps1[..., 3]
            q.view(torch.int32)[..., 6] = (kps1.view(torch.int32)[..., 5] == 0).to(torch.int32) * Q_VALID
            queries = q
        q_match, _, nm = self.SearchByProjection(kps2, desc2, n2, grid_start2, grid_idx2, queries, desc1, n1, grid, MODE_INIT, TH_LOW)
        # :972-975  vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt for the matched ones
        if isinstance(kps1, np.ndarray):
            bi, qi = np.nonzero(q_match >= 0)
            prev_matched[bi, qi] = kps2[bi, q_match[bi, qi], :2]
        else:
            import torch
            bi, qi = torch.nonzero(q_match >= 0, as_tuple=True)
            prev_matched[bi, qi] = kps2[bi, q_match[bi, qi].long(), :2]
        return q_match, nm

    # -- Fuse (search half; ORBmatcher.cc:1630-1882 with chi2_gate, :1884-2006 without)
    def Fuse(self, kps, desc, counts, grid_start, grid_idx, queries, qdesc, nq, grid, inv_level_sigma2=None, u_right=None, th_dist=TH_LOW,
             count_stride=1):
        B, cap_k = kps.shape[0], kps.shape[1]
        cap_q = qdesc.shape[1]
        q_match = _like(kps, (B, cap_q), np.int32)
        q_dist = _like(kps, (B, cap_q), np.int32)
        nfused = _like(kps, (B,), np.int32)
        prm = FuseParams(th_dist, 0 if inv_level_sigma2 is None else 1, GridParams(*grid),
                         (C.c_float * 16)(*([float(v) for v in inv_level_sigma2] + [0.0] * 16)[:16] if inv_level_sigma2 is not None else [0.0] * 16))
        self._check(self._L.orbm_fuse(_ptr(kps), _ptr(desc), _ptr(u_right), _ptr(counts), count_stride, cap_k, _ptr(grid_start), _ptr(grid_idx),
                                      _ptr(queries), _ptr(qdesc), _ptr(nq), cap_q, B, C.byref(prm), _ptr(q_match), _ptr(q_dist), _ptr(nfused),
                                      _stream(kps)))
        return q_match, q_dist, nfused

    # -- SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (ORBmatcher.cc:2008-2220)
    def SearchBySim3(self, kf1, kf2, q12, q12desc, q21, q21desc):
        """kf1 / kf2: dict(kps [B,cap,7], desc, counts [B], grid_start, grid_idx, grid).  q12[b][i1] = map point of key frame 1's keypoint i1
        projected into key frame 2 (VALID iff it exists, is not already matched and passed the gates of :2044-2080); q21 likewise the other
        way.  One query slot per keypoint (cap_q = cap_k).  -> (vpMatches12 as indices into key frame 2 or -1 [B,cap1], nFound [B])"""
        n1, n2 = kf1["counts"], kf2["counts"]
        m12, _, _ = self.Fuse(kf2["kps"], kf2["desc"], n2, kf2["grid_start"], kf2["grid_idx"], q12, q12desc, n1, kf2["grid"], th_dist=TH_HIGH)
        m21, _, _ = self.Fuse(kf1["kps"], kf1["desc"], n1, kf1["grid_start"], kf1["grid_idx"], q21, q21desc, n2, kf1["grid"], th_dist=TH_HIGH)
        B, cap1, cap2 = m12.shape[0], m12.shape[1], m21.shape[1]
        out = _like(m12, (B, cap1), np.int32)
        nfound = _like(m12, (B,), np.int32)
        self._check(self._L.orbm_mutual_matches(_ptr(m12), _ptr(m21), _ptr(n1), _ptr(n2), cap1, cap2, B, _ptr(out), _ptr(nfound), _stream(m12)))
        return out, nfound

    # -- SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse) (ORBmatcher.cc:1138-1428), pinhole / one camera
    def SearchForTriangulation(self, kf1, kf2, pairs, bOnlyStereo=False, bCoarse=False):
        """kf1 / kf2: dict(kps [B,cap,7], desc [B,cap,32], u_right [B,cap] or None, has_mp [B,cap] u8, node_id, node_start, feat_idx, n_nodes);
        pairs: u8 view of TRI_PAIR_DTYPE[B].  -> (vMatches12 [B,cap1] int32, nmatches [B])"""
        def side(d):
            return TriSide(_ptr(d["kps"]).value, _ptr(d["desc"]).value, _ptr(d.get("u_right")).value if d.get("u_right") is not None else None,
                           _ptr(d["has_mp"]).value, _ptr(d["node_id"]).value, _ptr(d["node_start"]).value, _ptr(d["feat_idx"]).value,
                           _ptr(d["n_nodes"]).value, d["desc"].shape[1], d["node_id"].shape[1])
        B = kf1["desc"].shape[0]
        m12 = _like(kf1["desc"], (B, kf1["desc"].shape[1]), np.int32)
        nm = _like(kf1["desc"], (B,), np.int32)
        a, b = side(kf1), side(kf2)
        self._check(self._L.orbm_search_for_triangulation(C.byref(a), C.byref(b), _ptr(pairs), B, int(bOnlyStereo), int(bCoarse),
                                                          int(self.mbCheckOrientation), _ptr(m12), _ptr(nm), _stream(kf1["desc"])))
        return m12, nm

    # -- SearchForTriangulation on KannalaBrandt8 key frames (monocular fisheye or rig with mpCamera2): ORBmatcher.cc:1138-1428 rig branches
    def SearchForTriangulationKB8(self, kf1, kf2, n_left1, n_left2, pairs, bOnlyStereo=False, bCoarse=False):
        """kf1 / kf2 as in SearchForTriangulation with kps = [mvKeys | mvKeysRight] (u_right ignored); n_left* [B] int32 (NLeft; unused when
        n_cams = 1); pairs: u8 view of TRI_KB8_PAIR_DTYPE[B].  -> (vMatches12 [B,cap1] int32, nmatches [B])"""
        def side(d):
            return TriSide(_ptr(d["kps"]).value, _ptr(d["desc"]).value, None, _ptr(d["has_mp"]).value, _ptr(d["node_id"]).value,
                           _ptr(d["node_start"]).value, _ptr(d["feat_idx"]).value, _ptr(d["n_nodes"]).value, d["desc"].shape[1], d["node_id"].shape[1])
        B = kf1["desc"].shape[0]
        m12 = _like(kf1["desc"], (B, kf1["desc"].shape[1]), np.int32)
        nm = _like(kf1["desc"], (B,), np.int32)
        a, b = side(kf1), side(kf2)
        self._check(self._L.orbm_search_for_triangulation_kb8(C.byref(a), C.byref(b), _ptr(n_left1), _ptr(n_left2), _ptr(pairs), B, int(bOnlyStereo),
                                                              int(bCoarse), int(self.mbCheckOrientation), _ptr(m12), _ptr(nm), _stream(kf1["desc"])))
        return m12, nm
