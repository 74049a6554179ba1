"""Host-side mirror of the LocalBundleAdjustment linearisation above the C ABI (include/orbhip.h "Stage 3").

`LbaWindow` holds the flattened g2o graph of Optimizer::LocalBundleAdjustment (reference src/Optimizer.cc:1957-2193):
poses (SE3Quat: t, q), points, landmark-major edges and the two CSR views (== BlockSolver::buildStructure).
`build_system` == BlockSolver::buildSystem (block_solver.hpp:502-560); `compute_errors` == computeActiveErrors.
Arrays may be torch CUDA tensors (product) or numpy arrays (emulated test build only)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import OrbHipError
from .matcher import _like, _ptr, _stream

EDGE_MONO, EDGE_STEREO, EDGE_BODY = 0, 1, 2
CAM_PINHOLE, CAM_KB8 = 0, 1
EDGE_DTYPE = np.dtype([("pose", "<i4"), ("point", "<i4"), ("kind", "<i2"), ("cam", "<i2"), ("obs", "<f4", (3,)), ("inv_sigma2", "<f4")])
CAM_DTYPE = np.dtype([("model", "<i4"), ("reserved", "<i4"), ("p", "<f8", (8,)), ("bf", "<f8"), ("trl_q", "<f8", (4,)), ("trl_t", "<f8", (3,))])
assert EDGE_DTYPE.itemsize == 28 and CAM_DTYPE.itemsize == 136
HUBER_MONO = float(np.float32(np.sqrt(5.991)))     # const float thHuberMono = sqrt(5.991)   Optimizer.cc:2052
HUBER_STEREO = float(np.float32(np.sqrt(7.815)))   # const float thHuberStereo = sqrt(7.815) Optimizer.cc:2053


class LbaProblem(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("poses", "pose_hidx", "points", "edges", "lm_start", "pose_start", "pose_edges", "cameras",
                                          "n_poses", "n_points", "n_edges")] + \
               [("cap_p", C.c_int32), ("cap_l", C.c_int32), ("cap_e", C.c_int32), ("n_cameras", C.c_int32),
                ("huber_mono", C.c_double), ("huber_stereo", C.c_double)]


class LbaSystem(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("Hpp", "bp", "Hll", "bl", "Hpl", "err", "chi2", "rho", "depth", "robust_chi2_sum")]


LBA_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)   # lba_allreduce_fn (include/orbhip.h)


def bind(lib):
    for name in ("lba_build_system", "lba_compute_errors"):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(LbaProblem), C.c_int, C.POINTER(LbaSystem), C.c_void_p]
    lib.lba_build_system_hint.restype = C.c_int
    lib.lba_build_system_hint.argtypes = [C.POINTER(LbaProblem), C.c_int, C.POINTER(LbaSystem), C.c_uint, C.c_void_p]
    lib.lba_lm_workspace_bytes.restype = C.c_size_t
    lib.lba_lm_workspace_bytes.argtypes = [C.POINTER(LbaProblem), C.c_int]
    lib.lba_optimize.restype = C.c_int
    lib.lba_optimize.argtypes = [C.POINTER(LbaProblem), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.lba_optimize_sharded.restype = C.c_int
    lib.lba_optimize_sharded.argtypes = [C.POINTER(LbaProblem), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, LBA_ALLREDUCE_FN, C.c_void_p, C.c_void_p]
    return lib


def build_structure(edges, n_poses, n_points):
    """CSR views of a landmark-major edge array (host side; once per optimize(), like BlockSolver::buildStructure)."""
    assert (np.diff(edges["point"]) >= 0).all(), "edges must be landmark-major (Optimizer.cc:2060-2190 insertion order)"
    lm_start = np.zeros(n_points + 1, np.int32)
    np.add.at(lm_start, edges["point"] + 1, 1)
    lm_start = np.cumsum(lm_start).astype(np.int32)
    order = np.argsort(edges["pose"], kind="stable").astype(np.int32)
    pose_start = np.zeros(n_poses + 1, np.int32)
    np.add.at(pose_start, edges["pose"] + 1, 1)
    pose_start = np.cumsum(pose_start).astype(np.int32)
    return lm_start, pose_start, order


class LbaWindows:
    """A batch of B windows in slab layout, resident wherever `xp` (to-device function) puts them."""

    def __init__(self, windows, cameras, to_dev=lambda a: a, lib=None, huber=(HUBER_MONO, HUBER_STEREO)):
        self._L = bind(lib if lib is not None else _lib.load())
        B = len(windows)
        self.B = B
        self.cap_p = max(len(w["poses"]) for w in windows)
        self.cap_l = max(len(w["points"]) for w in windows)
        self.cap_e = max(len(w["edges"]) for w in windows)
        h = dict(poses=np.zeros((B, self.cap_p, 7)), pose_hidx=np.full((B, self.cap_p), -1, np.int32), points=np.zeros((B, self.cap_l, 3)),
                 edges=np.zeros((B, self.cap_e), EDGE_DTYPE), lm_start=np.zeros((B, self.cap_l + 1), np.int32),
                 pose_start=np.zeros((B, self.cap_p + 1), np.int32), pose_edges=np.zeros((B, self.cap_e), np.int32),
                 n_poses=np.zeros(B, np.int32), n_points=np.zeros(B, np.int32), n_edges=np.zeros(B, np.int32))
        for b, w in enumerate(windows):
            npz, nl, ne = len(w["poses"]), len(w["points"]), len(w["edges"])
            h["poses"][b, :npz] = w["poses"]; h["pose_hidx"][b, :npz] = w["pose_hidx"]; h["points"][b, :nl] = w["points"]
            h["edges"][b, :ne] = w["edges"]
            ls, ps, pe = build_structure(w["edges"], npz, nl)
            h["lm_start"][b, :nl + 1] = ls; h["lm_start"][b, nl + 1:] = ne
            h["pose_start"][b, :npz + 1] = ps; h["pose_start"][b, npz + 1:] = ne
            h["pose_edges"][b, :ne] = pe
            h["n_poses"][b], h["n_points"][b], h["n_edges"][b] = npz, nl, ne
        self.host = h
        self.d = {k: to_dev(v.view(np.uint8).reshape(B, -1) if v.dtype == EDGE_DTYPE else v) for k, v in h.items()}
        self.d["cameras"] = to_dev(np.ascontiguousarray(cameras).view(np.uint8))
        self.n_cameras = len(cameras)
        cam_models = np.asarray(cameras["model"])
        self.mono_pinhole = all((w["edges"]["kind"] == EDGE_MONO).all() and (cam_models[w["edges"]["cam"]] == CAM_PINHOLE).all() for w in windows)
        self.pinhole = all((w["edges"]["kind"] != EDGE_BODY).all() and (cam_models[w["edges"]["cam"]] == CAM_PINHOLE).all() for w in windows)
        self.huber = huber
        like = self.d["poses"]
        f8 = np.float64
        self.out = dict(Hpp=_like64(like, (B, self.cap_p, 36)), bp=_like64(like, (B, self.cap_p, 6)), Hll=_like64(like, (B, self.cap_l, 9)),
                        bl=_like64(like, (B, self.cap_l, 3)), Hpl=_like64(like, (B, self.cap_e, 18)), err=_like64(like, (B, self.cap_e, 3)),
                        chi2=_like64(like, (B, self.cap_e)), rho=_like64(like, (B, self.cap_e, 2)), depth=_like64(like, (B, self.cap_e)),
                        robust_chi2_sum=_like64(like, (B,)))

    def _structs(self, outputs):
        d = self.d
        P = LbaProblem(*[_ptr(d[k]).value for k in ("poses", "pose_hidx", "points", "edges", "lm_start", "pose_start", "pose_edges",
                                                    "cameras", "n_poses", "n_points", "n_edges")],
                       self.cap_p, self.cap_l, self.cap_e, self.n_cameras, self.huber[0], self.huber[1])
        S = LbaSystem(*[(_ptr(self.out[k]).value if k in outputs else None) for k in
                        ("Hpp", "bp", "Hll", "bl", "Hpl", "err", "chi2", "rho", "depth", "robust_chi2_sum")])
        return P, S

    def build_system(self, outputs=("Hpp", "bp", "Hll", "bl", "Hpl", "err", "chi2", "rho", "depth")):
        P, S = self._structs(outputs)
        # LBA_HINT_MONO_PINHOLE / LBA_HINT_PINHOLE when the host-side edge arrays say so (the caller flattened the graph: it knows the edge kinds)
        rc = self._L.lba_build_system_hint(C.byref(P), self.B, C.byref(S), 1 if self.mono_pinhole else 2 if self.pinhole else 0, _stream(self.d["poses"]))
        if rc != 0:
            raise OrbHipError(rc, "lba_build_system failed")
        return self.out

    def optimize(self, iterations):
        """optimizer.optimize(iterations) (Optimizer.cc:2205 / :2290): LM with Schur complement on the device; poses / points in self.d are
        updated in place.  -> stats [B,4] = iterations run, final robust chi2, final lambda, lambda trials."""
        P, _ = self._structs(())
        if getattr(self, "_lm_ws", None) is None:
            n = self._L.lba_lm_workspace_bytes(C.byref(P), self.B)
            self._lm_ws = _like(self.d["poses"], (n,), np.uint8)
        stats = np.zeros((self.B, 4), np.float64)
        rc = self._L.lba_optimize(C.byref(P), self.B, int(iterations), _ptr(self._lm_ws), stats.ctypes.data_as(C.c_void_p), None,
                                  _stream(self.d["poses"]))
        if rc != 0:
            raise OrbHipError(rc, "lba_optimize failed")
        return stats

    def optimize_sharded(self, iterations, group=None, owner=None):
        """optimize() for windows sharded by LANDMARK over the ranks of `group` (orbhip.dist.shard_window_by_landmark; include/orbhip.h
        lba_optimize_sharded): every rank holds all poses and the points / edges of its landmarks.  The library's reductions (pose-side blocks per
        linearisation; reduced camera system + three scalars per lambda trial) run as torch.distributed all-reduces on views of the workspace —
        RCCL on MI355X, gloo in the CPU tests.  -> stats as optimize(), identical on every rank; self.d["poses"] holds ALL updated poses,
        self.d["points"] this rank's landmarks.  self.reduce_stats: calls / doubles moved by the last call."""
        import torch
        import torch.distributed as dist
        P, _ = self._structs(())
        if getattr(self, "_lm_ws", None) is None:
            n = self._L.lba_lm_workspace_bytes(C.byref(P), self.B)
            self._lm_ws = _like(self.d["poses"], (n,), np.uint8)
        ws = self._lm_ws
        is_np = isinstance(ws, np.ndarray)
        base = ws.ctypes.data if is_np else ws.data_ptr()
        owner = (dist.get_rank(group) == 0) if owner is None else bool(owner)
        self.reduce_stats = dict(calls=0, doubles=0, error=None)

        def _reduce(user, buf, n, op, stream):
            try:
                off = int(buf) - base
                assert 0 <= off and off + 8 * n <= ws.shape[0] and off % 8 == 0
                t = torch.from_numpy(ws[off:off + 8 * n].view(np.float64)) if is_np else ws[off:off + 8 * n].view(torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM, group=group)   # (stream-ordered on RCCL: the kernels run on torch's current stream)
                self.reduce_stats["calls"] += 1
                self.reduce_stats["doubles"] += int(n)
                return 0
            except Exception as err:   # noqa: BLE001  (no exception may cross the C ABI)
                self.reduce_stats["error"] = "%s: %s" % (type(err).__name__, err)
                return 1
        cb = LBA_ALLREDUCE_FN(_reduce)
        stats = np.zeros((self.B, 4), np.float64)
        rc = self._L.lba_optimize_sharded(C.byref(P), self.B, int(iterations), _ptr(ws), stats.ctypes.data_as(C.c_void_p), 1 if owner else 0, cb, None,
                                          _stream(self.d["poses"]))
        if rc != 0:
            raise OrbHipError(rc, "lba_optimize_sharded failed" + (": " + self.reduce_stats["error"] if self.reduce_stats["error"] else ""))
        return stats

    def compute_errors(self, outputs=("err", "chi2", "rho", "depth", "robust_chi2_sum")):
        P, S = self._structs(outputs)
        rc = self._L.lba_compute_errors(C.byref(P), self.B, C.byref(S), _stream(self.d["poses"]))
        if rc != 0:
            raise OrbHipError(rc, "lba_compute_errors failed")
        return self.out


def _like64(a, shape):
    if isinstance(a, np.ndarray):
        return np.zeros(shape, np.float64)
    import torch
    return torch.zeros(shape, dtype=torch.float64, device=a.device)


# ---- synthetic windows: KFs on a circle looking inward, points in a box ------------------
def rot_to_quat(R):
    """Eigen Quaterniond(Matrix3d) + SE3Quat::normalizeRotation, as Converter::toSE3Quat -> SE3Quat(R,t) does."""
    m = R
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        q = np.array([(m[2, 1] - m[1, 2]) * t, (m[0, 2] - m[2, 0]) * t, (m[1, 0] - m[0, 1]) * t, w])
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        c = np.zeros(4)
        c[i] = 0.5 * t
        t = 0.5 / t
        c[3] = (m[k, j] - m[j, k]) * t
        c[j] = (m[j, i] + m[i, j]) * t
        c[k] = (m[k, i] + m[i, k]) * t
        q = c
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def _kb8_project(p, X):
    th = np.arctan2(np.hypot(X[..., 0], X[..., 1]), X[..., 2]); psi = np.arctan2(X[..., 1], X[..., 0])
    r = th + p[4] * th**3 + p[5] * th**5 + p[6] * th**7 + p[7] * th**9
    return p[0] * r * np.cos(psi) + p[2], p[1] * r * np.sin(psi) + p[3], th


def synth_window(seed=0, n_kf=100, n_fixed=20, n_pts=20000, max_obs=8, kind="mono", fx=458.654, fy=457.296, cx=367.215, cy=248.375,
                 W=752, H=480, bf=47.906, outliers=0.05):
    """-> (window dict, cameras array).  EuRoC pinhole intrinsics (Examples/Monocular/EuRoC.yaml:9-12); kind in
    {"mono","stereo","kb8","body","mixed"}.  Vectorised numpy; deterministic per seed."""
    rng = np.random.default_rng(seed)
    cams = np.zeros(2, CAM_DTYPE)
    cams[0]["model"] = CAM_PINHOLE
    cams[0]["p"][:4] = np.float32([fx, fy, cx, cy]); cams[0]["bf"] = np.float32(bf)
    cams[0]["trl_q"] = [0, 0, 0, 1]
    cams[1]["model"] = CAM_KB8   # TUM_512.yaml:8-18-like fisheye, also the "right camera" of body edges
    cams[1]["p"] = np.float32([190.978, 190.973, 254.932, 256.897, 0.0034823894, 0.0007150348, -0.0020532361, 0.00020293673])
    Rrl = _rodrigues(np.array([0.002, -0.01, 0.003]))
    cams[1]["trl_q"] = rot_to_quat(Rrl); cams[1]["trl_t"] = [-0.1, 0.001, 0.0005]
    trl_t = np.array(cams[1]["trl_t"])
    pk = np.array(cams[1]["p"])
    poses = np.zeros((n_kf, 7)); Rs = np.zeros((n_kf, 3, 3)); ts = np.zeros((n_kf, 3))
    for i in range(n_kf):
        a = 2 * np.pi * i / n_kf
        c = np.array([10 * np.cos(a), 10 * np.sin(a), rng.normal(0, 0.2)])
        z = -c / np.linalg.norm(c); z = z + rng.normal(0, 0.05, 3); z /= np.linalg.norm(z)
        x = np.cross(np.array([0, 0, 1.0]), z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        Rs[i] = np.stack([x, y, z]).astype(np.float32).astype(np.float64)   # map poses are float32 (Converter::toSE3Quat widens)
        ts[i] = (-Rs[i] @ c).astype(np.float32).astype(np.float64)
        poses[i, :3] = ts[i]; poses[i, 3:] = rot_to_quat(Rs[i])
    pts = np.stack([rng.uniform(-5, 5, n_pts), rng.uniform(-5, 5, n_pts), rng.uniform(-2, 2, n_pts)], 1).astype(np.float32).astype(np.float64)
    scale2 = (np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2
    Xc = np.einsum("kij,lj->kli", Rs, pts) + ts[:, None, :]          # (K, L, 3)
    kidx = np.arange(n_kf)[:, None]
    lidx = np.arange(n_pts)[None, :]
    if kind in ("kb8", "body"):
        use_kb8 = np.ones((n_kf, n_pts), bool)
    elif kind == "mixed":
        use_kb8 = np.broadcast_to(kidx % 3 == 0, (n_kf, n_pts))
    else:
        use_kb8 = np.zeros((n_kf, n_pts), bool)
    with np.errstate(divide="ignore", invalid="ignore"):
        up, vp = fx * Xc[..., 0] / Xc[..., 2] + cx, fy * Xc[..., 1] / Xc[..., 2] + cy
    uk, vk, thk = _kb8_project(pk, Xc)
    vis_p = (up >= 0) & (up < W) & (vp >= 0) & (vp < H)
    vis_k = (uk >= 0) & (uk < 512) & (vk >= 0) & (vk < 512) & (thk < 1.4)
    vis = (Xc[..., 2] >= 0.5) & np.where(use_kb8, vis_k, vis_p)
    key = rng.random((n_kf, n_pts)); key[~vis] = 2.0
    order = np.argsort(key, axis=0)[:max_obs]                        # (max_obs, L) random visible KFs first
    sel = np.take_along_axis(key, order, 0) < 1.5
    pi = order.T[sel.T]                                              # landmark-major edge list
    li = np.broadcast_to(lidx.T, (n_pts, max_obs))[sel.T] if max_obs else np.zeros(0, int)
    ne = len(pi)
    X = Xc[pi, li]
    kb = use_kb8[pi, li]
    octave = rng.integers(0, 8, ne)
    sig = np.sqrt(scale2[octave]).astype(np.float64)
    noise = rng.normal(0, 1, (ne, 3)) * sig[:, None]
    out = rng.random(ne) < outliers
    noise[out, :2] = rng.normal(0, 25, (int(out.sum()), 2))
    edges = np.zeros(ne, EDGE_DTYPE)
    edges["pose"], edges["point"] = pi, li
    edges["inv_sigma2"] = np.float32(1.0) / scale2[octave]
    body = kb & ((kind == "body") | ((kind == "mixed") & (pi % 2 == 1)))
    stereo = ~kb & ((kind == "stereo") | ((kind == "mixed") & (li % 2 == 0)))
    edges["kind"] = np.where(body, EDGE_BODY, np.where(stereo, EDGE_STEREO, EDGE_MONO))
    edges["cam"] = np.where(kb, 1, 0)
    u = np.where(kb, uk[pi, li], up[pi, li]); v = np.where(kb, vk[pi, li], vp[pi, li])
    if body.any():
        Xr = X[body] @ Rrl.T + trl_t
        ub, vb, _ = _kb8_project(pk, Xr)
        u[body], v[body] = ub, vb
    obs = np.zeros((ne, 3))
    obs[:, 0], obs[:, 1] = u + noise[:, 0], v + noise[:, 1]
    obs[stereo, 2] = (u + noise[:, 0] - bf / X[:, 2] + noise[:, 2])[stereo]
    edges["obs"] = obs.astype(np.float32)
    # perturb the estimates so that the linearisation point is not the optimum
    pts_est = pts + rng.normal(0, 0.02, pts.shape)
    hidx = np.full(n_kf, -1, np.int32); hidx[n_fixed:] = np.arange(n_kf - n_fixed)
    for i in range(n_fixed, n_kf):
        dR = _rodrigues(rng.normal(0, 0.002, 3))
        poses[i, 3:] = rot_to_quat(dR @ Rs[i]); poses[i, :3] = dR @ ts[i] + rng.normal(0, 0.01, 3)
    return dict(poses=poses, pose_hidx=hidx, points=pts_est, edges=edges), cams


def _rodrigues(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * (K @ K)


# ---- Optimizer::PoseOptimization (Optimizer.cc:907-1273) ---------------------------------------------------------
POSE_EDGE_DTYPE = np.dtype([("xw", "<f4", (3,)), ("obs", "<f4", (3,)), ("inv_sigma2", "<f4"), ("kind", "<i2"), ("cam", "<i2")])
assert POSE_EDGE_DTYPE.itemsize == 32


def bind_pose(lib):
    lib.pose_optimize.restype = C.c_int
    lib.pose_optimize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]
    lib.pose_optimize_hint.restype = C.c_int
    lib.pose_optimize_hint.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_uint, C.c_void_p]
    return lib


HINT_MONO_PINHOLE, HINT_PINHOLE = 1, 2


def pose_optimization(poses, edges, n_edges, cameras, lib=None, pinhole=False):
    """Batched Optimizer::PoseOptimization: poses [B,7] f64 (t, q of Tcw), edges [B,cap_e] POSE_EDGE_DTYPE viewed as u8 [B,cap_e*32],
    n_edges [B] i32, cameras u8 view of CAM_DTYPE[n].  All arrays device tensors (or numpy in the emulated build).
    pinhole=True: the caller states that every edge is EDGE_MONO / EDGE_STEREO on a pinhole camera (LBA_HINT_PINHOLE: same results, leaner kernel).
    -> (poses_out [B,7], outlier [B,cap_e] u8, n_good [B] i32) — n_good == the reference's return value nInitialCorrespondences-nBad."""
    L = bind_pose(lib if lib is not None else _lib.load())
    B = poses.shape[0]
    cap_e = edges.shape[1] // POSE_EDGE_DTYPE.itemsize if edges.dtype != POSE_EDGE_DTYPE else edges.shape[1]
    out = _like64(poses, (B, 7))
    outlier = _like(poses, (B, cap_e), np.uint8)
    n_good = _like(poses, (B,), np.int32)
    n_cam = cameras.shape[0] // CAM_DTYPE.itemsize if cameras.dtype != CAM_DTYPE else cameras.shape[0]
    rc = L.pose_optimize_hint(_ptr(poses), _ptr(edges), _ptr(n_edges), cap_e, B, _ptr(cameras), n_cam, _ptr(out), _ptr(outlier), _ptr(n_good),
                              HINT_PINHOLE if pinhole else 0, _stream(poses))
    if rc != 0:
        raise OrbHipError(rc, "pose_optimize failed")
    return out, outlier, n_good


def synth_pose_frames(seed=0, batch=4, n_pts=300, kind="mono", outlier_frac=0.1, pose_noise=(0.02, 0.05), fx=458.654, fy=457.296, cx=367.215,
                      cy=248.375, bf=47.9, width=752, height=480):
    """Synthetic tracking frames: map points in front of a camera, observations with octave-dependent sigma, gross outliers, and an
    initial pose perturbed off the truth.  kind: mono | stereo (mix of mono and stereo edges, RGB-D/stereo) | body (fisheye rig, KB8)."""
    rng = np.random.default_rng(seed)
    cams = np.zeros(2, CAM_DTYPE)
    if kind == "body":
        for c in cams:
            c["model"] = CAM_KB8; c["p"][:8] = [190.978, 190.973, 254.93, 256.9, 0.0034, 0.0007, -0.0020, 0.0002]
        cams[1]["trl_q"] = rot_to_quat(_rodrigues(np.array([0.0, 0.02, 0.0]))); cams[1]["trl_t"] = [-0.1, 0.001, 0.0005]
        cams[0]["trl_q"] = [0, 0, 0, 1]
        width = height = 512
    else:
        for c in cams:
            c["model"] = CAM_PINHOLE; c["p"][:4] = [fx, fy, cx, cy]; c["bf"] = bf; c["trl_q"] = [0, 0, 0, 1]
    poses0 = np.zeros((batch, 7)); edges = np.zeros((batch, n_pts), POSE_EDGE_DTYPE); n_edges = np.zeros(batch, np.int32)
    sig2 = 1.2 ** (2 * np.arange(8))
    for b in range(batch):
        R = _rodrigues(rng.normal(0, 0.3, 3)); t = rng.normal(0, 1.0, 3)
        n = n_pts if b % 4 != 3 else max(3, n_pts // (b + 2))
        if kind == "body":
            u = rng.uniform(60, width - 60, n); v = rng.uniform(60, height - 60, n); z = rng.uniform(1.0, 12.0, n)
            p = cams[0]["p"]; mx, my = (u - p[2]) / p[0], (v - p[3]) / p[1]
            th = np.hypot(mx, my); s = np.where(th > 1e-9, np.tan(np.minimum(th, 1.3)) / np.maximum(th, 1e-9), 1.0)
            Xc = np.stack([mx * s * z, my * s * z, z], 1)
        else:
            u = rng.uniform(20, width - 20, n); v = rng.uniform(20, height - 20, n); z = rng.uniform(0.8, 15.0, n)
            Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
        Xw = (Xc - t) @ R   # R^T (Xc - t)
        e = edges[b]
        e["xw"][:n] = Xw.astype(np.float32)
        Xw32 = e["xw"][:n].astype(np.float64)
        Xc = Xw32 @ R.T + t
        octave = rng.integers(0, 8, n)
        e["inv_sigma2"][:n] = (1.0 / sig2[octave]).astype(np.float32)
        noise = rng.normal(0, 1, (n, 2)) * np.sqrt(sig2[octave])[:, None] * 0.7
        if kind == "body":
            right = rng.random(n) < 0.4
            e["kind"][:n] = np.where(right, EDGE_BODY, EDGE_MONO); e["cam"][:n] = np.where(right, 1, 0)
            Rrl = _quat_to_rot(cams[1]["trl_q"]); trl = cams[1]["trl_t"]
            Xe = np.where(right[:, None], Xc @ Rrl.T + trl, Xc)
            p = cams[0]["p"]
            r = np.hypot(Xe[:, 0], Xe[:, 1]); th = np.arctan2(r, Xe[:, 2])
            thd = th * (1 + p[4] * th**2 + p[5] * th**4 + p[6] * th**6 + p[7] * th**8)
            sc = np.where(r > 1e-12, thd / np.maximum(r, 1e-12), 1.0)
            uv = np.stack([p[0] * sc * Xe[:, 0] + p[2], p[1] * sc * Xe[:, 1] + p[3]], 1)
            e["obs"][:n, :2] = (uv + noise).astype(np.float32)
        else:
            uv = np.stack([fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy], 1)
            e["obs"][:n, :2] = (uv + noise).astype(np.float32)
            if kind == "stereo":
                st = rng.random(n) < 0.6
                e["kind"][:n] = np.where(st, EDGE_STEREO, EDGE_MONO)
                ur = uv[:, 0] - bf / Xc[:, 2] + noise[:, 0] * 0.5
                e["obs"][:n, 2] = np.where(st, ur, 0).astype(np.float32)
        bad = rng.random(n) < outlier_frac
        e["obs"][:n, :2] += (bad[:, None] * rng.normal(0, 40, (n, 2))).astype(np.float32)
        Rn = _rodrigues(rng.normal(0, pose_noise[0], 3)) @ R; tn = t + rng.normal(0, pose_noise[1], 3)
        poses0[b, :3] = tn; poses0[b, 3:] = rot_to_quat(Rn)
        n_edges[b] = n
    return dict(poses=poses0, edges=edges, n_edges=n_edges, cameras=cams)


def _quat_to_rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
