"""Host-side mirror of Optimizer::LocalInertialBA's optimisation (reference src/Optimizer.cc:4753-5365; vertices / edges of
include/G2oTypes.h) above the C ABI (include/orbhip.h, "liba_*"): visual-inertial local bundle adjustment windows, batched.

A window = key frames (ImuCamPose + velocity + gyro / acc bias, `KF_DTYPE`), the rig calibration (`Rig`), map points, visual edges
(`EDGE_DTYPE` of orbhip.lba: kind 0 = EdgeMono, 1 = EdgeStereo, cam = camera index) and one `IMU_EDGE_DTYPE` record per preintegration
(EdgeInertial + EdgeGyroRW + EdgeAccRW).  Arrays are torch CUDA tensors (product) or numpy (emulated test build)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import OrbHipError
from .lba import EDGE_DTYPE, EDGE_MONO, EDGE_STEREO, HUBER_MONO, HUBER_STEREO, _kb8_project, _rodrigues

KF_DTYPE = np.dtype([("Rwb", "<f8", (9,)), ("twb", "<f8", (3,)), ("Rcw", "<f8", (2, 9)), ("tcw", "<f8", (2, 3)), ("v", "<f8", (3,)),
                     ("bg", "<f8", (3,)), ("ba", "<f8", (3,)), ("pose_fixed", "<i4"), ("has_imu", "<i4"), ("imu_fixed", "<i4"), ("reserved", "<i4")])
IMU_EDGE_DTYPE = np.dtype([("kf1", "<i4"), ("kf2", "<i4"), ("dR", "<f4", (9,)), ("dV", "<f4", (3,)), ("dP", "<f4", (3,)), ("JRg", "<f4", (9,)),
                           ("JVg", "<f4", (9,)), ("JVa", "<f4", (9,)), ("JPg", "<f4", (9,)), ("JPa", "<f4", (9,)), ("b", "<f4", (6,)), ("dT", "<f4"),
                           ("pad", "<f4"), ("huber", "<f8"), ("info", "<f8", (81,)), ("info_g", "<f8", (9,)), ("info_a", "<f8", (9,))])
assert KF_DTYPE.itemsize == 376 and IMU_EDGE_DTYPE.itemsize == 1080
HUBER_INERTIAL = float(np.sqrt(16.92))   # rki->setDelta(sqrt(16.92))  Optimizer.cc:5012
GRAVITY_VALUE = 9.81                     # ImuTypes.h:40


class Rig(C.Structure):
    """Calibration members of ImuCamPose (G2oTypes.h:60-72): per camera Rcb, tcb, Rbc, tbc; bf; camera model + parameters."""
    _fields_ = [("n_cams", C.c_int32), ("reserved", C.c_int32), ("Rcb", (C.c_double * 9) * 2), ("tcb", (C.c_double * 3) * 2),
                ("Rbc", (C.c_double * 9) * 2), ("tbc", (C.c_double * 3) * 2), ("bf", C.c_double), ("model", C.c_int32 * 2), ("p", (C.c_double * 8) * 2)]


def make_rig(Tcb_list, bf, models, params):
    r = Rig()
    r.n_cams = len(Tcb_list)
    r.bf = float(bf)
    for c, (Rcb, tcb) in enumerate(Tcb_list):
        Rbc = Rcb.T
        tbc = -Rbc @ tcb
        for i in range(9):
            r.Rcb[c][i] = float(Rcb.reshape(-1)[i]); r.Rbc[c][i] = float(Rbc.reshape(-1)[i])
        for i in range(3):
            r.tcb[c][i] = float(tcb[i]); r.tbc[c][i] = float(tbc[i])
        r.model[c] = int(models[c])
        for i in range(8):
            r.p[c][i] = float(params[c][i]) if i < len(params[c]) else 0.0
    return r


def set_cam_poses(kf, rig):
    """ImuCamPose constructor / Update tail: Rcw[i] = Rcb[i] * Rbw, tcw[i] = Rcb[i] * tbw + tcb[i] (G2oTypes.cc:213-220)."""
    Rwb = kf["Rwb"].reshape(3, 3)
    Rbw = Rwb.T
    tbw = -Rbw @ kf["twb"]
    for c in range(rig.n_cams):
        Rcb = np.array(rig.Rcb[c]).reshape(3, 3)
        kf["Rcw"][c] = (Rcb @ Rbw).reshape(-1)
        kf["tcw"][c] = Rcb @ tbw + np.array(rig.tcb[c])


def synth_inertial_window(seed=0, n_opt=8, n_fixed_vis=3, n_pts=500, max_obs=6, kind="mono", outliers=0.03, fx=458.654, fy=457.296, cx=367.215,
                          cy=248.375, W=752, H=480, bf=47.906, dt=0.3):
    """-> dict(kfs, rig, points, edges, imu).  kind: "mono" / "stereo" (EuRoC pinhole, one camera) or "fisheye" (two KB8 cameras, TUM-VI like).
    Key frames in ascending id order: n_fixed_vis vision-only fixed KFs, the fixed IMU KF just before the temporal window, then n_opt optimisable
    ones.  Preintegrations are the exact relative motion of the true trajectory plus noise; estimates start perturbed."""
    rng = np.random.default_rng(seed)
    g = np.array([0.0, 0.0, -GRAVITY_VALUE])
    n_imu_kf = n_opt + 1
    n_kf = n_fixed_vis + n_imu_kf
    # rig
    Rcb0 = _rodrigues(np.array([0.01, -0.02, 0.015])); tcb0 = np.array([0.05, -0.02, 0.01])
    if kind == "fisheye":
        pk = np.float32([190.978, 190.973, 254.932, 256.897, 0.0034823894, 0.0007150348, -0.0020532361, 0.00020293673]).astype(np.float64)
        Rrl = _rodrigues(np.array([0.002, -0.01, 0.003])); trl = np.array([-0.1, 0.001, 0.0005])
        rig = make_rig([(Rcb0, tcb0), (Rrl @ Rcb0, Rrl @ tcb0 + trl)], 0.0, [1, 1], [pk, pk * np.array([1.001, 0.999, 1.002, 0.998, 1, 1, 1, 1])])
    else:
        rig = make_rig([(Rcb0, tcb0)], np.float32(bf), [0], [np.float32([fx, fy, cx, cy]).astype(np.float64)])
    # true trajectory: an arc around the point cloud, cameras looking inwards; body = camera up to Tcb
    Rwb_t = np.zeros((n_kf, 3, 3)); twb_t = np.zeros((n_kf, 3)); v_t = np.zeros((n_kf, 3))

    def body_pose(t):      # t in seconds
        a = 0.4 * t
        c = np.array([9 * np.cos(a), 9 * np.sin(a), 0.3 * np.sin(2.0 * t)])
        z = -c / np.linalg.norm(c)
        x = np.cross(np.array([0, 0, 1.0]), z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        Rwc = np.stack([x, y, z], 1)             # camera axes in world
        Rwb = Rwc @ Rcb0                         # Rwc = Rwb * Rbc  ->  Rwb = Rwc * Rcb
        twb = c - Rwb @ (-Rcb0.T @ tcb0)         # twc = twb + Rwb * tbc
        return Rwb, twb
    times = np.concatenate([-(np.arange(n_fixed_vis, 0, -1) * 0.9 + 0.6), np.arange(n_imu_kf) * dt])
    for k in range(n_kf):
        Rwb_t[k], twb_t[k] = body_pose(times[k])
        e = 1e-5
        v_t[k] = (body_pose(times[k] + e)[1] - body_pose(times[k] - e)[1]) / (2 * e)
    # key-frame records (estimates = truth + perturbation for the optimisable ones); map values are float32 widened
    kfs = np.zeros(n_kf, KF_DTYPE)
    b0 = np.float32(rng.normal(0, [0.002] * 3 + [0.01] * 3))          # preintegration bias (bax bay baz bwx bwy bwz)
    first_opt = n_fixed_vis + 1
    for k in range(n_kf):
        R, t, v = Rwb_t[k].copy(), twb_t[k].copy(), v_t[k].copy()
        if k >= first_opt:
            R = R @ _rodrigues(rng.normal(0, 0.004, 3)); t = t + rng.normal(0, 0.015, 3); v = v + rng.normal(0, 0.03, 3)
        kfs[k]["Rwb"] = R.astype(np.float32).astype(np.float64).reshape(-1)
        kfs[k]["twb"] = t.astype(np.float32)
        kfs[k]["v"] = v.astype(np.float32)
        kfs[k]["ba"] = (b0[:3] + np.float32(rng.normal(0, 0.003, 3))).astype(np.float32)
        kfs[k]["bg"] = (b0[3:] + np.float32(rng.normal(0, 0.0005, 3))).astype(np.float32)
        kfs[k]["pose_fixed"] = 0 if k >= first_opt else 1
        kfs[k]["has_imu"] = 1 if k >= n_fixed_vis else 0
        kfs[k]["imu_fixed"] = 0 if k >= first_opt else 1
        set_cam_poses(kfs[k], rig)
    # preintegrations, newest first like the loop of Optimizer.cc:4964-5062
    imu = np.zeros(n_opt, IMU_EDGE_DTYPE)
    for i in range(n_opt):
        k2 = n_kf - 1 - i
        k1 = k2 - 1
        T = times[k2] - times[k1]
        R1 = Rwb_t[k1]
        E = imu[i]
        E["kf1"], E["kf2"] = k1, k2
        E["dT"] = np.float32(T)
        Td = float(np.float32(T))
        E["dR"] = (R1.T @ Rwb_t[k2] @ _rodrigues(rng.normal(0, 5e-4, 3))).reshape(-1)
        vv1, vv2 = v_t[k1], v_t[k2]
        E["dV"] = R1.T @ (vv2 - vv1 - g * Td) + rng.normal(0, 2e-3, 3)
        E["dP"] = R1.T @ (twb_t[k2] - twb_t[k1] - vv1 * Td - 0.5 * g * Td * Td) + rng.normal(0, 2e-3, 3)
        E["JRg"] = (-Td * np.eye(3) + rng.normal(0, 0.01, (3, 3))).reshape(-1)
        E["JVg"] = rng.normal(0, 0.05, 9); E["JVa"] = (-Td * R1.T @ R1 + rng.normal(0, 0.01, (3, 3))).reshape(-1)
        E["JPg"] = rng.normal(0, 0.01, 9); E["JPa"] = (-0.5 * Td * Td * np.eye(3) + rng.normal(0, 0.003, (3, 3))).reshape(-1)
        E["b"] = b0
        L = np.diag(np.sqrt([3e4] * 3 + [2e3] * 3 + [8e3] * 3)) @ (np.eye(9) + 0.05 * rng.normal(0, 1, (9, 9)))
        info = (L @ L.T).astype(np.float32).astype(np.float64)
        info = (info + info.T) / 2
        if i == n_opt - 1:      # oldest edge: Huber + information * 1e-2 (Optimizer.cc:5005-5012)
            info = info * 1e-2
            E["huber"] = HUBER_INERTIAL
        E["info"] = info.reshape(-1)
        E["info_g"] = (np.eye(3) * 4e5 + 1e3 * rng.normal(0, 1, (3, 3))).astype(np.float32).reshape(-1)
        E["info_a"] = (np.eye(3) * 2e3 + 10 * rng.normal(0, 1, (3, 3))).astype(np.float32).reshape(-1)
    # map points and observations from the TRUE poses
    pts = np.stack([rng.uniform(-4, 4, n_pts), rng.uniform(-4, 4, n_pts), rng.uniform(-1.5, 1.5, n_pts)], 1).astype(np.float32).astype(np.float64)
    scale2 = (np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2
    edges = []
    true_kf = np.zeros(n_kf, KF_DTYPE)
    for k in range(n_kf):
        true_kf[k]["Rwb"] = Rwb_t[k].reshape(-1); true_kf[k]["twb"] = twb_t[k]
        set_cam_poses(true_kf[k], rig)
    for l in range(n_pts):
        ks = rng.permutation(n_kf)
        nob = 0
        for k in ks:
            if nob >= max_obs:
                break
            got = False
            for c in range(rig.n_cams):
                Xc = true_kf[k]["Rcw"][c].reshape(3, 3) @ pts[l] + true_kf[k]["tcw"][c]
                if Xc[2] < 0.5:
                    continue
                if rig.model[c] == 0:
                    u, v = fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy
                    ok = 0 <= u < W and 0 <= v < H
                else:
                    uu, vv, th = _kb8_project(np.array(rig.p[c]), Xc[None, :])
                    u, v = float(uu[0]), float(vv[0])
                    ok = 0 <= u < 512 and 0 <= v < 512 and th[0] < 1.3
                if not ok:
                    continue
                o = int(rng.integers(0, 8))
                sg = float(np.sqrt(scale2[o]))
                n2 = rng.normal(0, 1, 3) * sg
                if rng.random() < outliers:
                    n2[:2] = rng.normal(0, 25, 2)
                stereo = kind == "stereo" and (l % 3 != 0)
                obs = (u + n2[0], v + n2[1], (u + n2[0] - bf / Xc[2] + n2[2]) if stereo else 0.0)
                edges.append((k, l, EDGE_STEREO if stereo else EDGE_MONO, c, obs, np.float32(1.0) / scale2[o]))
                got = True
            nob += got
    ea = np.zeros(len(edges), EDGE_DTYPE)
    for i, (k, l, kd, c, obs, s) in enumerate(edges):
        ea[i]["pose"], ea[i]["point"], ea[i]["kind"], ea[i]["cam"], ea[i]["obs"], ea[i]["inv_sigma2"] = k, l, kd, c, np.float32(obs), s
    ea = ea[np.argsort(ea["point"], kind="stable")]
    return {"kfs": kfs, "rig": rig, "points": pts + rng.normal(0, 0.03, pts.shape), "edges": ea, "imu": imu}


# ---- device path ------------------------------------------------------------------------------------------------------------------------
class LibaProblem(C.Structure):
    _fields_ = [("kfs", C.c_void_p), ("n_kf", C.c_void_p), ("rigs", C.c_void_p), ("points", C.c_void_p), ("n_points", C.c_void_p),
                ("edges", C.c_void_p), ("n_edges", C.c_void_p), ("imu", C.c_void_p), ("n_imu", C.c_void_p),
                ("cap_kf", C.c_int32), ("cap_l", C.c_int32), ("cap_e", C.c_int32), ("cap_i", C.c_int32), ("rig_stride", C.c_int32),
                ("max_free", C.c_int32), ("huber_mono", C.c_double), ("huber_stereo", C.c_double)]


def bind(lib):
    vp, i32, sz, f64 = C.c_void_p, C.c_int, C.c_size_t, C.c_double
    protos = {
        "liba_workspace_bytes": (sz, [C.POINTER(LibaProblem), i32]),
        "liba_optimize": (i32, [C.POINTER(LibaProblem), i32, f64, i32, vp, vp, vp]),
        "liba_compute_errors": (i32, [C.POINTER(LibaProblem), i32, vp, vp, vp, vp, vp]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)
    assert a.is_contiguous()
    return C.c_void_p(a.data_ptr())


class InertialWindows:
    """A batch of LocalInertialBA windows resident on the device.  `to_dev` maps a numpy array to device memory (torch.from_numpy(a).cuda()
    for the product; identity for the emulated build).  Structured arrays travel as uint8 views."""

    def __init__(self, windows, to_dev, *, lib=None, huber=(HUBER_MONO, HUBER_STEREO)):
        self._L = bind(lib if lib is not None else _lib.load())
        B = len(windows)
        self.B = B
        self.cap_kf = max(len(w["kfs"]) for w in windows)
        self.cap_l = max(len(w["points"]) for w in windows)
        self.cap_e = max(len(w["edges"]) for w in windows)
        self.cap_i = max(1, max(len(w["imu"]) for w in windows))
        self.max_free = max(int((w["kfs"]["pose_fixed"] == 0).sum()) for w in windows)
        kfs = np.zeros((B, self.cap_kf), KF_DTYPE); pts = np.zeros((B, self.cap_l, 3)); edges = np.zeros((B, self.cap_e), EDGE_DTYPE)
        imu = np.zeros((B, self.cap_i), IMU_EDGE_DTYPE)
        n = np.zeros((4, B), np.int32)
        rigs = (Rig * B)()
        for b, w in enumerate(windows):
            n[:, b] = len(w["kfs"]), len(w["points"]), len(w["edges"]), len(w["imu"])
            kfs[b, :n[0, b]] = w["kfs"]; pts[b, :n[1, b]] = w["points"]; edges[b, :n[2, b]] = w["edges"]; imu[b, :n[3, b]] = w["imu"]
            rigs[b] = w["rig"]
        self.n = n
        rig_bytes = np.frombuffer(bytes(rigs), np.uint8).copy()
        self.d = {"kfs": to_dev(kfs.view(np.uint8).reshape(B, -1)), "points": to_dev(pts), "edges": to_dev(edges.view(np.uint8).reshape(B, -1)),
                  "imu": to_dev(imu.view(np.uint8).reshape(B, -1)), "rigs": to_dev(rig_bytes),
                  "n_kf": to_dev(n[0].copy()), "n_points": to_dev(n[1].copy()), "n_edges": to_dev(n[2].copy()), "n_imu": to_dev(n[3].copy())}
        self._to_dev = to_dev
        d = self.d
        self.prob = LibaProblem(_ptr(d["kfs"]), _ptr(d["n_kf"]), _ptr(d["rigs"]), _ptr(d["points"]), _ptr(d["n_points"]), _ptr(d["edges"]), _ptr(d["n_edges"]),
                                _ptr(d["imu"]), _ptr(d["n_imu"]), self.cap_kf, self.cap_l, self.cap_e, self.cap_i, 1, self.max_free, huber[0], huber[1])
        self._work = None

    def _stream(self):
        a = self.d["points"]
        if isinstance(a, np.ndarray):
            return None
        import torch
        return C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)

    def optimize(self, lambda_init, iterations):
        """optimizer.optimize(iterations); key frames / points updated in place on the device.  -> stats [B, 5] (device array)."""
        if self._work is None:
            nbytes = self._L.liba_workspace_bytes(C.byref(self.prob), self.B)
            self._work = self._to_dev(np.zeros(nbytes, np.uint8))
        stats = self._to_dev(np.zeros((self.B, 5)))
        rc = self._L.liba_optimize(C.byref(self.prob), self.B, float(lambda_init), int(iterations), _ptr(self._work), _ptr(stats), self._stream())
        if rc != 0:
            raise OrbHipError(rc, "liba_optimize failed")
        return stats

    def compute_errors(self):
        vchi = self._to_dev(np.zeros((self.B, self.cap_e))); vdp = self._to_dev(np.zeros((self.B, self.cap_e), np.uint8))
        ichi = self._to_dev(np.zeros((self.B, self.cap_i, 3))); rs = self._to_dev(np.zeros(self.B))
        rc = self._L.liba_compute_errors(C.byref(self.prob), self.B, _ptr(vchi), _ptr(vdp), _ptr(ichi), _ptr(rs), self._stream())
        if rc != 0:
            raise OrbHipError(rc, "liba_compute_errors failed")
        return {"vis_chi2": vchi, "vis_depth_pos": vdp, "imu_chi2": ichi, "robust_chi2_sum": rs}

    def keyframes(self):
        a = self.d["kfs"]
        a = a if isinstance(a, np.ndarray) else a.cpu().numpy()
        return a.reshape(self.B, -1).view(KF_DTYPE).reshape(self.B, self.cap_kf)

    def points(self):
        a = self.d["points"]
        return a if isinstance(a, np.ndarray) else a.cpu().numpy()


# ---- Optimizer::PoseInertialOptimizationLastKeyFrame (tracking, inertial modes) -------------------------------------------------------
from .lba import POSE_EDGE_DTYPE  # noqa: E402

EDGE_CLOSE = 0x100   # pose_edge.kind flag: pFrame->mvpMapPoints[idx]->mTrackDepth < 10.f (Optimizer.cc:7852)


def synth_inertial_frame(seed=0, n_pts=300, kind="mono", outliers=0.08):
    """-> dict(frame, keyframe, rig, edges, imu): one tracked frame, its last key frame (fixed) and the preintegration between them."""
    w = synth_inertial_window(seed, n_opt=1, n_fixed_vis=0, n_pts=n_pts, max_obs=2, kind=kind, outliers=outliers)
    rng = np.random.default_rng(seed + 991)
    e = w["edges"][w["edges"]["pose"] == 1]
    pe = np.zeros(len(e), POSE_EDGE_DTYPE)
    pe["xw"] = w["points"][e["point"]].astype(np.float32)
    pe["obs"], pe["inv_sigma2"], pe["cam"] = e["obs"], e["inv_sigma2"], e["cam"]
    pe["kind"] = e["kind"] | np.where(rng.random(len(e)) < 0.6, EDGE_CLOSE, 0).astype(np.int16)
    imu = w["imu"][:1].copy()
    imu["huber"] = 0.0
    imu["info"] = imu["info"] * 1e2     # synth_inertial_window scaled the oldest edge's information by 1e-2
    return {"frame": w["kfs"][1:2].copy(), "keyframe": w["kfs"][0:1].copy(), "rig": w["rig"], "edges": pe, "imu": imu}


def pose_inertial_optimization_last_keyframe(frames, keyframes, rigs, edges, n_edges, imu, to_dev, *, rec_init=False, lib=None):
    """Batched Optimizer::PoseInertialOptimizationLastKeyFrame.  frames / keyframes: KF_DTYPE [B]; rigs: list of Rig (len B) or one Rig;
    edges: POSE_EDGE_DTYPE [B, cap_e]; n_edges int32 [B]; imu: IMU_EDGE_DTYPE [B].  -> (frames' [B] KF_DTYPE, outlier [B, cap_e] u8, H [B,15,15], n_good [B])"""
    L = lib if lib is not None else _lib.load()
    fn = L.liba_pose_inertial_kf
    vp, i32 = C.c_void_p, C.c_int
    fn.restype = i32
    fn.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp]
    B, cap_e = edges.shape[0], edges.shape[1]
    rl = rigs if isinstance(rigs, (list, tuple)) else [rigs]
    rig_bytes = np.frombuffer(b"".join(bytes(r) for r in rl), np.uint8).copy()
    d_f = to_dev(np.ascontiguousarray(frames).view(np.uint8).reshape(B, -1).copy())
    d_k = to_dev(np.ascontiguousarray(keyframes).view(np.uint8).reshape(B, -1))
    d_r = to_dev(rig_bytes)
    d_e = to_dev(np.ascontiguousarray(edges).view(np.uint8).reshape(B, -1))
    d_n = to_dev(np.ascontiguousarray(n_edges, np.int32))
    d_i = to_dev(np.ascontiguousarray(imu).view(np.uint8).reshape(B, -1))
    d_o, d_H, d_g = to_dev(np.zeros((B, cap_e), np.uint8)), to_dev(np.zeros((B, 225))), to_dev(np.zeros(B, np.int32))
    stream = None
    if not isinstance(d_f, np.ndarray):
        import torch
        stream = C.c_void_p(torch.cuda.current_stream(d_f.device).cuda_stream)
    rc = fn(_ptr(d_f), _ptr(d_k), _ptr(d_r), 1 if len(rl) > 1 else 0, _ptr(d_e), _ptr(d_n), cap_e, _ptr(d_i), B, int(rec_init), _ptr(d_o), _ptr(d_H), _ptr(d_g),
            stream)
    if rc != 0:
        raise OrbHipError(rc, "liba_pose_inertial_kf failed")
    host = lambda a: a if isinstance(a, np.ndarray) else a.cpu().numpy()
    return host(d_f).reshape(B, -1).view(KF_DTYPE).reshape(B), host(d_o), host(d_H).reshape(B, 15, 15), host(d_g)


PRIOR_DTYPE = np.dtype([("Rwb", "<f8", (9,)), ("twb", "<f8", (3,)), ("vwb", "<f8", (3,)), ("bg", "<f8", (3,)), ("ba", "<f8", (3,)), ("H", "<f8", (225,))])
assert PRIOR_DTYPE.itemsize == 246 * 8


def synth_prior(prev, seed=0):
    """A ConstraintPoseImu for key-frame record `prev`: linearisation point near the state, H = a positive definite 15x15 of realistic scale."""
    rng = np.random.default_rng(seed + 4242)
    c = np.zeros(1, PRIOR_DTYPE)
    c["Rwb"] = (prev["Rwb"].reshape(3, 3) @ _rodrigues(rng.normal(0, 1e-3, 3))).reshape(-1)
    c["twb"] = prev["twb"] + rng.normal(0, 5e-3, 3); c["vwb"] = prev["v"] + rng.normal(0, 1e-2, 3)
    c["bg"] = prev["bg"] + rng.normal(0, 1e-4, 3); c["ba"] = prev["ba"] + rng.normal(0, 1e-3, 3)
    Lm = np.diag(np.sqrt([2e4] * 3 + [5e3] * 3 + [1e3] * 3 + [3e5] * 3 + [2e3] * 3)) @ (np.eye(15) + 0.05 * rng.normal(0, 1, (15, 15)))
    H = Lm @ Lm.T
    c["H"] = ((H + H.T) / 2).reshape(-1)
    return c


def pose_inertial_optimization_last_frame(frames, prevs, rigs, edges, n_edges, imu, priors, to_dev, *, rec_init=False, lib=None):
    """Batched Optimizer::PoseInertialOptimizationLastFrame.  -> (frames', prevs', outlier [B, cap_e], H [B,15,15] (the marginalised prior), n_good [B])"""
    L = lib if lib is not None else _lib.load()
    fn = L.liba_pose_inertial_lastframe
    vp, i32 = C.c_void_p, C.c_int
    fn.restype = i32
    fn.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, vp, i32, i32, vp, vp, vp, vp]
    B, cap_e = edges.shape[0], edges.shape[1]
    rl = rigs if isinstance(rigs, (list, tuple)) else [rigs]
    rig_bytes = np.frombuffer(b"".join(bytes(r) for r in rl), np.uint8).copy()
    u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(B, -1)
    d_f, d_p = to_dev(u8(frames).copy()), to_dev(u8(prevs).copy())
    d_r, d_e, d_n, d_i, d_c = to_dev(rig_bytes), to_dev(u8(edges)), to_dev(np.ascontiguousarray(n_edges, np.int32)), to_dev(u8(imu)), to_dev(u8(priors))
    d_o, d_H, d_g = to_dev(np.zeros((B, cap_e), np.uint8)), to_dev(np.zeros((B, 225))), to_dev(np.zeros(B, np.int32))
    stream = None
    if not isinstance(d_f, np.ndarray):
        import torch
        stream = C.c_void_p(torch.cuda.current_stream(d_f.device).cuda_stream)
    rc = fn(_ptr(d_f), _ptr(d_p), _ptr(d_r), 1 if len(rl) > 1 else 0, _ptr(d_e), _ptr(d_n), cap_e, _ptr(d_i), _ptr(d_c), B, int(rec_init), _ptr(d_o), _ptr(d_H),
            _ptr(d_g), stream)
    if rc != 0:
        raise OrbHipError(rc, "liba_pose_inertial_lastframe failed")
    host = lambda a: a if isinstance(a, np.ndarray) else a.cpu().numpy()
    kfv = lambda a: host(a).reshape(B, -1).view(KF_DTYPE).reshape(B)
    return kfv(d_f), kfv(d_p), host(d_o), host(d_H).reshape(B, 15, 15), host(d_g)


class PoseInertialBatch:
    """Device-resident batch for liba_pose_inertial_kf / liba_pose_inertial_lastframe: upload once, run many times (bench / pipelines)."""

    def __init__(self, frames, others, rigs, edges, n_edges, imu, to_dev, priors=None, lib=None):
        self._L = lib if lib is not None else _lib.load()
        B, self.cap_e = edges.shape[0], edges.shape[1]
        self.B = B
        rl = rigs if isinstance(rigs, (list, tuple)) else [rigs]
        self.rig_stride = 1 if len(rl) > 1 else 0
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(B, -1)
        self.f0, self.o0 = to_dev(u8(frames).copy()), to_dev(u8(others).copy())
        self.f, self.o = to_dev(u8(frames).copy()), to_dev(u8(others).copy())
        self.r, self.e, self.n, self.i = to_dev(np.frombuffer(b"".join(bytes(r) for r in rl), np.uint8).copy()), to_dev(u8(edges)), to_dev(np.ascontiguousarray(n_edges, np.int32)), to_dev(u8(imu))
        self.c = None if priors is None else to_dev(u8(priors))
        self.out, self.H, self.g = to_dev(np.zeros((B, self.cap_e), np.uint8)), to_dev(np.zeros((B, 225))), to_dev(np.zeros(B, np.int32))

    def run(self, rec_init=False):
        vp, i32 = C.c_void_p, C.c_int
        if not isinstance(self.f, np.ndarray):
            import torch
            self.f.copy_(self.f0); self.o.copy_(self.o0)
            stream = C.c_void_p(torch.cuda.current_stream(self.f.device).cuda_stream)
        else:
            self.f[...] = self.f0; self.o[...] = self.o0
            stream = None
        if self.c is None:
            fn = self._L.liba_pose_inertial_kf
            fn.restype = i32
            fn.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp]
            rc = fn(_ptr(self.f), _ptr(self.o), _ptr(self.r), self.rig_stride, _ptr(self.e), _ptr(self.n), self.cap_e, _ptr(self.i), self.B, int(rec_init), _ptr(self.out),
                    _ptr(self.H), _ptr(self.g), stream)
        else:
            fn = self._L.liba_pose_inertial_lastframe
            fn.restype = i32
            fn.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, vp, i32, i32, vp, vp, vp, vp]
            rc = fn(_ptr(self.f), _ptr(self.o), _ptr(self.r), self.rig_stride, _ptr(self.e), _ptr(self.n), self.cap_e, _ptr(self.i), _ptr(self.c), self.B, int(rec_init),
                    _ptr(self.out), _ptr(self.H), _ptr(self.g), stream)
        if rc != 0:
            raise OrbHipError(rc, "liba_pose_inertial_* failed")
        return self.g
