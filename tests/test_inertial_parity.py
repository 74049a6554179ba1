"""Visual-inertial local BA (Optimizer::LocalInertialBA, SURVEY N4 tail): oracle pins (central differences of every Jacobian through the
vertices' own oplus, convergence) and HIP-vs-oracle parity of the optimised states."""
import numpy as np
import pytest

import oracle_lib as O
from orbhip.inertial import HUBER_INERTIAL, synth_inertial_window
from orbhip.lba import HUBER_MONO, HUBER_STEREO

HUBER = (HUBER_MONO, HUBER_STEREO)
_W = {}


def window(kind, seed=0, **kw):
    k = (kind, seed, tuple(sorted(kw.items())))
    if k not in _W:
        _W[k] = synth_inertial_window(seed, kind=kind, **kw)
    return _W[k]


@pytest.mark.parametrize("kind", ["mono", "stereo", "fisheye"])
def test_oracle_inertial_jacobian_matches_central_differences(kind):
    w = window(kind, n_pts=120)
    kfs, rig, imu = w["kfs"], w["rig"], w["imu"]
    for E in imu:
        k1, k2 = kfs[E["kf1"]:E["kf1"] + 1], kfs[E["kf2"]:E["kf2"] + 1]
        J = O.inertial_imu_jacobian(E, k1, k2)
        e0 = O.inertial_imu_error(E, k1, k2, rig)
        assert np.abs(e0).max() < 0.2          # the synthetic preintegration is consistent with the trajectory
        Jn = np.zeros((9, 24))
        # the error goes through float32 preintegration getters and a float-normalised ExpSO3 (rule R3): only float-smooth -> large step
        h = 2e-3
        for c in range(24):
            d = np.zeros(24); d[c] = h
            Jn[:, c] = (O.inertial_imu_error(E, k1, k2, rig, d) - O.inertial_imu_error(E, k1, k2, rig, -d)) / (2 * h)
        assert np.abs(J - Jn).max() < 2e-3 * max(1.0, np.abs(J).max()), (np.abs(J - Jn).max(), np.unravel_index(np.abs(J - Jn).argmax(), J.shape))


@pytest.mark.parametrize("kind", ["mono", "stereo", "fisheye"])
def test_oracle_visual_jacobians_match_central_differences(kind):
    w = window(kind, n_pts=120)
    kfs, rig, pts, edges = w["kfs"], w["rig"], w["points"], w["edges"]
    rng = np.random.default_rng(1)
    for ei in rng.choice(len(edges), 10, replace=False):
        E = edges[ei:ei + 1]
        kf, X = kfs[E["pose"][0]:E["pose"][0] + 1], pts[E["point"][0]]
        e0, A, B = O.inertial_vis_error(E, kf, rig, X, jac=True)
        D = 3 if E["kind"][0] == 1 else 2
        h = 2e-3 if rig.model[E["cam"][0]] == 1 else 1e-4     # KB8 rounds theta / psi through atan2f; pose oplus through a float ExpSO3
        for c in range(6):
            d = np.zeros(6); d[c] = h
            n = (O.inertial_vis_error(E, kf, rig, X, dpose=d) - O.inertial_vis_error(E, kf, rig, X, dpose=-d)) / (2 * h)
            assert np.abs(n[:D] - B[:D, c]).max() < 5e-3 * max(1.0, np.abs(B).max()), (ei, c)
        hp = 2e-3 if rig.model[E["cam"][0]] == 1 else 1e-6
        for c in range(3):
            d = np.zeros(3); d[c] = hp
            n = (O.inertial_vis_error(E, kf, rig, X, dpoint=d) - O.inertial_vis_error(E, kf, rig, X, dpoint=-d)) / (2 * hp)
            assert np.abs(n[:D] - A[:D, c]).max() < (5e-3 if rig.model[E["cam"][0]] == 1 else 1e-5) * max(1.0, np.abs(A).max()), (ei, c)


@pytest.mark.parametrize("kind,lam", [("mono", 1.0), ("stereo", 1e-2), ("fisheye", 1.0)])
def test_oracle_inertial_ba_converges(kind, lam):
    w = window(kind)
    e0 = O.inertial_errors(w, HUBER)
    kfs, pts, st = O.inertial_optimize(w, HUBER, lam, 10)
    assert st[0] >= 3 and st[4] == pytest.approx(e0["robust_chi2_sum"]) and st[1] < 0.85 * st[4], st
    fixed = w["kfs"]["pose_fixed"] == 1
    assert np.array_equal(kfs[fixed], w["kfs"][fixed])
    assert not np.array_equal(kfs[~fixed]["v"], w["kfs"][~fixed]["v"]) and not np.array_equal(kfs[~fixed]["bg"], w["kfs"][~fixed]["bg"])
    assert (e0["imu_chi2"][:, 0] > 0).all() and e0["vis_depth_pos"].all()
    assert w["imu"]["huber"][-1] == HUBER_INERTIAL


# ---- HIP path vs oracle ---------------------------------------------------------------------------------------------------------------
from orbhip.inertial import InertialWindows  # noqa: E402


def to_dev(backend):
    if backend == "emu":
        return lambda a: a
    import torch
    return lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def to_host(a):
    return a if isinstance(a, np.ndarray) else a.cpu().numpy()


def check_inertial(lib, backend, kinds, lam, its, **kw):
    ws = [window(kind, seed=20 + i, n_opt=6 + i, n_pts=260 + 40 * i, **kw) for i, kind in enumerate(kinds)]
    IW = InertialWindows(ws, to_dev(backend), lib=lib, huber=HUBER)
    e = {k: to_host(v) for k, v in IW.compute_errors().items()}
    for b, w in enumerate(ws):
        o = O.inertial_errors(w, HUBER)
        ne, ni = len(w["edges"]), len(w["imu"])
        kb8 = w["rig"].model[0] == 1
        tol = 2e-3 if kb8 else 1e-9      # KB8: float(atan2(double)) vs atan2f, DESIGN.md section 2 (1 float ulp of theta ~ 2e-5 px)
        assert np.abs(e["vis_chi2"][b, :ne] - o["vis_chi2"]).max() <= tol * max(1.0, o["vis_chi2"].max())
        assert np.array_equal(e["vis_depth_pos"][b, :ne], o["vis_depth_pos"])
        assert np.abs(e["imu_chi2"][b, :ni] - o["imu_chi2"]).max() <= 1e-9 * max(1.0, o["imu_chi2"].max())
        assert abs(e["robust_chi2_sum"][b] - o["robust_chi2_sum"]) <= tol * o["robust_chi2_sum"]
    stats = to_host(IW.optimize(lam, its))
    kf, pts = IW.keyframes(), IW.points()
    for b, w in enumerate(ws):
        okf, opts, ost = O.inertial_optimize(w, HUBER, lam, its)
        nk, nl = len(w["kfs"]), len(w["points"])
        kb8 = w["rig"].model[0] == 1
        assert stats[b, 0] == ost[0] and stats[b, 3] == ost[3], (stats[b], ost)
        assert abs(stats[b, 1] - ost[1]) <= (1e-4 if kb8 else 1e-7) * ost[1] and abs(stats[b, 4] - ost[4]) <= (1e-4 if kb8 else 1e-9) * ost[4]
        # north_star bar: 1e-4 on BA states.  Pinhole: double arithmetic on both sides, different summation order + float-rounded ExpSO3
        # (rule R3) -> ~1e-7; KB8 adds the atan2f tolerance
        tol = 2e-5 if kb8 else 5e-6
        for f in ("Rwb", "twb", "v", "bg", "ba", "Rcw", "tcw"):
            assert np.abs(kf[b, :nk][f] - okf[f]).max() < tol, (f, np.abs(kf[b, :nk][f] - okf[f]).max())
        assert np.abs(pts[b, :nl] - opts).max() < 10 * tol
        fixed = w["kfs"]["pose_fixed"] == 1
        assert np.array_equal(kf[b, :nk][fixed], w["kfs"][fixed])
        assert ost[1] < 0.9 * ost[4]


@pytest.mark.parametrize("kinds,lam,its", [(("mono", "stereo"), 1.0, 5), (("fisheye",), 1e-2, 4)], ids=["mono+stereo", "fisheye"])
def test_emu_inertial_ba_matches_oracle(emu_lib, kinds, lam, its):
    check_inertial(emu_lib, "emu", kinds, lam, its)


@pytest.mark.gpu
@pytest.mark.parametrize("kinds,lam,its", [(("mono", "stereo", "mono"), 1.0, 10), (("fisheye", "stereo"), 1e-2, 4)], ids=["mono+stereo", "fisheye"])
def test_hip_inertial_ba_matches_oracle(hip_lib, kinds, lam, its):
    check_inertial(hip_lib, "hip", kinds, lam, its)


@pytest.mark.gpu
def test_hip_inertial_ba_is_deterministic_and_rejects_bad_windows(hip_lib):
    ws = [window("stereo", seed=31, n_opt=10, n_pts=600, max_obs=8), window("mono", seed=32, n_opt=5, n_pts=300)]
    runs = []
    for _ in range(2):
        IW = InertialWindows(ws, to_dev("hip"), lib=hip_lib, huber=HUBER)
        st = to_host(IW.optimize(1.0, 6))
        runs.append((st.copy(), IW.keyframes().copy(), IW.points().copy()))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
    # unsorted (not landmark-major) edges -> the window is refused (stats[0] = -1) and nothing is touched
    bad = dict(ws[1]); bad["edges"] = ws[1]["edges"][::-1].copy()
    IW = InertialWindows([ws[0], bad], to_dev("hip"), lib=hip_lib, huber=HUBER)
    st = to_host(IW.optimize(1.0, 3))
    assert st[0, 0] >= 1 and st[1, 0] == -1
    assert np.array_equal(IW.keyframes()[1, :len(bad["kfs"])], bad["kfs"])


@pytest.mark.gpu
def test_hip_inertial_ba_large_window(hip_lib):
    """bLarge (Optimizer.cc:4760-4764): 25 optimisable key frames (375 reduced unknowns), lambda 1e-2, optimize(4)."""
    w = window("stereo", seed=41, n_opt=25, n_fixed_vis=8, n_pts=900, max_obs=10, dt=0.12)
    IW = InertialWindows([w], to_dev("hip"), lib=hip_lib, huber=HUBER)
    st = to_host(IW.optimize(1e-2, 4))[0]
    okf, opts, ost = O.inertial_optimize(w, HUBER, 1e-2, 4)
    kf = IW.keyframes()[0, :len(w["kfs"])]
    assert st[0] == ost[0] and st[3] == ost[3] and abs(st[1] - ost[1]) < 1e-7 * ost[1], (st, ost)
    for f in ("Rwb", "twb", "v", "bg", "ba"):
        assert np.abs(kf[f] - okf[f]).max() < 5e-6, f
    assert np.abs(IW.points()[0, :len(w["points"])] - opts).max() < 5e-5 and ost[1] < 0.9 * ost[4]
    # more optimisable key frames than LIBA_MAX_FREE: refused by the wrapper's max_free check (ORB_E_INVALID)
    big = window("mono", seed=42, n_opt=33, n_fixed_vis=1, n_pts=200, dt=0.1)
    with pytest.raises(Exception):
        InertialWindows([big], to_dev("hip"), lib=hip_lib, huber=HUBER).optimize(1.0, 2)


# ---- Optimizer::PoseInertialOptimizationLastKeyFrame ----------------------------------------------------------------------------------
from orbhip.inertial import pose_inertial_optimization_last_keyframe, synth_inertial_frame  # noqa: E402
from orbhip.lba import POSE_EDGE_DTYPE  # noqa: E402


@pytest.mark.parametrize("kind", ["mono", "stereo", "fisheye"])
def test_oracle_pose_inertial_recovers_the_frame(kind):
    f = synth_inertial_frame(5, 320, kind)
    fr, outl, H, n = O.pose_inertial_kf(f["frame"], f["keyframe"], f["rig"], f["edges"], f["imu"])
    assert n == len(f["edges"]) - int(outl.sum()) and 0.5 * len(f["edges"]) < n < len(f["edges"])
    # the frame started 1.5 cm / 0.4 deg off the trajectory the observations and the preintegration were generated from
    truth = synth_inertial_frame(5, 320, kind, outliers=0.0)
    assert np.abs(fr["twb"] - f["frame"]["twb"]).max() > 1e-3
    Hs = (H + H.T) / 2
    assert np.linalg.eigvalsh(Hs).min() > 0 and np.abs(H[:9, :9] - H[:9, :9].T).max() < 1e-6 * np.abs(H).max()
    assert truth["frame"].dtype == fr.dtype


def check_pose_inertial(lib, backend, kinds, rec_init=False, n_pts=300):
    fs = [synth_inertial_frame(70 + i, n_pts + 40 * i, k) for i, k in enumerate(kinds)]
    B, cap = len(fs), max(len(f["edges"]) for f in fs) + 5
    edges = np.zeros((B, cap), POSE_EDGE_DTYPE)
    n = np.zeros(B, np.int32)
    for b, f in enumerate(fs):
        edges[b, :len(f["edges"])] = f["edges"]; n[b] = len(f["edges"])
    frames = np.concatenate([f["frame"] for f in fs]); kfs = np.concatenate([f["keyframe"] for f in fs]); imu = np.concatenate([f["imu"] for f in fs])
    fr, outl, H, good = pose_inertial_optimization_last_keyframe(frames, kfs, [f["rig"] for f in fs], edges, n, imu, to_dev(backend), rec_init=rec_init, lib=lib)
    for b, f in enumerate(fs):
        ofr, ooutl, oH, on = O.pose_inertial_kf(f["frame"], f["keyframe"], f["rig"], f["edges"], f["imu"], rec_init)
        kb8 = f["rig"].model[0] == 1
        assert good[b] == on and np.array_equal(outl[b, :n[b]], ooutl) and (outl[b, n[b]:] == 0).all(), (b, good[b], on)
        tol = 2e-5 if kb8 else 5e-6
        for fld in ("Rwb", "twb", "v", "bg", "ba", "Rcw", "tcw"):
            assert np.abs(fr[b][fld] - ofr[0][fld]).max() < tol, (b, fld, np.abs(fr[b][fld] - ofr[0][fld]).max())
        assert np.abs(H[b] - oH).max() <= (1e-3 if kb8 else 1e-6) * np.abs(oH).max(), b


def test_emu_pose_inertial_matches_oracle(emu_lib):
    check_pose_inertial(emu_lib, "emu", ("mono", "stereo", "fisheye"))


@pytest.mark.gpu
@pytest.mark.parametrize("rec", [False, True])
def test_hip_pose_inertial_matches_oracle(hip_lib, rec):
    check_pose_inertial(hip_lib, "hip", ("mono", "stereo", "fisheye", "mono"), rec_init=rec)


@pytest.mark.gpu
def test_hip_pose_inertial_few_inliers_recovery(hip_lib):
    """< 30 inliers and !bRecInit: the recovery pass (Optimizer.cc:7904-7934) re-admits edges below chi2 18 / 24."""
    check_pose_inertial(hip_lib, "hip", ("mono", "stereo"), n_pts=24)


# ---- Optimizer::PoseInertialOptimizationLastFrame -------------------------------------------------------------------------------------
from orbhip.inertial import PRIOR_DTYPE, pose_inertial_optimization_last_frame, synth_prior  # noqa: E402


def test_oracle_pose_inertial_lastframe_prior_is_psd_and_marginal():
    f = synth_inertial_frame(6, 300, "stereo")
    pr = synth_prior(f["keyframe"][0], 2)
    fr, pv, outl, H, n = O.pose_inertial_lastframe(f["frame"], f["keyframe"], f["rig"], f["edges"], f["imu"], pr)
    assert n == len(f["edges"]) - int(outl.sum()) and n > 150
    assert np.linalg.eigvalsh((H + H.T) / 2).min() > 0 and np.abs(pv["twb"] - f["keyframe"]["twb"]).max() > 1e-4    # the previous frame moved too
    assert np.abs(H - H.T).max() < 1e-3 * np.abs(H).max()      # symmetric up to the slightly non-symmetric random-walk information matrices


def check_pose_inertial_lastframe(lib, backend, kinds, rec_init=False, n_pts=300):
    fs = [synth_inertial_frame(90 + i, n_pts + 40 * i, k) for i, k in enumerate(kinds)]
    B, cap = len(fs), max(len(f["edges"]) for f in fs) + 5
    edges = np.zeros((B, cap), POSE_EDGE_DTYPE)
    n = np.zeros(B, np.int32)
    for b, f in enumerate(fs):
        edges[b, :len(f["edges"])] = f["edges"]; n[b] = len(f["edges"])
    frames = np.concatenate([f["frame"] for f in fs]); prevs = np.concatenate([f["keyframe"] for f in fs]); imu = np.concatenate([f["imu"] for f in fs])
    priors = np.concatenate([synth_prior(f["keyframe"][0], b) for b, f in enumerate(fs)])
    fr, pv, outl, H, good = pose_inertial_optimization_last_frame(frames, prevs, [f["rig"] for f in fs], edges, n, imu, priors, to_dev(backend), rec_init=rec_init,
                                                                  lib=lib)
    for b, f in enumerate(fs):
        ofr, opv, ooutl, oH, on = O.pose_inertial_lastframe(f["frame"], f["keyframe"], f["rig"], f["edges"], f["imu"], priors[b:b + 1], rec_init)
        kb8 = f["rig"].model[0] == 1
        assert good[b] == on and np.array_equal(outl[b, :n[b]], ooutl), (b, good[b], on)
        tol = 2e-5 if kb8 else 5e-6
        for fld in ("Rwb", "twb", "v", "bg", "ba", "Rcw", "tcw"):
            assert np.abs(fr[b][fld] - ofr[0][fld]).max() < tol and np.abs(pv[b][fld] - opv[0][fld]).max() < tol, (b, fld)
        assert np.abs(H[b] - oH).max() <= (1e-3 if kb8 else 1e-5) * np.abs(oH).max(), (b, np.abs(H[b] - oH).max() / np.abs(oH).max())


def test_emu_pose_inertial_lastframe_matches_oracle(emu_lib):
    check_pose_inertial_lastframe(emu_lib, "emu", ("mono", "stereo"))


@pytest.mark.gpu
@pytest.mark.parametrize("rec", [False, True])
def test_hip_pose_inertial_lastframe_matches_oracle(hip_lib, rec):
    check_pose_inertial_lastframe(hip_lib, "hip", ("mono", "stereo", "fisheye", "stereo"), rec_init=rec)
