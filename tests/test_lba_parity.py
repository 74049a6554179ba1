"""Stage-3 parity: HIP LBA linearisation vs the oracle's restatement of g2o's buildSystem for the LBA graph.
Tolerances: H blocks 1e-10 relative, chi2 1e-9, residuals 1e-4 absolute (north_star)."""
import numpy as np
import pytest

import oracle_lib as O
import orbhip
from orbhip.lba import EDGE_BODY, EDGE_MONO, EDGE_STEREO, HUBER_MONO, HUBER_STEREO, LbaWindows, rot_to_quat, synth_window

HUBER = (HUBER_MONO, HUBER_STEREO)
_W = {}


def window(kind, seed=0, n_kf=12, n_fixed=3, n_pts=300):
    k = (kind, seed, n_kf, n_fixed, n_pts)
    if k not in _W:
        _W[k] = synth_window(seed, n_kf, n_fixed, n_pts, 6, kind)
    return _W[k]


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def to_dev(backend):
    if backend == "emu":
        return lambda a: a
    import torch
    return lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def to_host(a):
    return a if isinstance(a, np.ndarray) else a.cpu().numpy()


@pytest.mark.parametrize("kind", ["mono", "stereo", "kb8", "body"])
def test_oracle_jacobians_match_finite_differences(kind):
    """Pins linearizeOplus (and the sign/layout conventions of Appendix A.16) to the error function by central differences."""
    w, cams = window(kind)
    o = O.lba_build_system(w, cams, (0.0, 0.0))  # no robust kernel: H = J^T Omega J exactly
    rng = np.random.default_rng(0)
    for ei in rng.choice(len(w["edges"]), 12, replace=False):
        E = w["edges"][ei]
        pose, pt, cam = w["poses"][E["pose"]], w["points"][E["point"]], cams[E["cam"]]
        D = 3 if E["kind"] == EDGE_STEREO else 2
        # the stereo edge projects with a float32 1/z (types_six_dof_expmap.cpp:191) and KB8 rounds theta/psi through
        # atan2f/sqrtf (KannalaBrandt8.cpp:54-55): those error functions are only float-smooth -> large central-difference step
        h, tol = (2e-3, 5e-3) if (E["kind"] == EDGE_STEREO or cam["model"] == 1) else (1e-6, 2e-5)
        J = np.zeros((D, 9))
        for k in range(9):
            dp, dx = np.zeros(6), np.zeros(3)
            (dp if k < 6 else dx)[k if k < 6 else k - 6] = h
            ep = O.lba_edge_error(pose, pt, E, cam, dp, dx)
            em = O.lba_edge_error(pose, pt, E, cam, -dp, -dx)
            J[:, k] = (ep - em)[:D] / (2 * h)
        B, A = J[:, :6], J[:, 6:]
        s = float(E["inv_sigma2"])
        Hpl = (B.T * s) @ A   # 6x3
        if w["pose_hidx"][E["pose"]] >= 0:
            got = o["Hpl"][ei].reshape(3, 6).T
            assert np.abs(got - Hpl).max() <= tol * max(np.abs(Hpl).max(), 1.0), (kind, ei)
        e0 = O.lba_edge_error(pose, pt, E, cam, np.zeros(6), np.zeros(3))
        assert np.allclose(o["err"][ei], e0, atol=1e-12)


def test_quaternion_conversion_roundtrip():
    rng = np.random.default_rng(1)
    for _ in range(50):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        q2 = O.quat_from_matrix(R)
        assert np.allclose(q2, q if w >= 0 else -q, atol=1e-12) and np.allclose(rot_to_quat(R), q2, atol=1e-14)


def check_window(lib, backend, kinds, B=2):
    ws, cams = [], None
    for i, kind in enumerate(kinds):
        w, cams = window(kind, seed=i, n_kf=12 + i, n_pts=300 + 17 * i)
        ws.append(w)
    L = LbaWindows(ws, cams, to_dev(backend), lib=lib, huber=HUBER)
    out = {k: to_host(v) for k, v in L.build_system().items()}
    errs = {k: to_host(v).copy() for k, v in L.compute_errors().items()}
    for b, w in enumerate(ws):
        o = O.lba_build_system(w, cams, HUBER)
        ne, nl, nf = len(w["edges"]), len(w["points"]), o["nfree"]
        # KB8 edges: the reference rounds theta/psi through libm's atan2f (KannalaBrandt8.cpp:54-55), whose last-ulp behaviour is
        # libm-specific; the product uses float(atan2(double)).  1 float ulp of theta ~ 2e-5 px: inside the 1e-4 bar, and the
        # only place where this stage is not ~1e-12 against the oracle.
        kb8 = bool((cams[w["edges"]["cam"]]["model"] == 1).any())
        e_tol, r_tol = (1e-4, 1e-4) if kb8 else (1e-9, 1e-10)
        assert np.abs(out["err"][b, :ne] - o["err"]).max() < e_tol      # north_star bar: 1e-4
        assert rel(out["chi2"][b, :ne], o["chi2"]) < max(r_tol, 1e-9) and np.abs(out["rho"][b, :ne] - o["rho"]).max() < max(r_tol, 1e-9) * max(1, o["rho"].max())
        assert np.abs(out["depth"][b, :ne] - o["depth"]).max() < 1e-10
        for k, n in (("Hll", nl), ("bl", nl), ("Hpl", ne), ("Hpp", nf), ("bp", nf)):
            assert rel(out[k][b, :n], o[k][:n]) < r_tol, (k, rel(out[k][b, :n], o[k][:n]))
        assert (out["Hpp"][b, nf:] == 0).all() and (out["bp"][b, nf:] == 0).all()
        assert np.abs(errs["err"][b, :ne] - o["err"]).max() < e_tol and rel(errs["chi2"][b, :ne], o["chi2"]) < max(r_tol, 1e-9)
        assert abs(errs["robust_chi2_sum"][b] - o["robust_chi2_sum"][0]) < max(r_tol, 1e-9) * o["robust_chi2_sum"][0]
        assert (o["rho"][:, 1] < 1).any() and (o["rho"][:, 1] == 1).any()   # both Huber branches exercised
        Hpp = out["Hpp"][b, :nf].reshape(nf, 6, 6)
        assert np.allclose(Hpp, Hpp.transpose(0, 2, 1), rtol=0, atol=0)       # symmetric blocks


@pytest.mark.parametrize("kinds", [("mono", "stereo"), ("kb8", "body"), ("mixed", "mixed")], ids=lambda k: "+".join(k))
def test_emu_lba_matches_oracle(emu_lib, kinds):
    check_window(emu_lib, "emu", kinds)


@pytest.mark.gpu
@pytest.mark.parametrize("kinds", [("mono", "stereo"), ("kb8", "body"), ("mixed", "mixed")], ids=lambda k: "+".join(k))
def test_hip_lba_matches_oracle(hip_lib, kinds):
    check_window(hip_lib, "hip", kinds)


@pytest.mark.gpu
def test_hip_lba_c5_size_window(hip_lib):
    """BASELINE configs[4]-size window (100 KF / 20k landmarks): parity on the full problem + determinism."""
    w, cams = synth_window(7, 100, 20, 20000, 8, "mono")
    assert 1.0e5 < len(w["edges"]) < 1.7e5
    L = LbaWindows([w], cams, to_dev("hip"), lib=hip_lib, huber=HUBER)
    a = {k: to_host(v).copy() for k, v in L.build_system().items()}
    b = {k: to_host(v).copy() for k, v in L.build_system().items()}
    o = O.lba_build_system(w, cams, HUBER)
    for k in ("Hpp", "bp", "Hll", "bl", "Hpl", "chi2"):
        n = len(o[k]) if k not in ("Hpp", "bp") else o["nfree"]
        assert np.array_equal(a[k], b[k]), "run-to-run determinism of " + k
        assert rel(a[k][0, :n], o[k][:n]) < 1e-10, k


# ---- the LM loop (Schur complement + Cholesky + rho test) ------------------------------------------------------------
def check_optimize(lib, backend, kinds, iterations, huber=HUBER, pose_tol=1e-7):
    ws, cams = [], None
    for i, kind in enumerate(kinds):
        w, cams = window(kind, seed=10 + i, n_kf=10 + i, n_pts=200 + 31 * i)
        ws.append(w)
    L = LbaWindows(ws, cams, to_dev(backend), lib=lib, huber=huber)
    stats = L.optimize(iterations)
    poses, points = to_host(L.d["poses"]), to_host(L.d["points"])
    for b, w in enumerate(ws):
        op, ox, ost = O.lba_optimize(w, cams, huber, iterations)
        npz, nl = len(w["poses"]), len(w["points"])
        assert stats[b, 0] == ost[0] and stats[b, 3] == ost[3], (stats[b], ost)          # same iterations / lambda trials
        assert abs(stats[b, 1] - ost[1]) < 1e-6 * ost[1]
        # north_star bar: 1e-4 on BA poses; double arithmetic in both, different summation orders
        assert np.abs(poses[b, :npz] - op).max() < pose_tol, np.abs(poses[b, :npz] - op).max()
        assert np.abs(points[b, :nl] - ox).max() < 10 * pose_tol
        assert ost[1] < 0.9 * O.lba_build_system(w, cams, huber)["robust_chi2_sum"][0]                                # it really optimised
        assert np.abs(poses[b, :npz][w["pose_hidx"] < 0] - w["poses"][w["pose_hidx"] < 0]).max() == 0    # fixed KFs untouched


@pytest.mark.parametrize("kinds,its", [(("mono", "stereo"), 5), (("mono",), 10), (("body", "mixed"), 4)], ids=["mono+stereo_5it", "mono_10it", "body+mixed_4it"])
def test_emu_lba_optimize_matches_oracle(emu_lib, kinds, its):
    # fisheye / right-camera edges: KannalaBrandt8::project rounds theta and psi to float (KannalaBrandt8.cpp:52-66), a step function of the pose —
    # two double-precision solvers that differ in the last bits can land on different float steps: 2e-7 on these windows (the same figure with the
    # round-1 Schur / Cholesky kernels), bar 1e-4
    check_optimize(emu_lib, "emu", kinds, its, pose_tol=1e-6 if "body" in kinds else 1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("kinds,its", [(("mono", "stereo"), 5), (("mono", "mono", "stereo"), 10), (("body", "kb8", "mixed"), 5)],
                         ids=["mono+stereo_5it", "3win_10it", "body+kb8+mixed_5it"])
def test_hip_lba_optimize_matches_oracle(hip_lib, kinds, its):
    check_optimize(hip_lib, "hip", kinds, its, pose_tol=1e-6 if "body" in kinds else 1e-7)


def _optimize_fuzz(lib, backend, seeds):
    """Random small windows: 1 ... 23 free key frames (odd and even counts: the cyclic block split has a tie slot for even ones; panels
    end in ragged blocks), 1 ... 6 fixed ones, every edge family, batches of three different windows (ragged reduced systems in one launch)."""
    rng = np.random.default_rng(2024)
    hard, checked = [], [0]
    for seed in seeds:
        kind = ("mono", "stereo", "mixed", "body", "kb8")[seed % 5]
        ws, cams = [], None
        for j in range(3):
            nfree, nfix = int(rng.integers(1, 24)), int(rng.integers(1, 7))   # always a fixed key frame: the reference never optimises a free gauge
            nfix = max(nfix, 3 - nfree, 2 if kind in ("mono", "kb8") else 1)   # >= 3 key frames; a monocular map needs two fixed ones (scale)
            w, cams = synth_window(1000 + 7 * seed + j, nfree + nfix, nfix, int(rng.integers(40, 400)), min(int(rng.integers(3, 9)), nfree + nfix), kind)
            ws.append(w)
        L = LbaWindows(ws, cams, to_dev(backend), lib=lib, huber=HUBER)
        its = int(rng.integers(2, 6))
        stats = L.optimize(its)
        poses = to_host(L.d["poses"])
        for b, w in enumerate(ws):
            op, ox, ost = O.lba_optimize(w, cams, HUBER, its)
            assert stats[b, 0] == ost[0], (seed, b, kind, stats[b], ost)
            if max(ost[3], stats[b, 3]) > 2 * its:
                # A window that rejects step after step (seen: 15 lambda trials in the first iteration of a fisheye rig with one fixed key frame:
                # the damping climbs from 1e-50 to 1e-13 before a step is accepted) solves a nearly singular reduced system; rounding-level
                # differences between two correct solvers (the device's atan2 / MFMA sums vs glibc / serial sums) come out of that solve
                # amplified — 9 cm in a pose at chi2 equal to 7e-5 — and a rho test within rounding of zero can fall either way.  Such a window
                # is compared on what LM minimises, and there may be only a few of them.
                # (one solver may even accept at once what the other rejects fourteen times: 3 vs 17 trials, final chi2 equal to 9e-5.)
                assert abs(stats[b, 1] - ost[1]) <= 2e-4 * ost[1], (seed, b, kind, stats[b], ost)
                hard.append((seed, b))
                continue
            assert stats[b, 3] == ost[3], (seed, b, kind, stats[b], ost)
            # fisheye edges: KannalaBrandt8::project rounds theta / psi to float, and the device's double atan2 is within an ulp of glibc's, not
            # identical — now and then the float lands one step apart (PoseOptimization's fisheye cases carry the same 5e-5)
            fish = kind in ("body", "kb8", "mixed")
            assert abs(stats[b, 1] - ost[1]) <= (1e-4 if fish else 1e-6) * max(ost[1], 1e-9), (seed, b, kind)
            # poses are compared where the window pins them down (a key frame with a couple of dozen observations sits in a flat valley of chi2)
            obs_per_pose = np.bincount(w["edges"]["pose"], minlength=len(w["poses"]))[w["pose_hidx"] >= 0]
            if obs_per_pose.min() >= 40:
                assert np.abs(poses[b, :len(w["poses"])] - op).max() < (5e-5 if fish else 1e-7), (seed, b, kind)
                checked[0] += 1
    print("lba optimize fuzz [%s]: %d windows, %d compared pose by pose, %d ill-conditioned (chi2 only): %s" % (backend, 3 * len(list(seeds)), checked[0], len(hard), hard))
    assert checked[0] >= len(list(seeds)), checked   # at least one window per seed on average is well-posed enough for the pose comparison
    assert len(hard) <= max(1, len(list(seeds)) // 6), hard


def test_emu_lba_optimize_fuzz(emu_lib):
    _optimize_fuzz(emu_lib, "emu", range(4))


@pytest.mark.gpu
def test_hip_lba_optimize_fuzz(hip_lib):
    _optimize_fuzz(hip_lib, "hip", range(25))


@pytest.mark.gpu
def test_hip_lba_optimize_c5_size(hip_lib):
    """optimize() at BASELINE configs[4]'s size: 100 key frames (80 free), 20 000 landmarks, ~160 000 monocular edges — 480 unknowns in the
    reduced camera system, the 32-column Cholesky panel, the monocular-pinhole kernels — and a 60-KF / 6 000-landmark stereo window (the
    generic kernels) in the same batch class.  Same iteration / lambda-trial counts as the oracle's g2o restatement, poses within 1e-6."""
    for kind, args in (("mono", (100, 20, 20000)), ("stereo", (60, 10, 6000))):
        w, cams = synth_window(77, args[0], args[1], args[2], 8, kind)
        L = LbaWindows([w], cams, to_dev("hip"), lib=hip_lib, huber=HUBER)
        assert L.mono_pinhole == (kind == "mono")
        stats = L.optimize(2)
        op, ox, ost = O.lba_optimize(w, cams, HUBER, 2)
        assert stats[0, 0] == ost[0] and stats[0, 3] == ost[3], (stats[0], ost)
        assert abs(stats[0, 1] - ost[1]) < 1e-6 * ost[1]
        assert np.abs(to_host(L.d["poses"])[0, :args[0]] - op).max() < 1e-6
        assert np.abs(to_host(L.d["points"])[0, :args[2]] - ox).max() < 1e-5


def test_emu_global_ba_parameterisation(emu_lib):
    """Optimizer::BundleAdjustment / GlobalBundleAdjustemnt (Optimizer.cc:66-458) builds the same graph as the local BA — EdgeSE3ProjectXYZ,
    EdgeStereoSE3ProjectXYZ, EdgeSE3ProjectXYZToBody, first key frame fixed — and runs optimizer.optimize(nIterations); with bRobust == false
    the edges carry no robust kernel.  That is lba_optimize with huber deltas 0 (delta > 0 gates the kernel)."""
    check_optimize(emu_lib, "emu", ("mono", "stereo"), 10, huber=(0.0, 0.0))


@pytest.mark.gpu
def test_hip_global_ba_parameterisation(hip_lib):
    # without the Huber kernel the 5 % gross outliers of the synthetic windows pull hard: 10 iterations amplify the summation-order
    # difference to ~2e-7 (measured 1.8e-7 on MI355X); the bar is 1e-4
    check_optimize(hip_lib, "hip", ("mono", "stereo", "body"), 10, huber=(0.0, 0.0), pose_tol=1e-6)


# ---- lba_optimize sizes the reduced camera system by the FREE key frames (round-1 limit: cap_p <= 180 including the fixed ones) -------------
def _many_fixed(lib, backend):
    """A local window whose fixed key frames (lFixedCameras, Optimizer.cc:2011-2039: every key frame that sees a local point) far outnumber the
    optimised ones: 214 poses, 14 free.  The reference puts no limit on them."""
    w, cams = synth_window(31, 214, 200, 260, 7, "mono")
    assert (w["pose_hidx"] >= 0).sum() == 14 and len(w["poses"]) > 180
    L = LbaWindows([w], cams, to_dev(backend), lib=lib, huber=HUBER)
    stats = L.optimize(5)
    op, ox, ost = O.lba_optimize(w, cams, HUBER, 5)
    assert stats[0, 0] == ost[0] and stats[0, 3] == ost[3]
    assert np.abs(to_host(L.d["poses"])[0, :214] - op).max() < 1e-7
    assert np.abs(to_host(L.d["points"])[0, :len(ox)] - ox).max() < 1e-6


def test_emu_lba_optimize_many_fixed_keyframes(emu_lib):
    _many_fixed(emu_lib, "emu")


@pytest.mark.gpu
def test_hip_lba_optimize_many_fixed_keyframes(hip_lib):
    _many_fixed(hip_lib, "hip")


def test_emu_lba_optimize_global_memory_panel():
    """WG_CHOL_LDS_MAX_LD=30: every window's reduced system (>= 42 unknowns) is 'too large for LDS', so the Cholesky panel lives in the
    workspace (the path GlobalBundleAdjustemnt of a map with more than 180 free key frames takes); results must not change."""
    import ctypes
    import build_emu
    from orbhip import _lib
    # (LM_CHOL_SPLIT_MAX_BATCH=0: few windows per call would otherwise take one launch per panel, which has no LDS panel at all)
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("WG_CHOL_LDS_MAX_LD=30", "LM_CHOL_SPLIT_MAX_BATCH=0"), tag="cholext")))
    check_optimize(lib, "emu", ("mono", "stereo"), 5)


def test_emu_lba_optimize_schur_row_chunks():
    """LM_SCHUR_ROWCAP=3: every row of the reduced camera system longer than three blocks is produced in column chunks (the path a map with more
    than 384 free key frames takes); results must not change."""
    import ctypes
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("LM_SCHUR_ROWCAP=3",), tag="schurchunk")))
    check_optimize(lib, "emu", ("mono", "stereo"), 5)


def test_emu_lba_optimize_split_schur_rows():
    """LM_SCHUR_SPLIT_MIN_EDGES=4: every row of the reduced camera system is split over several workgroups (k_lm_schur_rows slices + k_lm_schur_combine
    — what ONE LocalMapping window per call takes on the GPU, where 80 rows would otherwise use 80 of 256 compute units); results must not change."""
    import ctypes
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("LM_SCHUR_SPLIT_MIN_EDGES=4",), tag="schursplit")))
    check_optimize(lib, "emu", ("mono", "stereo"), 5)


def test_emu_lba_optimize_split_and_chunked_schur_rows():
    """Both knobs at once (round-5 advisor finding): a row that is split over workgroups leaves its slice in ONE column chunk of schurPart, so rows
    longer than the LDS chunk (n / 2 + 1 > LM_SCHUR_ROWCAP) must not be split — lm_schur_groups returns 1 for them; results must not change."""
    import ctypes
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("LM_SCHUR_ROWCAP=3", "LM_SCHUR_SPLIT_MIN_EDGES=4"), tag="schurchunksplit")))
    check_optimize(lib, "emu", ("mono", "stereo"), 5)


def test_emu_lba_optimize_one_workgroup_per_window():
    """LM_CHOL_SPLIT_MAX_BATCH=0: the factorisation as ONE workgroup per window with 16-column panels (k_lm_chol<16>, what a batch of more than 48
    small windows takes).  The default build's few-window calls — every other LM test of this tier — take one launch per panel (k_lm_chol_step +
    k_lm_chol_back_x: inverse of the diagonal block, panel rows and trailing tiles on the fp64 matrix core)."""
    import ctypes
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("LM_CHOL_SPLIT_MAX_BATCH=0",), tag="cholmono16")))
    check_optimize(lib, "emu", ("mono", "stereo"), 5)


def test_emu_lba_optimize_wide_panel_one_workgroup():
    """WG_CHOL_NB32_MIN_LD=0 + LM_CHOL_SPLIT_MAX_BATCH=0: the 32-column panel inside ONE workgroup per window (k_lm_chol<32>, what a batch of more
    than 48 windows of 54 ... 88 free key frames takes) on the small test windows."""
    import ctypes
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("WG_CHOL_NB32_MIN_LD=0", "LM_CHOL_SPLIT_MAX_BATCH=0"), tag="cholnb32mono")))
    check_optimize(lib, "emu", ("mono", "stereo"), 5)


@pytest.mark.gpu
def test_hip_cholesky_per_phase_launches_agree_with_one_workgroup(hip_lib):
    """60 free key frames = 360 unknowns.  One window per call takes one launch per 32-column panel (k_lm_chol_step: reciprocal-square-root pivots,
    the panel rows as products with the INVERSE of the diagonal block on the fp64 matrix core, Schur rows split over three workgroups each),
    129 windows per call one workgroup per window (sqrt / divide, triangular solves): the same factorisation to rounding — poses within 1e-9 of
    each other, both within 1e-6 of the oracle — and the 129 copies of the batch bit-identical among themselves."""
    w, cams = synth_window(5, 70, 10, 5000, 8, "mono")
    assert (w["pose_hidx"] >= 0).sum() == 60
    L1 = LbaWindows([w], cams, to_dev("hip"), lib=hip_lib, huber=HUBER)
    s1 = L1.optimize(3)
    L33 = LbaWindows([w] * 129, cams, to_dev("hip"), lib=hip_lib, huber=HUBER)
    s33 = L33.optimize(3)
    p1, p33 = to_host(L1.d["poses"])[0], to_host(L33.d["poses"])
    assert s1[0, 0] == s33[0, 0] and s1[0, 3] == s33[0, 3] and abs(s1[0, 1] - s33[0, 1]) < 1e-9 * s33[0, 1]
    assert np.abs(p1 - p33[0]).max() < 1e-9
    assert np.array_equal(p33[0], p33[128]) and np.array_equal(s33[0], s33[128])
    assert np.array_equal(to_host(L33.d["points"])[3], to_host(L33.d["points"])[17])
    op, ox, ost = O.lba_optimize(w, cams, HUBER, 3)
    assert s1[0, 0] == ost[0] and s1[0, 3] == ost[3]
    assert np.abs(p1[:70] - op).max() < 1e-6 and np.abs(p33[0, :70] - op).max() < 1e-6


@pytest.mark.gpu
def test_hip_lba_optimize_120_free_keyframes(hip_lib):
    """720 unknowns: 16-column Cholesky panel in LDS (between the 32-column panel's 88 free key frames and the global-memory panel's 177+)."""
    w, cams = synth_window(43, 130, 10, 2000, 8, "stereo")
    assert (w["pose_hidx"] >= 0).sum() == 120
    L = LbaWindows([w], cams, to_dev("hip"), lib=hip_lib, huber=(0.0, 0.0))
    stats = L.optimize(3)
    op, ox, ost = O.lba_optimize(w, cams, (0.0, 0.0), 3)
    assert stats[0, 0] == ost[0] and stats[0, 3] == ost[3]
    assert np.abs(to_host(L.d["poses"])[0, :130] - op).max() < 1e-5


@pytest.mark.gpu
def test_hip_global_ba_more_than_180_free_keyframes(hip_lib):
    """Optimizer::BundleAdjustment over a map of 200 key frames (first one fixed): 199 free poses = 1194 unknowns, beyond the LDS panel."""
    # stereo observations: with one fixed key frame a monocular map keeps its scale gauge freedom, the reduced system is singular up to the LM
    # damping and two correct solvers legitimately walk different lambda sequences (seen on MI355X: 17 vs 3 trials on a 170-KF mono map)
    w, cams = synth_window(41, 200, 1, 2500, 8, "stereo")
    assert (w["pose_hidx"] >= 0).sum() == 199
    L = LbaWindows([w], cams, to_dev("hip"), lib=hip_lib, huber=(0.0, 0.0))
    stats = L.optimize(3)
    op, ox, ost = O.lba_optimize(w, cams, (0.0, 0.0), 3)
    assert stats[0, 0] == ost[0] and stats[0, 3] == ost[3]
    assert np.abs(to_host(L.d["poses"])[0, :200] - op).max() < 1e-5   # 1194 unknowns, no robust kernel; the bar is 1e-4
