"""SURVEY N3 parity: HIP Optimizer::PoseOptimization (one workgroup per frame, whole 4x10 LM schedule in one launch) vs the oracle's
restatement of Optimizer.cc:907-1273 + g2o's Levenberg-Marquardt.  Outlier flags and the return value (n inliers) must be identical;
the pose to 1e-7 (the reductions add in a different order than the serial loop).
KB8 frames: the device rounds theta/psi as float(atan2(double)) while glibc's atan2f differs from that in rare 1-ulp cases (DESIGN.md
"libm dependence"); with the oracle's atan2f substituted the emulated kernel agrees to 4e-16, with the device formula the LM's
accept/reject decisions near convergence ride on that float noise -> KB8_TOL."""
import numpy as np
import pytest

import oracle_lib as O
from orbhip.lba import POSE_EDGE_DTYPE, _quat_to_rot, pose_optimization, synth_pose_frames
from test_lba_parity import to_dev, to_host

KB8_TOL = 5e-5


def oracle_batch(f):
    res = [O.pose_optimize(f["poses"][b], f["edges"][b, :f["n_edges"][b]], f["cameras"]) for b in range(len(f["poses"]))]
    return res


def run(lib, backend, f):
    """Through the generic kernel, and — when every edge is monocular / stereo on a pinhole camera — through the LBA_HINT_PINHOLE kernel as
    well: the two must agree bit for bit (same expressions in the same order)."""
    td = to_dev(backend)
    B, cap = f["edges"].shape
    args = (td(f["poses"]), td(f["edges"].view(np.uint8).reshape(B, -1)), td(f["n_edges"]), td(np.ascontiguousarray(f["cameras"]).view(np.uint8)))
    res = [to_host(x) for x in pose_optimization(*args, lib=lib)]
    valid = np.arange(cap)[None, :] < f["n_edges"][:, None]
    if (f["edges"]["kind"][valid] != 2).all() and (f["cameras"]["model"][f["edges"]["cam"][valid]] == 0).all():   # no EDGE_BODY, only CAM_PINHOLE
        hinted = [to_host(x) for x in pose_optimization(*args, lib=lib, pinhole=True)]
        for a, b in zip(res, hinted):
            assert np.array_equal(a, b)
    return res


def check(lib, backend, kind, seed, n_pts, batch=5, tol=1e-7, **kw):
    f = synth_pose_frames(seed=seed, batch=batch, n_pts=n_pts, kind=kind, **kw)
    out, outl, ng = run(lib, backend, f)
    for b, (po, oo, no) in enumerate(oracle_batch(f)):
        n = f["n_edges"][b]
        assert ng[b] == no, (b, ng[b], no)
        assert (outl[b, :n] == oo).all(), (b, np.flatnonzero(outl[b, :n] != oo))
        assert (outl[b, n:] == 0).all()
        assert np.abs(out[b] - po).max() < tol, (b, np.abs(out[b] - po).max())
    return f, out, outl, ng


def test_oracle_pose_optimization_recovers_truth():
    """Sanity pin of the restatement itself: with 10% gross outliers the optimised pose reprojects the inliers to ~noise level and the
    outlier flags mark (almost exactly) the corrupted observations."""
    f = synth_pose_frames(seed=3, batch=2, n_pts=400, kind="mono", outlier_frac=0.1)
    for b in range(2):
        n = f["n_edges"][b]
        E = f["edges"][b, :n]
        pose, outl, ngood = O.pose_optimize(f["poses"][b], E, f["cameras"])
        assert ngood == n - outl.sum()
        R = _quat_to_rot(pose[3:]); t = pose[:3]
        Xc = E["xw"].astype(np.float64) @ R.T + t
        p = f["cameras"][0]["p"]
        uv = np.stack([p[0] * Xc[:, 0] / Xc[:, 2] + p[2], p[1] * Xc[:, 1] / Xc[:, 2] + p[3]], 1)
        chi = ((uv - E["obs"][:, :2]) ** 2).sum(1) * E["inv_sigma2"]
        assert (chi[outl == 0] <= 5.991 * 1.0001).all()
        assert 0.05 * n < outl.sum() < 0.2 * n
        # start pose was off by centimetres / a degree: the initial chi2 is far above the final
        R0 = _quat_to_rot(f["poses"][b, 3:]); X0 = E["xw"].astype(np.float64) @ R0.T + f["poses"][b, :3]
        uv0 = np.stack([p[0] * X0[:, 0] / X0[:, 2] + p[2], p[1] * X0[:, 1] / X0[:, 2] + p[3]], 1)
        chi0 = ((uv0 - E["obs"][:, :2]) ** 2).sum(1) * E["inv_sigma2"]
        assert np.median(chi0) > 5 * np.median(chi[outl == 0])


def test_oracle_pose_optimization_degenerate():
    f = synth_pose_frames(seed=1, batch=1, n_pts=50, kind="mono")
    pose, outl, n = O.pose_optimize(f["poses"][0], f["edges"][0, :2], f["cameras"])   # < 3 correspondences -> 0, pose untouched
    assert n == 0 and (pose == f["poses"][0]).all()
    pose, outl, n = O.pose_optimize(f["poses"][0], f["edges"][0, :8], f["cameras"])   # < 10 edges: one round only (Optimizer.cc:1248)
    assert 0 < n <= 8


CASES = [("mono", 0, 300), ("stereo", 1, 300), ("body", 2, 300), ("mono", 5, 1000), ("stereo", 7, 40)]


@pytest.mark.parametrize("kind,seed,n_pts", CASES[:3] + CASES[4:])
def test_emu_pose_optimization_matches_oracle(emu_lib, kind, seed, n_pts):
    check(emu_lib, "emu", kind, seed, n_pts, batch=4, tol=KB8_TOL if kind == "body" else 1e-7)


def test_emu_pose_optimization_one_wave_per_frame():
    """POSE_T_FEW=64: the one-wave-per-frame instantiation that batches of more than 256 frames take (the default build gives the small test
    batches four waves per frame)."""
    import ctypes
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("POSE_T_FEW=64",), tag="posewave")))
    check(lib, "emu", "stereo", 7, 40, batch=4, tol=1e-7)
    check(lib, "emu", "mono", 0, 300, batch=2, tol=1e-7)


def test_emu_pose_optimization_small_frames(emu_lib):
    f = synth_pose_frames(seed=9, batch=4, n_pts=12, kind="mono", outlier_frac=0.0)
    f["n_edges"][:] = [2, 3, 9, 12]
    out, outl, ng = run(emu_lib, "emu", f)
    assert ng[0] == 0 and (out[0] == f["poses"][0]).all()
    for b, (po, oo, no) in enumerate(oracle_batch(f)):
        assert ng[b] == no and (outl[b, :f["n_edges"][b]] == oo).all() and np.abs(out[b] - po).max() < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("kind,seed,n_pts", CASES)
def test_hip_pose_optimization_matches_oracle(hip_lib, kind, seed, n_pts):
    check(hip_lib, "hip", kind, seed, n_pts, batch=8, tol=KB8_TOL if kind == "body" else 1e-7)


@pytest.mark.gpu
def test_hip_pose_optimization_many_frames(hip_lib):
    """512 frames per launch (the bench shape); spot-check 16 against the oracle, and determinism across two launches."""
    f = synth_pose_frames(seed=11, batch=512, n_pts=500, kind="stereo")
    out, outl, ng = run(hip_lib, "hip", f)
    out2, outl2, ng2 = run(hip_lib, "hip", f)
    assert (out == out2).all() and (outl == outl2).all() and (ng == ng2).all()
    for b in range(0, 512, 32):
        po, oo, no = O.pose_optimize(f["poses"][b], f["edges"][b, :f["n_edges"][b]], f["cameras"])
        assert ng[b] == no and (outl[b, :f["n_edges"][b]] == oo).all() and np.abs(out[b] - po).max() < 1e-7
