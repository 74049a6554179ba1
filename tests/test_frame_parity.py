"""Frame-constructor steps between extractor and matcher (Frame::UndistortKeyPoints, ComputeImageBounds, ComputeStereoFromRGBD): HIP path vs
the oracle, bit-exact floats.  The oracle's cv::undistortPoints restatement is pinned to its defining property (distort(undistort(p)) == p to the
5-iteration fixed point's accuracy) and to hand-checked values of the EuRoC calibration (Examples/Monocular/EuRoC.yaml)."""
import numpy as np
import pytest

import oracle_lib as O
import orbhip
from orbhip import KP_DTYPE
from orbhip.frame import Camera, FrameOps

EUROC = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, dist=(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05))   # EuRoC.yaml:9-17
TUM1 = dict(fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, dist=(0.262383, -0.953104, -0.005358, 0.002628, 1.163314))  # TUM1.yaml


def keypoints(rng, n, W, H):
    k = np.zeros(n, KP_DTYPE)
    k["x"] = rng.integers(16, W - 16, n).astype(np.float32) * rng.choice([1.0, 1.2, 1.44], n).astype(np.float32)
    k["x"] = np.minimum(k["x"], W - 1)
    k["y"] = np.minimum(rng.integers(16, H - 16, n).astype(np.float32) * rng.choice([1.0, 1.2], n).astype(np.float32), H - 1)
    k["size"], k["angle"], k["response"] = 31.0, rng.uniform(0, 360, n), rng.integers(7, 200, n)
    k["octave"], k["class_id"] = rng.integers(0, 8, n), -1
    return k


def distort(cam, x, y):
    """Forward Brown-Conrady model in double (cv::projectPoints)."""
    k1, k2, p1, p2, k3 = (list(cam["dist"]) + [0.0])[:5]
    xn, yn = (x - np.float32(cam["cx"]).astype(np.float64)) / np.float64(np.float32(cam["fx"])), (y - np.float64(np.float32(cam["cy"]))) / np.float64(np.float32(cam["fy"]))
    r2 = xn * xn + yn * yn
    cd = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = xn * cd + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
    yd = yn * cd + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn
    return xd * np.float64(np.float32(cam["fx"])) + np.float64(np.float32(cam["cx"])), yd * np.float64(np.float32(cam["fy"])) + np.float64(np.float32(cam["cy"]))


@pytest.mark.parametrize("cam", [EUROC, TUM1], ids=["euroc", "tum1"])
def test_oracle_undistort_inverts_the_distortion_model(cam):
    rng = np.random.default_rng(0)
    W, H = (752, 480) if cam is EUROC else (640, 480)
    k = keypoints(rng, 2000, W, H)
    u = O.undistort_keypoints(k, Camera.make(**cam).as_array())
    xd, yd = distort(cam, u["x"].astype(np.float64), u["y"].astype(np.float64))
    # 5 fixed-point iterations (not run to convergence): ~1e-2 px or better over most of the image, a few tenths of a pixel in the far corners
    ex, ey = np.abs(xd - k["x"]), np.abs(yd - k["y"])
    assert ex.max() < 0.5 and ey.max() < 0.5 and np.median(ex) < 2e-2 and np.median(ey) < 2e-2
    assert np.abs(u["x"] - k["x"]).max() > 1.0           # it really moved points
    for f in ("size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(u[f], k[f])
    # principal point is a fixed point of the model
    c = np.zeros(1, KP_DTYPE); c["x"], c["y"] = np.float32(cam["cx"]), np.float32(cam["cy"])
    uc = O.undistort_keypoints(c, Camera.make(**cam).as_array())
    assert abs(uc["x"][0] - c["x"][0]) < 1e-4 and abs(uc["y"][0] - c["y"][0]) < 1e-4


def test_oracle_image_bounds_euroc():
    b = O.image_bounds(Camera.make(**EUROC).as_array(), 752, 480)
    # barrel distortion: the undistorted corners lie outside the image
    assert b[0] < -50 and b[1] > 752 + 50 and b[2] < -30 and b[3] > 480 + 30
    assert b[4] == np.float32(64) / np.float32(b[1] - b[0]) and b[5] == np.float32(48) / np.float32(b[3] - b[2])
    nb = O.image_bounds(Camera.make(458.654, 457.296, 367.215, 248.375).as_array(), 752, 480)
    assert list(nb[:4]) == [0.0, 752.0, 0.0, 480.0]


def to_dev(backend):
    if backend == "emu":
        return lambda a: a
    import torch
    return lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def to_host(a):
    return a if isinstance(a, np.ndarray) else a.cpu().numpy()


def check_frame_ops(lib, backend, cam, W, H):
    rng = np.random.default_rng(3)
    dev = to_dev(backend)
    camera = Camera.make(**cam)
    F = FrameOps(camera, W, H, lib=lib)
    ob = O.image_bounds(camera.as_array(), W, H)
    assert [F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY] == [float(v) for v in ob[:4]]
    assert F.grid == (float(ob[0]), float(ob[2]), float(ob[4]), float(ob[5]))
    B, cap = 3, 1200
    counts = np.array([1200, 0, 777], np.int32)
    kps = np.zeros((B, cap), KP_DTYPE)
    for b in range(B):
        kps[b, :counts[b]] = keypoints(rng, counts[b], W, H)
    kf = kps.view(np.float32).reshape(B, cap, 7)
    un = to_host(F.UndistortKeyPoints(dev(kf), dev(counts)))
    depth = rng.uniform(-0.5, 6.0, (B, H, W)).astype(np.float32)
    depth[rng.random((B, H, W)) < 0.2] = 0.0
    ur, dz = F.ComputeStereoFromRGBD(dev(kf), dev(un), dev(counts), dev(depth), 40.0)
    ur, dz = to_host(ur), to_host(dz)
    # the fused launch (orbm_undistort_and_grid_build) writes the same records and the same CSR grid as the two separate steps
    un2, gs2, gi2 = [to_host(x) for x in F.UndistortAndGrid(dev(kf), dev(counts))]
    m = orbhip.ORBmatcher(lib=lib)
    gs1, gi1 = [to_host(x) for x in m.grid_build(dev(un), dev(counts), F.grid)]
    for b in range(B):
        n = counts[b]
        ou = O.undistort_keypoints(kps[b, :n], camera.as_array())
        assert np.array_equal(un[b, :n].view(np.uint8), ou.view(np.float32).reshape(n, 7).view(np.uint8)), b
        assert np.array_equal(un2[b, :n].view(np.uint8), un[b, :n].view(np.uint8)), b
        ogs, ogi = O.grid_build(ou, F.grid)
        tot = int(ogs[-1])
        assert np.array_equal(gs2[b], ogs) and np.array_equal(gi2[b, :tot], ogi[:tot]), ("fused grid vs oracle", b)
        assert np.array_equal(gs1[b], ogs) and np.array_equal(gi1[b, :tot], ogi[:tot]), ("separate grid vs oracle", b)
        if n:
            assert 0 < tot <= n
        our, odz = O.stereo_from_rgbd(kps[b, :n], ou, depth[b], 40.0)
        assert np.array_equal(ur[b, :n], our) and np.array_equal(dz[b, :n], odz)
        assert (ur[b, n:] == -1).all() and (dz[b, n:] == -1).all()
        if n:
            assert (odz > 0).any() and (odz < 0).any()


@pytest.mark.parametrize("cam,size", [(EUROC, (752, 480)), (TUM1, (640, 480)), (dict(fx=500.0, fy=500.0, cx=320.0, cy=240.0), (640, 480))],
                         ids=["euroc", "tum1_k3", "no_distortion"])
def test_emu_frame_ops_match_oracle(emu_lib, cam, size):
    check_frame_ops(emu_lib, "emu", cam, *size)


@pytest.mark.gpu
@pytest.mark.parametrize("cam,size", [(EUROC, (752, 480)), (TUM1, (640, 480)), (dict(fx=500.0, fy=500.0, cx=320.0, cy=240.0), (640, 480))],
                         ids=["euroc", "tum1_k3", "no_distortion"])
def test_hip_frame_ops_match_oracle(hip_lib, cam, size):
    check_frame_ops(hip_lib, "hip", cam, *size)


def check_undistort_and_grid_edges(lib, backend):
    """orbm_undistort_and_grid_build on the shapes its kernel treats specially: in place, an empty frame, key points outside the grid (dropped by PosInGrid),
    a capacity beyond four key points per thread (the call falls back to the two separate launches), cap not a multiple of anything."""
    rng = np.random.default_rng(11)
    dev = to_dev(backend)
    camera = Camera.make(**EUROC)
    F = FrameOps(camera, 752, 480, lib=lib)
    for cap, counts in ((1063, [1063, 0, 1, 500]), (4100, [4100, 37]), (64, [64, 64])):
        B = len(counts)
        counts = np.array(counts, np.int32)
        kps = np.zeros((B, cap), KP_DTYPE)
        for b in range(B):
            k = keypoints(rng, counts[b], 752, 480)
            if counts[b] > 10:                            # a few far outside the image: no grid cell
                k["x"][:3] = [-500.0, 5000.0, 100.0]; k["y"][:3] = [100.0, 100.0, -900.0]
            kps[b, :counts[b]] = k
        kf = kps.view(np.float32).reshape(B, cap, 7).copy()
        buf = dev(kf.copy())
        un, gs, gi = F.UndistortAndGrid(buf, dev(counts), out=(buf, dev(np.zeros((B, 64 * 48 + 1), np.int32)), dev(np.zeros((B, cap), np.int32))))   # in place
        un, gs, gi = to_host(un), to_host(gs), to_host(gi)
        for b in range(B):
            n = counts[b]
            ou = O.undistort_keypoints(kps[b, :n], camera.as_array())
            assert np.array_equal(un[b, :n].view(np.uint8), ou.view(np.float32).reshape(n, 7).view(np.uint8)), (cap, b)
            ogs, ogi = O.grid_build(ou, F.grid)
            tot = int(ogs[-1])
            assert np.array_equal(gs[b], ogs) and np.array_equal(gi[b, :tot], ogi[:tot]), (cap, b)
            assert tot <= n and (n <= 10 or tot <= n - 3)


def test_emu_undistort_and_grid_edges(emu_lib):
    check_undistort_and_grid_edges(emu_lib, "emu")


@pytest.mark.gpu
def test_hip_undistort_and_grid_edges(hip_lib):
    check_undistort_and_grid_edges(hip_lib, "hip")


# ---- Frame::ComputeStereoFishEyeMatches (two KannalaBrandt8 cameras) ------------------------------------------------------------------
from orbhip.frame import ComputeStereoFishEyeMatches, FisheyeRig  # noqa: E402
from orbhip.lba import _kb8_project, _rodrigues  # noqa: E402

KB_L = np.float32([190.978, 190.973, 254.932, 256.897, 0.0034823894, 0.0007150348, -0.0020532361, 0.00020293673])   # TUM_512.yaml-like
KB_R = np.float32([190.442, 190.434, 252.597, 254.917, 0.0034003171, 0.0017662670, -0.0026630025, 0.00032995968])


def fisheye_frame(seed, n_stereo=260, n_mono_l=90, n_mono_r=70):
    """One fisheye stereo frame: 3-D points seen by both cameras (lapping area) + monocular-only keypoints in front of them in the arrays."""
    rng = np.random.default_rng(seed)
    R_lr = _rodrigues(np.array([0.004, -0.012, 0.003])).astype(np.float32); t_lr = np.float32([0.101, -0.0011, 0.0006])   # x_l = R_lr x_r + t_lr
    X = np.stack([rng.uniform(-3, 3, n_stereo), rng.uniform(-2, 2, n_stereo), rng.uniform(1.2, 9, n_stereo)], 1)     # left-camera frame
    Xr = (X - t_lr) @ R_lr.astype(np.float64)                                                                        # R_lr^T (x - t)
    ul, vl, _ = _kb8_project(KB_L.astype(np.float64), X); ur, vr, _ = _kb8_project(KB_R.astype(np.float64), Xr)

    def cam(u, v, n_mono):
        n = n_mono + len(u)
        k = np.zeros(n, KP_DTYPE)
        k["x"][:n_mono] = rng.uniform(20, 490, n_mono); k["y"][:n_mono] = rng.uniform(20, 490, n_mono)
        k["x"][n_mono:] = u + rng.normal(0, 0.4, len(u)); k["y"][n_mono:] = v + rng.normal(0, 0.4, len(u))
        k["octave"] = rng.integers(0, 8, n); k["size"], k["angle"], k["response"], k["class_id"] = 31, rng.uniform(0, 360, n), 50, -1
        return k
    kl, kr = cam(ul, vl, n_mono_l), cam(ur, vr, n_mono_r)
    base = rng.integers(0, 256, (n_stereo, 32), dtype=np.uint8)
    dl = np.concatenate([rng.integers(0, 256, (n_mono_l, 32), dtype=np.uint8), base])
    flips = np.zeros((n_stereo, 256), np.uint8)
    for i in range(n_stereo):
        flips[i, rng.choice(256, int(rng.integers(0, 40)), replace=False)] = 1
    dr = np.concatenate([rng.integers(0, 256, (n_mono_r, 32), dtype=np.uint8), base ^ np.packbits(flips, axis=1, bitorder="little")])
    # a few gross geometric outliers (wrong right position) and near-duplicate descriptors (fail the ratio test)
    nb = min(25, n_stereo // 4)
    bad = rng.choice(n_stereo, nb, replace=False)
    kr["x"][n_mono_r + bad[:nb // 2]] += rng.uniform(15, 60, nb // 2)
    dr[n_mono_r + bad[nb // 2:]] = dr[n_mono_r + (bad[nb // 2:] + 1) % n_stereo]
    perm = rng.permutation(n_stereo)                       # right-camera order is unrelated to the left one
    kr[n_mono_r:] = kr[n_mono_r:][perm]; dr[n_mono_r:] = dr[n_mono_r:][perm]
    sig2 = (np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2
    return kl, dl, n_mono_l, kr, dr, n_mono_r, FisheyeRig.make(KB_L, KB_R, R_lr, t_lr, sig2), sig2


def check_fisheye(lib, backend):
    dev = to_dev(backend)
    frames = [fisheye_frame(50), fisheye_frame(51, n_stereo=120, n_mono_l=0, n_mono_r=200), fisheye_frame(52, n_stereo=1, n_mono_l=5, n_mono_r=5)]
    rig = frames[0][6]
    B, capL, capR = len(frames), max(len(f[0]) for f in frames) + 3, max(len(f[3]) for f in frames) + 7
    kl = np.zeros((B, capL), KP_DTYPE); kr = np.zeros((B, capR), KP_DTYPE); dl = np.zeros((B, capL, 32), np.uint8); dr = np.zeros((B, capR, 32), np.uint8)
    n = np.zeros((4, B), np.int32)
    for b, f in enumerate(frames):
        kl[b, :len(f[0])], dl[b, :len(f[0])], kr[b, :len(f[3])], dr[b, :len(f[3])] = f[0], f[1], f[3], f[4]
        n[:, b] = len(f[0]), f[2], len(f[3]), f[5]
    v7 = lambda k: k.view(np.float32).reshape(B, -1, 7)
    out = ComputeStereoFishEyeMatches(dev(v7(kl)), dev(dl), dev(n[0].copy()), dev(n[1].copy()), dev(v7(kr)), dev(dr), dev(n[2].copy()), dev(n[3].copy()), rig, lib=lib)
    l2r, r2l, depth, p3d, nm = [to_host(o) for o in out]
    for b, f in enumerate(frames):
        ol2r, or2l, od, op, on = O.stereo_fisheye(f[0], f[1], f[2], f[3], f[4], f[5], rig.as_array(), f[7])
        nl, nr = len(f[0]), len(f[3])
        assert nm[b] == on and np.array_equal(l2r[b, :nl], ol2r) and np.array_equal(r2l[b, :nr], or2l), b
        # rule R4 leaves libm's double atan2 / tan / cos as the only platform dependence: identical decisions, floats to ~1 ulp
        assert np.allclose(depth[b, :nl], od, rtol=2e-6, atol=0) and np.allclose(p3d[b, :nl], op, rtol=2e-6, atol=1e-7), b
        assert (l2r[b, nl:] == -1).all() and (depth[b, nl:] == -1).all()
    ol2r, _, od, op, on = O.stereo_fisheye(*frames[0][:6], rig.as_array(), frames[0][7])
    # planted pairs nearer than ~5 m are recovered (baseline 0.1 m: beyond that the cos-parallax gate 0.9998 refuses them); the displaced and
    # the ambiguous ones are rejected; monocular keypoints never match
    assert 60 < on < 200 and (ol2r[:frames[0][2]] == -1).all()
    m = ol2r >= 0
    assert np.median(np.abs(od[m] - op[m, 2])) == 0 and od[m].min() > 1.0 and od[m].max() < 5.5


def test_emu_stereo_fisheye_matches(emu_lib):
    check_fisheye(emu_lib, "emu")


@pytest.mark.gpu
def test_hip_stereo_fisheye_matches(hip_lib):
    check_fisheye(hip_lib, "hip")
