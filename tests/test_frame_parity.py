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
    for b in range(B):
        n = counts[b]
        ou = O.undistort_keypoints(kps[b, :n], camera.as_array())
        assert np.array_equal(un[b, :n].view(np.uint8), ou.view(np.float32).reshape(n, 7).view(np.uint8)), b
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
