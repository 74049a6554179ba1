#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the ORACLE (the reference itself cannot be built or imported here and ships no golden
vectors — SURVEY.md §4, §8(c)).  The fixtures freeze the oracle's outputs so that any later change of the
oracle (or of the synthetic generators) is visible; tests/test_golden.py checks oracle == fixtures on CPU and the HIP path ==
fixtures on the GPU.  Data only: inputs are regenerated from seeds, expected outputs are stored."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle_lib as O  # noqa: E402
from orbhip.lba import HUBER_MONO, HUBER_STEREO, synth_window  # noqa: E402
from orbhip.synth import synth_image  # noqa: E402


def extractor_case(seed, W, H, nf, lap):
    img = synth_image(seed, W, H, n_rect=90, n_disc=45)
    o = O.OrbOracle(nf, 1.2, 8, 20, 7)
    mono, k, d = o.extract(img, *lap)
    return dict(seed=seed, W=W, H=H, nf=nf, lap=np.array(lap), mono=mono, kps=k.view(np.uint8).reshape(len(k), 28), desc=d,
                img_sha=np.frombuffer(__import__("hashlib").sha256(img.tobytes()).digest(), np.uint8),
                cand_counts=np.array([len(o.level_candidates(l)) for l in range(8)]))


def matcher_scene():
    # matcher: frame B = frame A moved by (6,-4); queries = A's keypoints projected into B (test_matcher_parity.scene).
    # kp_match codes: >= 0 query index, -1 untouched, -2 claimed during the call and reset to NULL by the orientation cull (ORBmatcher.cc:2499)
    import test_matcher_parity as T
    S = T.scene()
    out = {}
    for name, mode, th, ratio, ori in (("motion", 1, 15, 0.9, True), ("local", 0, 5, 0.8, True)):
        q = T.make_queries(S, mode, th, np.random.default_rng(0))
        qm, km, n = O.search_by_projection(S["kb"], S["db"], q, S["da"], S["grid"], mode, 100, ratio, ori)
        out[name + "_kp_match"] = km
        out[name + "_q_match"] = qm
        out[name + "_n"] = n
    np.savez_compressed(os.path.join(HERE, "matcher_scene.npz"), **out)


def main():
    if sys.argv[1:] == ["matcher"]:   # regenerate one fixture only (npz bytes carry zip timestamps: an untouched fixture stays untouched in git)
        return matcher_scene()
    np.savez_compressed(os.path.join(HERE, "extract_320x240.npz"), **extractor_case(1, 320, 240, 300, (0, 1000)))
    np.savez_compressed(os.path.join(HERE, "extract_400x300_lap.npz"), **extractor_case(2, 400, 300, 400, (120, 260)))
    matcher_scene()
    # LBA: 12-KF / 300-point mixed window: per-block checksums
    w, cams = synth_window(0, 12, 3, 300, 6, "mixed")
    o = O.lba_build_system(w, cams, (HUBER_MONO, HUBER_STEREO))
    np.savez_compressed(os.path.join(HERE, "lba_mixed_12kf.npz"), n_edges=len(w["edges"]), Hpp=o["Hpp"], bp=o["bp"], Hll_sum=o["Hll"].sum(0),
                        bl_sum=o["bl"].sum(0), Hpl_sum=o["Hpl"].sum(0), chi2=o["chi2"], robust=o["robust_chi2_sum"])
    # Frame glue: cv::undistortPoints restatement on a grid of pixel positions, EuRoC calibration (Examples/Monocular/EuRoC.yaml:9-17)
    from orbhip.frame import Camera
    cam = Camera.make(458.654, 457.296, 367.215, 248.375, (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)).as_array()
    gx, gy = np.meshgrid(np.arange(0, 752, 47, dtype=np.float32), np.arange(0, 480, 43, dtype=np.float32))
    k = np.zeros(gx.size, O.KP_DTYPE); k["x"], k["y"] = gx.ravel(), gy.ravel()
    u = O.undistort_keypoints(k, cam)
    np.savez_compressed(os.path.join(HERE, "undistort_euroc.npz"), cam=cam, xy=np.stack([k["x"], k["y"]], 1), xy_un=np.stack([u["x"], u["y"]], 1),
                        bounds=O.image_bounds(cam, 752, 480))
    # visual-inertial local BA: seed-7 stereo window, optimize(1.0, 6): statistics + final states
    from orbhip.inertial import synth_inertial_window
    iw = synth_inertial_window(7, n_opt=6, n_fixed_vis=2, n_pts=250, max_obs=6, kind="stereo")
    kf, pts, st = O.inertial_optimize(iw, (HUBER_MONO, HUBER_STEREO), 1.0, 6)
    e0 = O.inertial_errors(iw, (HUBER_MONO, HUBER_STEREO))
    np.savez_compressed(os.path.join(HERE, "inertial_stereo_6kf.npz"), n_edges=len(iw["edges"]), stats=st, twb=kf["twb"], Rwb=kf["Rwb"], v=kf["v"], bg=kf["bg"],
                        ba=kf["ba"], points_sum=pts.sum(0), imu_chi2_0=e0["imu_chi2"], robust0=e0["robust_chi2_sum"])
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
