#!/usr/bin/env python3
"""Runs ON THE GPU BOX: Optimizer::PoseOptimization latency for ONE frame per call (what Tracking sees: 2-3 calls per frame), and the batched rate."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from orbhip.lba import pose_optimization, synth_pose_frames  # noqa: E402

dev = torch.device("cuda:0")
for kind, n_pts in (("mono", 300), ("stereo", 400), ("body", 300)):
    pf = synth_pose_frames(seed=3, batch=4, n_pts=n_pts, kind=kind)
    P = torch.from_numpy(pf["poses"][:1].copy()).to(dev)
    E = torch.from_numpy(np.ascontiguousarray(pf["edges"][:1]).view(np.uint8).reshape(1, -1)).to(dev)
    N = torch.from_numpy(pf["n_edges"][:1].copy()).to(dev)
    C = torch.from_numpy(np.ascontiguousarray(pf["cameras"]).view(np.uint8)).to(dev)
    ph = kind != "body"
    pose_optimization(P, E, N, C, pinhole=ph)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200):
        out = pose_optimization(P, E, N, C, pinhole=ph)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 200
    print("pose_optimization 1 frame (%s, %d edges): %.1f us per call incl. launch + sync, %d inliers" % (kind, int(N[0]), dt * 1e6, int(out[2][0])))
