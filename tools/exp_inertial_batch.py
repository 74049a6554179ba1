#!/usr/bin/env python3
"""Runs ON THE GPU BOX: LocalInertialBA throughput vs windows per launch (experiment)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orbhip.inertial import InertialWindows, synth_inertial_window
dev = torch.device("cuda:0")
iw = [synth_inertial_window(60 + i, n_opt=10, n_fixed_vis=6, n_pts=1200, max_obs=8, kind="stereo") for i in range(2)]
td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for IB in (256, 512, 1024, 2048):
    W = InertialWindows([iw[i % 2] for i in range(IB)], td)
    kf0, pt0 = W.d["kfs"].clone(), W.d["points"].clone()
    W.optimize(1.0, 10); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2):
        W.d["kfs"].copy_(kf0); W.d["points"].copy_(pt0)
        st = W.optimize(1.0, 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 2
    print(IB, "windows: %.1f ms -> %.0f windows/s, %.0f LM it/s" % (dt * 1e3, IB / dt, float(st[:, 0].sum().item()) / dt))
