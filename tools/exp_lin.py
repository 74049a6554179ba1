#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the linearisation leg of bench.py on its own (W C5-size windows, lba_build_system), for rocprofv3 kernel traces."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from orbhip.lba import LbaWindows, synth_window  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=16)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--kind", default="mono")
args = ap.parse_args()
dev = torch.device("cuda:0")
wins, cams = [], None
for i in range(2):
    w, cams = synth_window(100 + i, 100, 20, 20000, 8, args.kind)
    wins.append(w)
Lw = LbaWindows([wins[i % 2] for i in range(args.windows)], cams, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))
outs = ("Hpp", "bp", "Hll", "bl", "Hpl", "chi2")
for _ in range(3):
    Lw.build_system(outs)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(args.reps):
    Lw.build_system(outs)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("linearizations_per_s %.1f  ms_per_step %.4f" % (args.windows * args.reps / dt, dt / args.reps * 1e3))
