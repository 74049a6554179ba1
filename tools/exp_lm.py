#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the full-LM leg of bench.py on its own (W C5-size windows, optimize(5) per batch), for rocprofv3 kernel traces of the
LM kernels.  Prints LM iterations/s and the mean lambda-trial count."""
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
ap.add_argument("--windows", type=int, default=256)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--kf", type=int, default=100)
ap.add_argument("--fixed", type=int, default=20)
ap.add_argument("--points", type=int, default=20000)
ap.add_argument("--kind", default="mono", help="mono | stereo | kb8 | body | mixed (synth_window)")
ap.add_argument("--sorted", action="store_true", help="order every landmark's observations by pose index (what std::map<KeyFrame*> iteration "
                "gives when key frames are allocated in id order) instead of synth_window's random order")
args = ap.parse_args()
dev = torch.device("cuda:0")
wins, cams = [], None
for i in range(2):
    w, cams = synth_window(100 + i, args.kf, args.fixed, args.points, 8, args.kind)
    if args.sorted:
        o = np.lexsort((w["edges"]["pose"], w["edges"]["point"]))
        assert (w["edges"]["point"][o] == w["edges"]["point"]).all()   # still landmark-major
        w["edges"] = w["edges"][o]
    wins.append(w)
Lw = LbaWindows([wins[i % 2] for i in range(args.windows)], cams, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))
p0, x0 = Lw.d["poses"].clone(), Lw.d["points"].clone()
Lw.optimize(5)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(args.reps):
    Lw.d["poses"].copy_(p0); Lw.d["points"].copy_(x0)
    st = Lw.optimize(5)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("lm_iterations_per_s %.1f  ms_per_optimize5_batch %.2f  trials/window %.2f  chi2[0] %.6f" %
      (float(st[:, 0].sum()) * args.reps / dt, dt / args.reps * 1e3, float(st[:, 3].mean()), float(st[0, 1])))

try:
    import ctypes as C
    from orbhip import _lib
    L = _lib.load()
    buf = (C.c_ulonglong * 8)()
    L.lba_debug_chol_prof(buf, 1)
    v = np.array(list(buf), np.float64)
    names = ["panel load + barrier", "diagonal block (wave 0) + barrier", "panel rows + barrier", "panel store", "trailing update", "end barrier", "backward substitution"]
    print("k_lm_chol phases of workgroup thread 0 (share of the summed ticks):")
    for i, n in enumerate(names):
        print("   %-36s %5.1f %%" % (n, 100 * v[i] / v[:7].sum()))
except AttributeError:
    pass
