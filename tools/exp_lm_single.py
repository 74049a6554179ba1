#!/usr/bin/env python3
"""Runs ON THE GPU BOX: one C5-size window through LbaWindows.optimize(5), timed; with ORBHIP_LM_TRACE=1 only two calls (for a rocprofv3 kernel trace)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbhip.lba import LbaWindows, synth_window  # noqa: E402

dev = torch.device("cuda", 0)
KF, FIXED, PTS = [int(v) for v in (os.environ.get("ORBHIP_LM_SHAPE", "100,20,20000").split(","))]     # key frames, of which fixed, landmarks
w, cams = synth_window(100, KF, FIXED, PTS, 8, "mono")
L1 = LbaWindows([w], cams, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))
q0, y0 = L1.d["poses"].clone(), L1.d["points"].clone()
L1.optimize(5)
torch.cuda.synchronize()
n = 2 if os.environ.get("ORBHIP_LM_TRACE") else 10
t = time.perf_counter()
for _ in range(n):
    L1.d["poses"].copy_(q0); L1.d["points"].copy_(y0)
    st = L1.optimize(5)
torch.cuda.synchronize()
print("%d key frames (%d fixed), %d landmarks:" % (KF, FIXED, PTS), "ms per optimize(5):", round((time.perf_counter() - t) / n * 1e3, 3), "iterations", st[0, 0], "trials", st[0, 3])

try:   # -DCHOL_PROF builds only: where wave 0 of k_lm_chol_step spends its ticks
    import ctypes as C
    from orbhip import _lib
    L = _lib.load()
    buf = (C.c_ulonglong * 8)()
    L.lba_debug_chol_prof(buf, 1)
    v = np.array(list(buf), np.float64)
    names = ["operand + block loads", "diagonal block + inverse chain", "X to LDS, y", "L21 = A21 X^T (matrix core)", "tile update + stores"]
    print("k_lm_chol_step, wave 0 (share of the summed ticks; total %.0f ticks):" % v[:5].sum())
    for i, nm in enumerate(names):
        print("   %-36s %5.1f %%" % (nm, 100 * v[i] / max(1.0, v[:5].sum())))
except AttributeError:
    pass
