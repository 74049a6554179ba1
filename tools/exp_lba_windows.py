#!/usr/bin/env python3
"""Runs ON THE GPU BOX: BlockSolver::buildSystem-equivalent linearisations per second against the number of C5-size windows per launch."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbhip.lba import LbaWindows, synth_window  # noqa: E402

dev = torch.device("cuda", 0)
wins, cams = [], None
for i in range(2):
    w, cams = synth_window(100 + i, 100, 20, 20000, 8, "mono")
    wins.append(w)
outs = ("Hpp", "bp", "Hll", "bl", "Hpl", "chi2")
for nwin in (4, 16, 64, 256):
    Lw = LbaWindows([wins[i % 2] for i in range(nwin)], cams, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))
    for _ in range(2):
        Lw.build_system(outs)
    torch.cuda.synchronize()
    n = 10
    t = time.perf_counter()
    for _ in range(n):
        Lw.build_system(outs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("windows per launch %4d: %.3f ms per launch, %.0f linearisations/s" % (nwin, dt * 1e3, nwin / dt))
    del Lw
    torch.cuda.empty_cache()
