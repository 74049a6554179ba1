#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the LM leg with T host threads x (W / T) windows on T HIP streams (tools/exp_lm.py is the T = 1 case)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from orbhip.lba import LbaWindows, synth_window  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
wins, cams = [], None
for i in range(2):
    w, cams = synth_window(100 + i, 100, 20, 20000, 8, "mono")
    wins.append(w)
for T in (1, 2, 4):
    per = W // T
    streams = [torch.cuda.Stream(dev) for _ in range(T)]
    Ls = [LbaWindows([wins[i % 2] for i in range(per)], cams, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)) for _ in range(T)]
    p0 = [(L.d["poses"].clone(), L.d["points"].clone()) for L in Ls]
    res = [None] * T

    def run(k, reps):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[k]):
            for _ in range(reps):
                Ls[k].d["poses"].copy_(p0[k][0]); Ls[k].d["points"].copy_(p0[k][1])
                res[k] = Ls[k].optimize(5)
            streams[k].synchronize()
    for reps in (1, 2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        th = [threading.Thread(target=run, args=(k, reps)) for k in range(T)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    its = sum(float(r[:, 0].sum()) for r in res) * 2
    print("T=%d threads x %d windows: %.1f LM iterations/s (%.2f ms per round of optimize(5))" % (T, per, its / dt, dt / 2 * 1e3))
    del Ls, p0
    torch.cuda.empty_cache()
