#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the LM fuzz of tests/test_lba_parity.py (random windows of every edge family, three per call, optimize(2..5) against the oracle's LM) over
a seed range, with every deviation COLLECTED instead of asserted:  python tools/fuzz_lm.py [n] [seed0]
Prints: windows whose iteration / lambda-trial counts differ from the oracle's, the worst relative chi2 difference among well-posed windows and among the
ill-conditioned ones (more than 2 x its lambda trials: nearly singular reduced systems), the worst pose difference where poses are compared."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")):
    sys.path.insert(0, p)
import oracle_lib as O  # noqa: E402
import test_lba_parity as T  # noqa: E402
from orbhip import _lib  # noqa: E402
from orbhip.lba import LbaWindows, synth_window  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
lib = _lib.load()
rng = np.random.default_rng(2024)
nwin = 0
count_diff, hard = [], []
worst = {"chi2_well": 0.0, "chi2_hard": 0.0, "pose_pin": 0.0, "pose_fish": 0.0}
for seed in range(s0, s0 + n):
    kind = ("mono", "stereo", "mixed", "body", "kb8")[seed % 5]
    ws, cams = [], None
    for j in range(3):
        nfree, nfix = int(rng.integers(1, 24)), int(rng.integers(1, 7))
        nfix = max(nfix, 3 - nfree, 2 if kind in ("mono", "kb8") else 1)
        w, cams = synth_window(1000 + 7 * seed + j, nfree + nfix, nfix, int(rng.integers(40, 400)), min(int(rng.integers(3, 9)), nfree + nfix), kind)
        ws.append(w)
    L = LbaWindows(ws, cams, T.to_dev("hip"), lib=lib, huber=T.HUBER)
    its = int(rng.integers(2, 6))
    stats = L.optimize(its)
    poses = T.to_host(L.d["poses"])
    for b, w in enumerate(ws):
        nwin += 1
        op, ox, ost = O.lba_optimize(w, cams, T.HUBER, its)
        rel = abs(stats[b, 1] - ost[1]) / max(ost[1], 1e-9)
        if max(ost[3], stats[b, 3]) > 2 * its:
            hard.append((seed, b, kind, int(stats[b, 3]), int(ost[3]), rel))
            worst["chi2_hard"] = max(worst["chi2_hard"], rel)
            continue
        if stats[b, 0] != ost[0] or stats[b, 3] != ost[3]:
            count_diff.append((seed, b, kind, stats[b].tolist(), list(ost)))
            continue
        worst["chi2_well"] = max(worst["chi2_well"], rel)
        fish = kind in ("body", "kb8", "mixed")
        obs = np.bincount(w["edges"]["pose"], minlength=len(w["poses"]))[w["pose_hidx"] >= 0]
        if obs.min() >= 40:
            dp = float(np.abs(poses[b, :len(w["poses"])] - op).max())
            worst["pose_fish" if fish else "pose_pin"] = max(worst["pose_fish" if fish else "pose_pin"], dp)
print("lm fuzz: %d windows (seeds %d .. %d); iteration / trial counts differ in %d well-posed windows: %s" % (nwin, s0, s0 + n - 1, len(count_diff), count_diff[:4]))
print("   worst relative chi2 difference, well-posed: %.2e; worst pose difference: pinhole %.2e, fisheye %.2e" % (worst["chi2_well"], worst["pose_pin"], worst["pose_fish"]))
print("   ill-conditioned windows (> 2 x its lambda trials): %d, worst relative chi2 difference %.2e; above 2e-4: %s"
      % (len(hard), worst["chi2_hard"], [h for h in hard if h[5] > 2e-4]))
