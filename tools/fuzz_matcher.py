#!/usr/bin/env python3
"""Runs ON THE GPU BOX: randomised SearchByProjection parity (random scenes x random parameters, both modes, stereo gate / occupancy on or off)
beyond the 12 seeds of tests/test_matcher_parity.py.  usage: tools/fuzz_matcher.py [n_cases] [seed0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_matcher_parity as T  # noqa: E402
from orbhip import _lib  # noqa: E402
from orbhip.matcher import MODE_BEST_ONLY, MODE_LOCAL_MAP  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
lib = _lib.load()
bad = 0
for i in range(n):
    rng = np.random.default_rng(s0 + i)
    kw = dict(W=int(rng.integers(320, 900)), H=int(rng.integers(240, 600)), nf=int(rng.choice([300, 600, 1200])), seed=int(rng.integers(1000)),
              shift=(int(rng.integers(-9, 10)), int(rng.integers(-9, 10))))
    mode = MODE_BEST_ONLY if i % 2 else MODE_LOCAL_MAP
    try:
        T.check_sbp(lib, "hip", mode, int(rng.choice([1, 3, 7, 15, 30, 60])), float(rng.choice([0.6, 0.8, 0.9])), bool(rng.integers(2)),
                    stereo=bool(rng.integers(2)), occupied=bool(rng.integers(2)), seed=i, scene_kw=kw)
    except ValueError as e:   # the oracle refuses geometries on which the reference itself is undefined (portrait images, tiny levels)
        print("case", s0 + i, "skipped:", str(e)[:60])
    except Exception as e:   # noqa: PERF203,BLE001
        bad += 1
        print("case", s0 + i, kw, "MISMATCH", str(e)[:100])
print("matcher fuzz: %d cases, %d mismatches" % (n, bad))
