#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the single-frame drop-in call (ORBextractor::operator() through the host API: H2D image, kernels, D2H keypoints and
descriptors), for rocprofv3 kernel traces of the batch-1 path."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import orbhip  # noqa: E402
from orbhip.synth import synth_image  # noqa: E402

img = synth_image(5, 752, 480)
e = orbhip.ORBextractor(1000, 1.2, 8, 20, 7, device=0)
for _ in range(20):
    e(img, None, (0, 1000))
t = time.perf_counter()
N = 200
for _ in range(N):
    mono, k, d = e(img, None, (0, 1000))
dt = time.perf_counter() - t
print("single-frame extract: %.1f us per call (%d keypoints)" % (dt / N * 1e6, len(k)))
