#!/usr/bin/env python3
"""Runs ON THE GPU BOX with ORBHIP_LIB pointing at a -DORBX_PROF build: per-phase s_memtime sums of k_octree for ONE frame per call (the
single-frame instantiation, 1 024 threads per (frame, level) problem), as fractions of the kernel's summed wave time."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orbhip  # noqa: E402
from orbhip import _lib  # noqa: E402
from orbhip.synth import synth_image  # noqa: E402

L = _lib.load()
d = torch.from_numpy(synth_image(5, 752, 480)[None]).cuda()
ex = orbhip.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_batch=1)
out = ex.extract_batch(d, (0, 1000))
buf = (C.c_ulonglong * 32)()
L.orbx_debug_prof(buf, 1)
for _ in range(20):
    out = ex.extract_batch(d, (0, 1000), out=out)
L.orbx_debug_prof(buf, 1)
v = np.array(list(buf), np.float64)[:16].reshape(2, 8)[1]
names = ["roots + key assignment", "expandable scan + list", "first-round child counts", "sorted rounds: rank + cut", "step-4 scans", "next list built", "key move",
         "best key per node + outputs"]
print("k_octree (one frame per call) sum of wave time (ticks): %.3e" % v.sum(), ex.last_timing())
for n, x in zip(names, v):
    print("   %-30s %5.1f %%" % (n, 100 * x / v.sum()))
