#!/usr/bin/env python3
"""Runs ON THE GPU BOX with ORBHIP_LIB pointing at a -DORBX_PROF build (tools/build_variants.sh "-DORBX_PROF"): per-phase s_memtime sums of
k_fast / k_describe over one batch of the headline workload, as fractions of each kernel's summed wave time."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import orbhip  # noqa: E402
from orbhip import _lib  # noqa: E402

L = _lib.load()
B = 512
frames = bench.make_batch(B)
d = torch.from_numpy(frames).cuda()
ex = orbhip.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_batch=B)
out = ex.extract_batch(d, (0, 1000))
buf = (C.c_ulonglong * 32)()
L.orbx_debug_prof(buf, 1)
out = ex.extract_batch(d, (0, 1000), out=out)
L.orbx_debug_prof(buf, 1)
v = np.array(list(buf), np.float64)[:16].reshape(2, 8)
names = [["prologue+stage issue", "staging wait (barrier)", "stage1+compaction+stage2", "stage3 score", "barrier", "NMS+retry+list", "emit"],
         ["record+counts", "patch loads->LDS", "barrier1", "row reads+IC_Angle+row pass", "barrier2", "trig+column pass", "barrier3", "rBRIEF+outputs"]]
for k, kn in enumerate(("k_fast", "k_describe")):
    tot = v[k].sum()
    print(kn, "sum of wave time (ticks): %.3e" % tot, ex.last_timing())
    for i, n in enumerate(names[k]):
        print("   %-28s %5.1f %%" % (n, 100 * v[k][i] / tot))
