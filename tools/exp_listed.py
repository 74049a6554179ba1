"""Runs ON THE GPU BOX: the two-pass decision of k_fast over a few consecutive batch calls (python tools/exp_listed.py WxH nfeatures batch):
per call the mode it ran in and the listed share the handle holds afterwards."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbhip
from orbhip.synth import synth_image
W, H = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "752x480").split("x"))
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
base = np.stack([synth_image(i, W, H) for i in range(16)])
frames = torch.from_numpy(np.ascontiguousarray(base[np.arange(B) % 16])).cuda()
e = orbhip.ORBextractor(nf, 1.2, 8, 20, 7)
for it in range(20):
    e.extract_batch(frames, (0, 0))
    torch.cuda.synchronize()
    p = e.last_fast_passes()
    t = e.last_timing()
    print("%dx%d call %2d  two_pass=%d  listed %6d of %6d (%.3f)  k_fast %.3f ms" % (W, H, it, p["two_pass"], p["listed"], p["tiles"], p["listed"] / max(p["tiles"], 1), t["fast"]))
