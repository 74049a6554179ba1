import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, orbhip
from orbhip.synth import synth_image
e = orbhip.ORBextractor(1000, 1.2, 8, 20, 7)
mono, k, d = e(synth_image(0), None, (0, 1000))
print("OK", mono, len(k))
