import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_lib as O
from orbhip.lba import LbaWindows, synth_window, HUBER_MONO, HUBER_STEREO
td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for nkf, nfix, huber, its in ((200, 1, (0.0, 0.0), 3), (200, 1, (HUBER_MONO, HUBER_STEREO), 5), (260, 2, (0.0, 0.0), 3)):
    w, cams = synth_window(41, nkf, nfix, 2500, 8, "stereo")
    L = LbaWindows([w], cams, td, huber=huber)
    st = L.optimize(its)
    op, ox, ost = O.lba_optimize(w, cams, huber, its)
    P = L.d["poses"].cpu().numpy()[0, :nkf]
    d = np.abs(P - op)
    print(nkf, nfix, huber[0] > 0, "stats", st[0], "oracle", ost, "max pose diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "median", np.median(d), "free", int((w["pose_hidx"] >= 0).sum()))
