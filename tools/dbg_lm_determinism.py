import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from orbhip.lba import LbaWindows, synth_window
dev = torch.device("cuda:0")
w, cams = synth_window(5, 100, 20, 20000, 8, "mono")
res = []
for r in range(4):
    L = LbaWindows([w] * 8, cams, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))
    L.optimize(5)
    res.append(L.d["poses"].cpu().numpy().copy())
print("window-to-window identical:", all(np.array_equal(res[0][0], res[0][i]) for i in range(8)))
print("run-to-run identical:", all(np.array_equal(res[0], r) for r in res[1:]), "max diff", max(np.abs(res[0] - r).max() for r in res[1:]))
