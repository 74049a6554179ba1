import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, orbhip, oracle_lib as O
from orbhip.extractor import stereo_matches
from test_stereo_parity import stereo_pair, BF, FX
left, right = stereo_pair(3)
oL, oR = O.OrbOracle(600, 1.2, 8, 20, 7), O.OrbOracle(600, 1.2, 8, 20, 7)
_, kl, dl = oL.extract(left, 0, 0); _, kr, dr = oR.extract(right, 0, 0)
mb = BF / FX
our, odp = O.stereo_matches(oL, oR, kl, dl, kr, dr, mb, BF)
eL = orbhip.ORBextractor(600, 1.2, 8, 20, 7); eR = orbhip.ORBextractor(600, 1.2, 8, 20, 7)
outL = eL.extract_batch(torch.from_numpy(left[None]).cuda(), (0, 0)); outR = eR.extract_batch(torch.from_numpy(right[None]).cuda(), (0, 0))
u, d = stereo_matches(eL, eR, outL, outR, mb, BF)
torch.cuda.synchronize()
ur, dp = u.cpu().numpy()[0, :len(kl)], d.cpu().numpy()[0, :len(kl)]
bad = np.nonzero(ur.view(np.uint32) != our.view(np.uint32))[0]
print("n", len(kl), "mismatch uR", len(bad), "depth", (dp.view(np.uint32) != odp.view(np.uint32)).sum(), "valid oracle", (our >= 0).sum(), "valid hip", (ur >= 0).sum())
for i in bad[:12]:
    print(i, kl[i]["x"], kl[i]["y"], kl[i]["octave"], "oracle", our[i], odp[i], "hip", ur[i], dp[i])
