"""M9 parity: Frame::ComputeStereoMatches (reference src/Frame.cc:955-1133) — HIP vs the oracle, bit-exact floats.
Also M7 / M8: the relocalisation and Sim3 projection searches as parameterisations of the generic windowed search."""
import numpy as np
import pytest

import oracle_lib as O
import orbhip
from orbhip.extractor import stereo_matches, stereo_matches_host
from orbhip.synth import synth_image

BF, FX = 47.90639384423901, 458.654   # EuRoC: Camera.bf, fx (Examples/ROS/ORB_SLAM3/EuRoC.yaml:26, Examples/Monocular/EuRoC.yaml:9)


def stereo_pair(seed, W=480, H=360):
    """Right image = left image seen with per-band disparities (3 depth layers) + sensor noise."""
    left = synth_image(seed, W, H, n_rect=170, n_disc=85)
    rng = np.random.default_rng(seed + 9)
    right = np.zeros_like(left)
    for y0, y1, d in ((0, H // 3, 7), (H // 3, 2 * H // 3, 19), (2 * H // 3, H, 33)):
        right[y0:y1] = np.roll(left[y0:y1], -d, axis=1)
    right = np.clip(right.astype(np.int32) + rng.integers(-2, 3, right.shape), 0, 255).astype(np.uint8)
    return left, right


def _check(lib, backend, seed):
    left, right = stereo_pair(seed)
    oL, oR = O.OrbOracle(600, 1.2, 8, 20, 7), O.OrbOracle(600, 1.2, 8, 20, 7)
    _, kl, dl = oL.extract(left, 0, 0)
    _, kr, dr = oR.extract(right, 0, 0)
    mb = BF / FX
    our, odp = O.stereo_matches(oL, oR, kl, dl, kr, dr, mb, BF)
    eL = orbhip.ORBextractor(600, 1.2, 8, 20, 7, lib=lib)
    eR = orbhip.ORBextractor(600, 1.2, 8, 20, 7, lib=lib)
    if backend == "emu":
        eL(left); eR(right)          # fills the two pyramids
        ur, dp = stereo_matches_host(eL, eR, kl, dl, kr, dr, mb, BF)
    else:
        import torch
        dLimg, dRimg = torch.from_numpy(left[None]).cuda(), torch.from_numpy(right[None]).cuda()   # level 0 of each pyramid aliases
        outL = eL.extract_batch(dLimg, (0, 0))                                                        # the caller's image batch:
        outR = eR.extract_batch(dRimg, (0, 0))                                                        # keep it alive
        u, d = stereo_matches(eL, eR, outL, outR, mb, BF)
        ur, dp = u.cpu().numpy()[0, :len(kl)], d.cpu().numpy()[0, :len(kl)]
    assert np.array_equal(ur.view(np.uint32), our.view(np.uint32)), "mvuRight (bitwise)"
    assert np.array_equal(dp.view(np.uint32), odp.view(np.uint32)), "mvDepth (bitwise)"
    ok = our >= 0
    assert ok.sum() > 100
    disp = kl["x"][ok] - our[ok]
    band = np.minimum(kl["y"][ok].astype(int) // 120, 2)
    assert np.median(np.abs(disp - np.array([7, 19, 33])[band])) < 0.6   # the planted disparities are recovered


def test_emu_stereo_matches(emu_lib):
    _check(emu_lib, "emu", 3)


def test_emu_stereo_row_lists_in_global_memory():
    """k_stereo_rows fills and orders a frame's vRowIndices lists in LDS when they fit (always, at these sizes) and in global memory otherwise;
    STEREO_ROWS_LDS_MAX=0 sends every frame down the global-memory path.  Results must not change."""
    import ctypes
    import build_emu
    from orbhip import _lib
    _check(_lib.bind(ctypes.CDLL(build_emu.build(defines=("STEREO_ROWS_LDS_MAX=0",), tag="rowslds0"))), "emu", 5)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4])
def test_hip_stereo_matches(hip_lib, seed):
    _check(hip_lib, "hip", seed)


# ---- M7 / M8 through the generic search (mode BEST_ONLY), checked against the oracle's literal loop ------------------
def _variant(lib, backend, th, th_dist, ori, seed):
    import test_matcher_parity as T
    from orbhip.matcher import MODE_BEST_ONLY, Q_HAS_OBS, Q_VALID
    S = T.scene()
    rng = np.random.default_rng(seed)
    q = T.make_queries(S, MODE_BEST_ONLY, th, None)
    q["flags"] = Q_VALID | Q_HAS_OBS                       # any matched keypoint blocks later points (mvpMapPoints[i2] != NULL)
    q["flags"][rng.random(len(q)) < 0.2] = 0               # sAlreadyFound / isBad / out of image / distance gate
    occ = (rng.random(len(S["kb"])) < 0.25).astype(np.uint8)   # CurrentFrame.mvpMapPoints[i2] != NULL before the call
    oq, ok, on = O.search_by_projection(S["kb"], S["db"], q, S["da"], S["grid"], MODE_BEST_ONLY, th_dist, 0.9, ori, None, occ)
    _, _, qm, km, nm = T.run_sbp(lib, backend, S, q, MODE_BEST_ONLY, th_dist, 0.9, ori, None, occ)
    assert nm[0] == on and np.array_equal(km[0, :len(S["kb"])], ok) and np.array_equal(qm[0, :len(q)], oq)


VARIANTS = [("reloc_th10_dist100", 10, 100, True), ("reloc_th3_dist64", 3, 64, True),      # ORBmatcher.cc:2520-2652 (Tracking.cc:3403,3417)
            ("sim3_th10_ratio1.0", 10, 50, False), ("sim3_th8_ratio0.75", 8, 37, False)]   # ORBmatcher.cc:593-706: bestDist <= TH_LOW*ratioHamming


@pytest.mark.parametrize("v", VARIANTS[:2] + VARIANTS[3:], ids=lambda v: v[0])
def test_emu_reloc_and_sim3_searches(emu_lib, v):
    _variant(emu_lib, "emu", v[1], v[2], v[3], 11)


@pytest.mark.gpu
@pytest.mark.parametrize("v", VARIANTS, ids=lambda v: v[0])
def test_hip_reloc_and_sim3_searches(hip_lib, v):
    _variant(hip_lib, "hip", v[1], v[2], v[3], 11)
