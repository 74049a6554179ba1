"""Fisheye-rig (F.Nleft != -1) twins of the projection searches (right-camera branches of ORBmatcher.cc:59-258 and :2244-2509): HIP vs the
oracle restatement.  The rig frame = [left keypoints | right keypoints] like the reference's N-sized arrays; map points contribute a left
query and a right twin.  Bit-exact match arrays, including the reference's `continue` quirks and the stereo-partner copies."""
import numpy as np
import pytest

import oracle_lib as O
import orbhip
from orbhip.matcher import MODE_BEST_ONLY, MODE_LOCAL_MAP, Q_HAS_OBS, Q_RIGHT, Q_TWIN, Q_VALID, QUERY_DTYPE, TH_HIGH
from test_matcher_parity import scene, to_dev, to_host


def rig_problem(mode, th, seed=0):
    """Left camera = frame B of the scene, right camera = frame A; every map point (a keypoint of A) projects at +shift in the left camera
    and at its own position in the right camera."""
    S = scene()
    rng = np.random.default_rng(seed)
    ka, da, kb, db = S["ka"], S["da"], S["kb"], S["db"]
    kps = np.concatenate([kb, ka]); desc = np.concatenate([db, da])
    nleft = len(kb)
    # stereo partners: a sparse, consistent left<->right pairing
    link = np.full(len(kps), -1, np.int32)
    li = rng.permutation(nleft)[:nleft // 3]; ri = rng.permutation(len(ka))[:len(li)]
    link[li] = ri + nleft; link[ri + nleft] = li
    nmp = len(ka)
    q = np.zeros(2 * nmp, QUERY_DTYPE)
    lvl = ka["octave"]
    L, R = q[0::2], q[1::2]
    L["u"] = ka["x"] + np.float32(S["shift"][0]); L["v"] = ka["y"] + np.float32(S["shift"][1])
    R["u"] = ka["x"] + rng.normal(0, 0.7, nmp).astype(np.float32); R["v"] = ka["y"] + rng.normal(0, 0.7, nmp).astype(np.float32)
    for side in (L, R):
        side["angle"] = ka["angle"]
        if mode == MODE_BEST_ONLY:
            side["radius"] = np.float32(th) * S["scale"][lvl]; side["min_level"] = lvl - 1; side["max_level"] = lvl + 1
        else:
            side["radius"] = np.float32(4.0 * th) * S["scale"][lvl]; side["min_level"] = lvl - 1; side["max_level"] = lvl
    L["flags"] = Q_VALID | Q_HAS_OBS
    R["flags"] = Q_VALID | Q_HAS_OBS | Q_RIGHT | Q_TWIN
    # some map points are seen by one camera only, some are temporal points (no observations)
    only_r = rng.random(nmp) < 0.1; only_l = (rng.random(nmp) < 0.1) & ~only_r
    L["flags"][only_r] = 0; R["flags"][only_l] = Q_RIGHT | Q_TWIN
    noobs = rng.random(nmp) < 0.1
    L["flags"][noobs] &= ~np.uint32(Q_HAS_OBS); R["flags"][noobs] &= ~np.uint32(Q_HAS_OBS)
    qd = np.repeat(da, 2, axis=0)
    occ = (rng.random(len(kps)) < 0.1).astype(np.uint8)
    return S, kps, desc, nleft, link, q, qd, occ


def run_rig(lib, backend, mode, th, ratio, ori, seed=0, use_link=True):
    S, kps, desc, nleft, link, q, qd, occ = rig_problem(mode, th, seed)
    lk = link if use_link else None
    ogs, ogi = O.grid_build_rig(kps, nleft, S["grid"])
    oq, ok, on = O.search_by_projection_rig(kps, desc, nleft, lk, q, qd, S["grid"], mode, TH_HIGH, ratio, ori, occ)
    B, ck, cq = 2, len(kps) + 9, len(q) + 5
    m = orbhip.ORBmatcher(ratio, ori, lib=lib)
    d = lambda a: to_dev(a, backend)
    slab = lambda a, cap: np.concatenate([a[None]] * B)[:, :cap] if len(a) >= cap else np.concatenate(
        [np.concatenate([a, np.zeros((cap - len(a),) + a.shape[1:], a.dtype)])[None]] * B)
    dk = d(slab(np.ascontiguousarray(kps).view(np.float32).reshape(-1, 7), ck))
    nk, nl = d(np.full(B, len(kps), np.int32)), d(np.full(B, nleft, np.int32))
    gs, gi = m.grid_build_rig(dk, nk, nl, S["grid"])
    assert np.array_equal(to_host(gs)[0], ogs) and np.array_equal(to_host(gi)[0, :len(kps)], ogi[:len(kps)])
    lk_slab = None
    if lk is not None:
        lk_slab = np.full((B, ck), -1, np.int32); lk_slab[:, :len(kps)] = lk
    qm, km, nm = [to_host(x) for x in m.SearchByProjectionRig(dk, d(slab(desc, ck)), nk, gs, gi, d(slab(q, cq).view(np.uint8).reshape(B, cq, 28)),
                                                              d(slab(qd, cq)), d(np.full(B, len(q), np.int32)), S["grid"], mode, TH_HIGH,
                                                              kp_link=d(lk_slab), occupied0=d(slab(occ, ck)))]
    for b in range(B):
        assert nm[b] == on, (nm[b], on)
        assert np.array_equal(km[b, :len(kps)], ok) and np.array_equal(qm[b, :len(q)], oq)
    return S, q, oq, ok, on, nleft, link


def test_oracle_rig_quirks_are_exercised():
    """The restated quirks really fire on the test problem: partner copies, twins skipped by the left `continue`, right-only points."""
    S, kps, desc, nleft, link, q, qd, occ = rig_problem(MODE_LOCAL_MAP, 3, 0)
    oq, ok, on = O.search_by_projection_rig(kps, desc, nleft, link, q, qd, S["grid"], MODE_LOCAL_MAP, TH_HIGH, 0.8, True, occ)
    oq2, ok2, on2 = O.search_by_projection_rig(kps, desc, nleft, None, q, qd, S["grid"], MODE_LOCAL_MAP, TH_HIGH, 0.8, True, occ)
    assert on > on2 > 50                       # stereo-partner copies add matches
    assert ((ok >= 0) & (ok < len(q)) & (np.arange(len(ok)) >= nleft)).any() and (oq[1::2] >= nleft).any()   # right camera matched
    held_by_left = (ok[nleft:] >= 0) & (ok[nleft:] % 2 == 0)
    assert held_by_left.any()                  # a right keypoint holding a LEFT query's map point = a partner copy
    # dropping the TWIN flag (no skip rule) changes the result -> the skip rule is active
    q3 = q.copy(); q3["flags"] &= ~np.uint32(Q_TWIN)
    _, ok3, on3 = O.search_by_projection_rig(kps, desc, nleft, link, q3, qd, S["grid"], MODE_LOCAL_MAP, TH_HIGH, 0.8, True, occ)
    assert on3 != on or not np.array_equal(ok3, ok)


RIG_CASES = [(MODE_LOCAL_MAP, 3, 0.8, True, True), (MODE_BEST_ONLY, 15, 0.9, True, True), (MODE_BEST_ONLY, 4, 0.9, False, False),
             (MODE_LOCAL_MAP, 1, 0.6, True, False)]


@pytest.mark.parametrize("mode,th,ratio,ori,link", RIG_CASES[:3])
def test_emu_search_by_projection_rig(emu_lib, mode, th, ratio, ori, link):
    run_rig(emu_lib, "emu", mode, th, ratio, ori, use_link=link)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,th,ratio,ori,link", RIG_CASES)
def test_hip_search_by_projection_rig(hip_lib, mode, th, ratio, ori, link):
    run_rig(hip_lib, "hip", mode, th, ratio, ori, use_link=link)
