"""SURVEY rows M11 / M12 (the "next" row N1): SearchForInitialization, Fuse (both overloads' search half), SearchForTriangulation —
HIP kernels vs the oracle's restatements (oracle/match_oracle.cpp).  Bit-exact: identical match indices, distances and counts."""
import numpy as np
import pytest

import oracle_lib as O
import orbhip
from orbhip.matcher import Q_VALID, QUERY_DTYPE, TH_LOW, TRI_PAIR_DTYPE
from test_matcher_parity import feature_vector, scene, to_dev, to_host


def _slab(a, cap, B, dtype=None):
    a = np.asarray(a)
    out = np.zeros((B, cap) + a.shape[1:], dtype or a.dtype)
    out[:, :len(a)] = a
    return out


def kp_f32(k):
    return np.ascontiguousarray(k).view(np.float32).reshape(-1, 7)


# ---- M11 ----------------------------------------------------------------------------------------------------------------
def _init_case(lib, backend, window, ratio, ori, seed=0):
    S = scene()
    ka, da, kb, db = S["ka"], S["da"], S["kb"], S["db"]
    rng = np.random.default_rng(seed)
    prev0 = np.stack([ka["x"], ka["y"]], 1).astype(np.float32)     # Tracking.cc:1760-1762: vbPrevMatched = F1 keypoint positions
    prev0 += rng.integers(-2, 3, prev0.shape).astype(np.float32)
    B, c1, c2 = 2, len(ka) + 3, len(kb) + 6
    m = orbhip.ORBmatcher(ratio, ori, lib=lib)
    d = lambda a: to_dev(a, backend)
    k1, k2 = d(_slab(kp_f32(ka), c1, B)), d(_slab(kp_f32(kb), c2, B))
    d1, d2 = d(_slab(da, c1, B)), d(_slab(db, c2, B))
    n1, n2 = d(np.full(B, len(ka), np.int32)), d(np.full(B, len(kb), np.int32))
    gs, gi = m.grid_build(k2, n2, S["grid"])
    prev = d(_slab(prev0, c1, B))
    oprev = prev0
    total = 0
    for rnd in range(2):   # the second call starts from the updated vbPrevMatched, like consecutive frames during initialisation
        om, on, oprev = O.search_for_initialization(ka, da, kb, db, S["grid"], oprev, window, ratio, ori)
        qm, nm = m.SearchForInitialization(k1, d1, n1, k2, d2, n2, gs, gi, prev, S["grid"], window)
        qm, nm, ph = to_host(qm), to_host(nm), to_host(prev)
        for b in range(B):
            assert nm[b] == on, (rnd, nm[b], on)
            assert np.array_equal(qm[b, :len(ka)], om)
            assert np.array_equal(ph[b, :len(ka)], oprev)
        total += on
    assert total > 60
    return total


@pytest.mark.parametrize("window,ratio,ori", [(100, 0.9, True), (30, 0.9, False)])
def test_emu_search_for_initialization(emu_lib, window, ratio, ori):
    _init_case(emu_lib, "emu", window, ratio, ori)


@pytest.mark.gpu
@pytest.mark.parametrize("window,ratio,ori", [(100, 0.9, True), (30, 0.9, False), (10, 0.6, True), (200, 0.95, True)])
def test_hip_search_for_initialization(hip_lib, window, ratio, ori):
    _init_case(hip_lib, "hip", window, ratio, ori)


# ---- M12 Fuse -------------------------------------------------------------------------------------------------------------
def _fuse_case(lib, backend, chi2, th, seed=0):
    S = scene()
    ka, da, kb, db = S["ka"], S["da"], S["kb"], S["db"]
    rng = np.random.default_rng(seed)
    q = np.zeros(len(ka), QUERY_DTYPE)
    q["u"] = ka["x"] + np.float32(S["shift"][0]) + rng.normal(0, 0.8, len(ka)).astype(np.float32)
    q["v"] = ka["y"] + np.float32(S["shift"][1]) + rng.normal(0, 0.8, len(ka)).astype(np.float32)
    lvl = np.clip(ka["octave"] + rng.integers(-1, 2, len(ka)), 0, 7)          # nPredictedLevel (MapPoint::PredictScale)
    q["radius"] = np.float32(th) * S["scale"][lvl]
    q["min_level"] = lvl - 1; q["max_level"] = lvl
    q["u_right"] = q["u"] - np.float32(18.0)
    q["flags"] = np.where(rng.random(len(ka)) < 0.85, Q_VALID, 0)
    ur = (kb["x"] - np.float32(18.0) + rng.normal(0, 1.5, len(kb))).astype(np.float32)
    ur[rng.random(len(kb)) < 0.4] = -1
    inv_s2 = (np.float32(1.0) / (S["scale"] * S["scale"])).astype(np.float32)
    oqm, oqd, on = O.fuse(kb, db, q, da, S["grid"], TH_LOW, inv_s2 if chi2 else None, ur)
    B, ck, cq = 2, len(kb) + 7, len(q) + 2
    m = orbhip.ORBmatcher(lib=lib)
    d = lambda a: to_dev(a, backend)
    kps, nk = d(_slab(kp_f32(kb), ck, B)), d(np.full(B, len(kb), np.int32))
    gs, gi = m.grid_build(kps, nk, S["grid"])
    nqv = np.array([len(q), len(q) - 40], np.int32)
    qm, qd, nf = [to_host(x) for x in m.Fuse(kps, d(_slab(db, ck, B)), nk, gs, gi, d(_slab(q, cq, B).view(np.uint8).reshape(B, cq, 28)),
                                             d(_slab(da, cq, B)), d(nqv), S["grid"], inv_s2 if chi2 else None, d(_slab(ur, ck, B)))]
    assert nf[0] == on and np.array_equal(qm[0, :len(q)], oqm) and np.array_equal(qd[0, :len(q)], oqd)
    n1 = nqv[1]
    assert nf[1] == (oqm[:n1] >= 0).sum() and np.array_equal(qm[1, :n1], oqm[:n1]) and (qm[1, n1:] == -1).all()
    assert on > 80
    if chi2:   # the gate really removes candidates relative to the un-gated search
        _, _, on0 = O.fuse(kb, db, q, da, S["grid"], TH_LOW, None, ur)
        assert on0 > on


@pytest.mark.parametrize("chi2,th", [(True, 3.0), (False, 4.0)])
def test_emu_fuse(emu_lib, chi2, th):
    _fuse_case(emu_lib, "emu", chi2, th)


@pytest.mark.gpu
@pytest.mark.parametrize("chi2,th", [(True, 3.0), (False, 4.0), (True, 8.0)])
def test_hip_fuse(hip_lib, chi2, th):
    _fuse_case(hip_lib, "hip", chi2, th)


# ---- M12 SearchForTriangulation -----------------------------------------------------------------------------------------------
def _tri_case(lib, backend, only_stereo, coarse, ori, seed=0):
    S = scene()
    ka, da, kb, db = S["ka"], S["da"], S["kb"], S["db"]
    rng = np.random.default_rng(seed)
    sides = []
    for k, dsc in ((ka, da), (kb, db)):
        ids, st, fe = feature_vector(dsc, 60)
        ur = (k["x"] - np.float32(15.0)).astype(np.float32)
        ur[rng.random(len(k)) < 0.5] = -1
        sides.append(dict(kps=k, desc=dsc, u_right=ur, has_mp=(rng.random(len(k)) < 0.3).astype(np.uint8), node_id=ids, node_start=st,
                          feat_idx=fe, n_nodes=len(ids)))
    tx, ty = S["shift"]
    F12 = np.array([[0, 0, ty], [0, 0, -tx], [-ty, tx, 0]], np.float32) * np.float32(0.01)   # x1^T F12 x2 = 0 for x2 = x1 + s*(tx,ty)
    ep = np.array([S["W"] * 0.4, S["H"] * 0.55], np.float32)
    sig2 = (S["scale"] * S["scale"]).astype(np.float32)
    om, on = O.search_for_triangulation(sides[0], sides[1], F12, ep, sig2, S["scale"], only_stereo, coarse, ori)
    B = 3
    pairs = np.zeros(B, TRI_PAIR_DTYPE)
    pairs["F12"] = F12.reshape(9); pairs["ep"] = ep; pairs["level_sigma2_2"][:, :8] = sig2; pairs["scale_factors_2"][:, :8] = S["scale"]
    pairs["F12"][2] = 0            # degenerate pair: den == 0 -> epipolarConstrain false everywhere (unless bCoarse)
    d = lambda a: to_dev(a, backend)

    def slab(s, cap_f, cap_n):
        o = dict(kps=_slab(kp_f32(s["kps"]), cap_f, B), desc=_slab(s["desc"], cap_f, B), u_right=_slab(s["u_right"], cap_f, B),
                 has_mp=_slab(s["has_mp"], cap_f, B), node_id=_slab(s["node_id"], cap_n, B), node_start=_slab(s["node_start"], cap_n + 1, B),
                 feat_idx=_slab(s["feat_idx"], cap_f, B), n_nodes=np.full(B, s["n_nodes"], np.int32))
        return {k: d(v) for k, v in o.items()}
    m = orbhip.ORBmatcher(0.6, ori, lib=lib)
    m12, nm = [to_host(x) for x in m.SearchForTriangulation(slab(sides[0], len(ka) + 5, 70), slab(sides[1], len(kb) + 9, 64),
                                                            d(pairs.view(np.uint8).reshape(B, -1)), only_stereo, coarse)]
    for b in range(2):
        assert nm[b] == on, (nm[b], on)
        assert np.array_equal(m12[b, :len(ka)], om)
    om2, on2 = O.search_for_triangulation(sides[0], sides[1], np.zeros(9, np.float32), ep, sig2, S["scale"], only_stereo, coarse, ori)
    assert nm[2] == on2 and np.array_equal(m12[2, :len(ka)], om2)
    if not coarse:
        assert on2 == 0
    assert on > (5 if only_stereo else 25)
    return on


TRI_CASES = [(False, False, True), (True, False, True), (False, True, False)]


@pytest.mark.parametrize("only_stereo,coarse,ori", TRI_CASES)
def test_emu_search_for_triangulation(emu_lib, only_stereo, coarse, ori):
    _tri_case(emu_lib, "emu", only_stereo, coarse, ori)


@pytest.mark.gpu
@pytest.mark.parametrize("only_stereo,coarse,ori", TRI_CASES + [(False, False, False)])
def test_hip_search_for_triangulation(hip_lib, only_stereo, coarse, ori):
    _tri_case(hip_lib, "hip", only_stereo, coarse, ori)


def test_oracle_triangulation_epipolar_gate_is_selective():
    """The restated gate (Pinhole.cpp:155-177) rejects a wrong geometry: with F12 for a perpendicular translation almost nothing passes."""
    S = scene()
    ka, da, kb, db = S["ka"], S["da"], S["kb"], S["db"]
    sides = []
    for k, dsc in ((ka, da), (kb, db)):
        ids, st, fe = feature_vector(dsc, 60)
        sides.append(dict(kps=k, desc=dsc, u_right=None, has_mp=np.zeros(len(k), np.uint8), node_id=ids, node_start=st, feat_idx=fe, n_nodes=len(ids)))
    tx, ty = S["shift"]
    sig2 = (S["scale"] * S["scale"]).astype(np.float32)
    ep = np.array([-1e4, -1e4], np.float32)
    good = np.array([[0, 0, ty], [0, 0, -tx], [-ty, tx, 0]], np.float32)
    bad = np.array([[0, 0, tx], [0, 0, ty], [-tx, -ty, 0]], np.float32)
    _, n_good = O.search_for_triangulation(sides[0], sides[1], good, ep, sig2, S["scale"], False, False, False)
    _, n_bad = O.search_for_triangulation(sides[0], sides[1], bad, ep, sig2, S["scale"], False, False, False)
    assert n_good > 60 and n_bad < 0.25 * n_good


# ---- SearchBySim3 (ORBmatcher.cc:2008-2220): fuse-style search in both directions + agreement -----------------------------------------
def _sim3_case(lib, backend, th, seed=0):
    S = scene()
    ka, da, kb, db = S["ka"], S["da"], S["kb"], S["db"]
    rng = np.random.default_rng(seed)

    def queries(src, sgn):
        q = np.zeros(len(src), QUERY_DTYPE)
        q["u"] = src["x"] + np.float32(sgn * S["shift"][0]) + rng.normal(0, 0.7, len(src)).astype(np.float32)
        q["v"] = src["y"] + np.float32(sgn * S["shift"][1]) + rng.normal(0, 0.7, len(src)).astype(np.float32)
        lvl = np.clip(src["octave"] + rng.integers(-1, 2, len(src)), 0, 7)
        q["radius"] = np.float32(th) * S["scale"][lvl]
        q["min_level"] = lvl - 1; q["max_level"] = lvl
        q["flags"] = np.where(rng.random(len(src)) < 0.8, Q_VALID, 0)     # no map point / already matched / failed a gate
        return q
    q12, q21 = queries(ka, +1), queries(kb, -1)
    om, on = O.search_by_sim3(ka, da, S["grid"], kb, db, S["grid"], q12, da, q21, db)
    B, c1, c2 = 2, len(ka) + 5, len(kb) + 9
    m = orbhip.ORBmatcher(lib=lib)
    d = lambda a: to_dev(a, backend)
    sides = []
    for k, dsc, c in ((ka, da, c1), (kb, db, c2)):
        kps, n = d(_slab(kp_f32(k), c, B)), d(np.full(B, len(k), np.int32))
        gs, gi = m.grid_build(kps, n, S["grid"])
        sides.append(dict(kps=kps, desc=d(_slab(dsc, c, B)), counts=n, grid_start=gs, grid_idx=gi, grid=S["grid"]))
    dq = lambda q, c: d(_slab(q, c, B).view(np.uint8).reshape(B, c, 28))
    out, nf = m.SearchBySim3(sides[0], sides[1], dq(q12, c1), d(_slab(da, c1, B)), dq(q21, c2), d(_slab(db, c2, B)))
    out, nf = to_host(out), to_host(nf)
    for b in range(B):
        assert nf[b] == on and np.array_equal(out[b, :len(ka)], om) and (out[b, len(ka):] == -1).all()
    assert on > 60
    # the agreement pass really rejects one-directional matches
    one_way, _, _ = O.fuse(kb, db, q12, da, S["grid"], 100, None, None)
    assert (one_way >= 0).sum() > on


@pytest.mark.parametrize("th", [7.5])
def test_emu_search_by_sim3(emu_lib, th):
    _sim3_case(emu_lib, "emu", th)


@pytest.mark.gpu
@pytest.mark.parametrize("th", [7.5, 3.0, 15.0])
def test_hip_search_by_sim3(hip_lib, th):
    _sim3_case(hip_lib, "hip", th)
