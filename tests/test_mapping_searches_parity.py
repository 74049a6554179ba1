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


# ---- SearchForTriangulation on KannalaBrandt8 key frames (fisheye rig with mpCamera2, or one fisheye camera) ------------------------------
# ORBmatcher.cc:1138-1428 rig branches + KannalaBrandt8::epipolarConstrain (KannalaBrandt8.cpp:235-238 -> TriangulateMatches :334-400, rule R4)
from orbhip._lib import KP_DTYPE  # noqa: E402
from orbhip.lba import _kb8_project, _rodrigues  # noqa: E402
from orbhip.matcher import TRI_KB8_PAIR_DTYPE  # noqa: E402

KB_A = np.float32([190.978, 190.973, 254.932, 256.897, 0.0034823894, 0.0007150348, -0.0020532361, 0.00020293673])   # TUM_512.yaml-like
KB_B = np.float32([190.442, 190.434, 252.597, 254.917, 0.0034003171, 0.0017662670, -0.0026630025, 0.00032995968])


def kb8_keyframe_pair(seed, n_pts=420, rig=True, n_distract=60, wrong_geometry=False):
    """Two key frames of one fisheye rig (or of one fisheye camera) looking at the same 3-D points: per key frame the concatenation
    [mvKeys | mvKeysRight], descriptors, GetMapPoint flags and a FeatureVector in which both views of a point share a node."""
    rng = np.random.default_rng(seed)
    R_rl = _rodrigues(np.array([0.003, -0.011, 0.002])); t_rl = np.array([-0.101, 0.0012, -0.0007])     # x_right = R_rl x_left + t_rl
    T = []                                                                                               # [kf][cam] -> (R_cw, t_cw)
    for kf, (w, t) in enumerate((((0.01, -0.02, 0.005), (0.0, 0.0, 0.0)), ((-0.03, 0.05, -0.01), (-0.35, 0.04, 0.06)))):
        Rl, tl = _rodrigues(np.array(w)), np.array(t)
        T.append([(Rl, tl), (R_rl @ Rl, R_rl @ tl + t_rl)])
    X = np.stack([rng.uniform(-4, 4, n_pts), rng.uniform(-3, 3, n_pts), rng.uniform(1.5, 9, n_pts)], 1)
    base_desc = rng.integers(0, 256, (n_pts, 32), dtype=np.uint8)
    base_ang = rng.uniform(0, 360, n_pts)
    kb = (KB_A, KB_B)
    sides, nlefts = [], []
    for kf in range(2):
        kps, descs, nodes = [], [], []
        for cam in range(2 if rig else 1):
            R, t = T[kf][cam]
            Xc = X @ R.T + t
            u, v, th = _kb8_project(kb[cam].astype(np.float64), Xc)
            vis = (Xc[:, 2] > 0.3) & (th < 1.25) & (u > 5) & (u < 507) & (v > 5) & (v < 507) & (rng.random(n_pts) < 0.8)
            ids = np.nonzero(vis)[0]
            k = np.zeros(len(ids) + n_distract, KP_DTYPE)
            k["x"][:len(ids)] = u[ids] + rng.normal(0, 0.35, len(ids)); k["y"][:len(ids)] = v[ids] + rng.normal(0, 0.35, len(ids))
            k["x"][len(ids):] = rng.uniform(20, 490, n_distract); k["y"][len(ids):] = rng.uniform(20, 490, n_distract)
            k["octave"] = rng.integers(0, 4, len(k)); k["size"], k["response"], k["class_id"] = 31, 40, -1
            k["angle"][:len(ids)] = (base_ang[ids] + 12.0 * kf + rng.normal(0, 2, len(ids))) % 360
            k["angle"][len(ids):] = rng.uniform(0, 360, n_distract)
            flips = np.zeros((len(ids), 256), np.uint8)
            for i in range(len(ids)):
                flips[i, rng.choice(256, int(rng.integers(0, 30)), replace=False)] = 1
            d_true = base_desc[ids] ^ np.packbits(flips, axis=1, bitorder="little")
            # distractors: half of them copy a real point's descriptor (and its node) at a wrong image position -> only the geometric gate rejects them
            src = rng.choice(n_pts, n_distract)
            d_dis = np.where((np.arange(n_distract) % 2 == 0)[:, None], base_desc[src], rng.integers(0, 256, (n_distract, 32), dtype=np.uint8))
            kps.append(k); descs.append(np.concatenate([d_true, d_dis]).astype(np.uint8))
            nodes.append(np.concatenate([ids % 37, src % 37]))
        nlefts.append(len(kps[0]) if rig else -1)
        k_all, d_all, node = np.concatenate(kps), np.concatenate(descs), np.concatenate(nodes)
        perm = rng.permutation(len(kps[0]))      # feature order inside each camera is unrelated to the point order
        k_all[:len(perm)], d_all[:len(perm)], node[:len(perm)] = k_all[perm], d_all[perm], node[perm]
        ids_n = np.unique(node)
        start, feat = [0], []
        for nid in ids_n:
            feat += np.nonzero(node == nid)[0].tolist(); start.append(len(feat))
        sides.append(dict(kps=k_all, desc=d_all, u_right=None, has_mp=(rng.random(len(k_all)) < 0.25).astype(np.uint8), node_id=ids_n.astype(np.int32),
                          node_start=np.array(start, np.int32), feat_idx=np.array(feat, np.int32), n_nodes=len(ids_n)))
    pair = np.zeros(1, TRI_KB8_PAIR_DTYPE)
    pair["n_cams"] = 2 if rig else 1
    pair["k1"][0] = [KB_A, KB_B]; pair["k2"][0] = [KB_A, KB_B]
    for b1 in range(2):
        for b2 in range(2):
            R1, t1 = T[0][b1 if rig else 0]; R2, t2 = T[1][b2 if rig else 0]
            R12 = (R1.astype(np.float32) @ R2.astype(np.float32).T).astype(np.float32)          # ORBmatcher.cc:1176-1177, 1181-1193 in CV_32F
            t12 = (-(R12 @ t2.astype(np.float32)) + t1.astype(np.float32)).astype(np.float32)
            if wrong_geometry:
                t12 = (t12[[1, 2, 0]] * np.float32(-1.5)).astype(np.float32)
            pair["R12"][0, b1 * 2 + b2] = R12.reshape(9); pair["t12"][0, b1 * 2 + b2] = t12
    R1, t1 = T[0][0]; R2, t2 = T[1][0]
    C2 = R2 @ (-R1.T @ t1) + t2                                                                    # KF1's camera centre in KF2's camera frame (:1144-1148)
    ue, ve, _ = _kb8_project(KB_A.astype(np.float64), C2[None])
    pair["ep"][0] = [ue[0], ve[0]]
    sf = (np.float32(1.2) ** np.arange(16, dtype=np.float32)).astype(np.float32)
    pair["scale_factors_2"][0] = sf; pair["level_sigma2_1"][0] = sf * sf; pair["level_sigma2_2"][0] = sf * sf
    return sides, nlefts, pair


def _tri_kb8_case(lib, backend, rig, only_stereo, coarse, ori, seed=3):
    variants = [kb8_keyframe_pair(seed, rig=rig), kb8_keyframe_pair(seed, rig=rig, wrong_geometry=True), kb8_keyframe_pair(seed + 1, n_pts=150, rig=rig, n_distract=200)]
    B = len(variants)
    c1 = max(len(v[0][0]["kps"]) for v in variants) + 5
    c2 = max(len(v[0][1]["kps"]) for v in variants) + 9
    d = lambda a: to_dev(a, backend)

    def slabs(which, cap_f, cap_n):
        o = dict(kps=np.zeros((B, cap_f, 7), np.float32), desc=np.zeros((B, cap_f, 32), np.uint8), has_mp=np.zeros((B, cap_f), np.uint8),
                 node_id=np.zeros((B, cap_n), np.int32), node_start=np.zeros((B, cap_n + 1), np.int32), feat_idx=np.zeros((B, cap_f), np.int32),
                 n_nodes=np.zeros(B, np.int32))
        for b, v in enumerate(variants):
            s = v[0][which]
            n, nn = len(s["kps"]), s["n_nodes"]
            o["kps"][b, :n] = kp_f32(s["kps"]); o["desc"][b, :n] = s["desc"]; o["has_mp"][b, :n] = s["has_mp"]
            o["node_id"][b, :nn] = s["node_id"]; o["node_start"][b, :nn + 1] = s["node_start"]; o["feat_idx"][b, :len(s["feat_idx"])] = s["feat_idx"]
            o["n_nodes"][b] = nn
        return {k: d(v) for k, v in o.items()}
    pairs = np.concatenate([v[2] for v in variants])
    nl1 = np.array([v[1][0] for v in variants], np.int32); nl2 = np.array([v[1][1] for v in variants], np.int32)
    m = orbhip.ORBmatcher(0.6, ori, lib=lib)
    m12, nm = [to_host(x) for x in m.SearchForTriangulationKB8(slabs(0, c1, 40), slabs(1, c2, 44), d(nl1), d(nl2), d(pairs.view(np.uint8).reshape(B, -1)),
                                                               only_stereo, coarse)]
    counts = []
    for b, (sides, nlefts, pair) in enumerate(variants):
        om, on = O.search_for_triangulation_kb8(sides[0], sides[1], nlefts[0], nlefts[1], pair, only_stereo, coarse, ori)
        n1 = len(sides[0]["kps"])
        assert nm[b] == on, (b, nm[b], on)
        assert np.array_equal(m12[b, :n1], om) and (m12[b, n1:] == -1).all(), b
        counts.append(on)
    if only_stereo:
        assert counts == [0, 0, 0]            # bStereo1 is false for fisheye key frames (:1241-1245)
    elif coarse:
        assert counts[0] > 100 and counts[1] > 100   # bCoarse short-circuits the gate: geometry is irrelevant
    else:
        assert counts[0] > 60 and counts[1] < 0.5 * counts[0], counts   # the gate accepts the true geometry and refuses most of a wrong one
    return counts


TRI_KB8 = [(True, False, False, True), (False, False, False, True), (True, False, True, False), (True, True, False, True), (False, False, False, False)]


@pytest.mark.parametrize("rig,only_stereo,coarse,ori", TRI_KB8[:4])
def test_emu_search_for_triangulation_kb8(emu_lib, rig, only_stereo, coarse, ori):
    _tri_kb8_case(emu_lib, "emu", rig, only_stereo, coarse, ori)


@pytest.mark.gpu
@pytest.mark.parametrize("rig,only_stereo,coarse,ori", TRI_KB8)
def test_hip_search_for_triangulation_kb8(hip_lib, rig, only_stereo, coarse, ori):
    _tri_kb8_case(hip_lib, "hip", rig, only_stereo, coarse, ori)
