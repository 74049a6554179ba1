"""Stage-2 parity: HIP matcher kernels vs the oracle's restatement of ORBmatcher / Frame grid helpers.
backend "emu": product kernels compiled against tests/emu (CPU logic check); "hip": real library on an MI355X."""
import numpy as np
import pytest

import oracle_lib as O
import orbhip
from orbhip.matcher import (MODE_BEST_ONLY, MODE_LOCAL_MAP, Q_HAS_OBS, Q_STEREO, Q_VALID, QUERY_DTYPE, TH_HIGH)
from orbhip.synth import synth_image

_SCENES = {}


def scene(W=480, H=360, nf=600, seed=40, shift=(6, -4)):
    """Two frames of one synthetic scene (frame B = frame A shifted + re-noised), extracted by the oracle."""
    key = (W, H, nf, seed, shift)
    if key not in _SCENES:
        a = synth_image(seed, W, H, n_rect=160, n_disc=80)
        rng = np.random.default_rng(seed + 1)
        b = np.clip(np.roll(a, (shift[1], shift[0]), (0, 1)).astype(np.int32) + rng.integers(-3, 4, a.shape), 0, 255).astype(np.uint8)
        o = O.OrbOracle(nf, 1.2, 8, 20, 7)
        _, ka, da = o.extract(a, 0, 0)
        _, kb, db = o.extract(b, 0, 0)
        scale = o.tables()["scale"]
        grid = (0.0, 0.0, np.float32(64) / np.float32(W), np.float32(48) / np.float32(H))
        _SCENES[key] = dict(ka=ka, da=da, kb=kb, db=db, scale=scale, grid=tuple(float(g) for g in grid), shift=shift, W=W, H=H)
    return _SCENES[key]


def make_queries(S, mode, th, rng=None, stereo=False):
    ka = S["ka"]
    q = np.zeros(len(ka), QUERY_DTYPE)
    q["u"] = ka["x"] + np.float32(S["shift"][0]); q["v"] = ka["y"] + np.float32(S["shift"][1])
    lvl = ka["octave"]
    if mode == MODE_BEST_ONLY:   # ORBmatcher.cc:2309-2331 (neither forward nor backward)
        q["radius"] = np.float32(th) * S["scale"][lvl]
        q["min_level"] = lvl - 1; q["max_level"] = lvl + 1
    else:                        # ORBmatcher.cc:88-103: r = RadiusByViewingCos * th * scale[level], levels [l-1, l]
        viewcos = np.where(np.arange(len(ka)) % 3 == 0, 0.9990, 0.95).astype(np.float32)
        r = np.where(viewcos > np.float32(0.998), np.float32(2.5), np.float32(4.0)) * np.float32(th)
        q["radius"] = r.astype(np.float32) * S["scale"][lvl]
        q["min_level"] = lvl - 1; q["max_level"] = lvl
    q["angle"] = ka["angle"]
    q["flags"] = Q_VALID | Q_HAS_OBS
    if rng is not None:
        drop = rng.random(len(ka)) < 0.1
        q["flags"][drop] = 0
        q["flags"][rng.random(len(ka)) < 0.1] &= ~np.uint32(Q_HAS_OBS)   # temporal points: Observations()==0 -> may be overwritten
    if stereo:
        q["flags"] |= Q_STEREO
        q["u_right"] = q["u"] - np.float32(20.0)
    return q


def to_dev(a, backend):
    if backend == "emu" or a is None:
        return a
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def to_host(a):
    return a if isinstance(a, np.ndarray) else a.cpu().numpy()


def _view_u16(a):
    a = to_host(a)
    return a.view(np.uint16) if a.dtype == np.int16 else a


def run_sbp(lib, backend, S, q, mode, th_dist, nnratio, check_ori, u_right=None, occupied0=None, reps=1):
    m = orbhip.ORBmatcher(nnratio, check_ori, lib=lib)
    kb, db = S["kb"], S["db"]
    B = reps
    cap_k, cap_q = len(kb) + 5, len(q) + 3
    kps = np.zeros((B, cap_k, 7), np.float32); kps[:, :len(kb)] = kb.view(np.float32).reshape(-1, 7)
    desc = np.zeros((B, cap_k, 32), np.uint8); desc[:, :len(kb)] = db
    qs = np.zeros((B, cap_q), QUERY_DTYPE); qs[:, :len(q)] = q
    qd = np.zeros((B, cap_q, 32), np.uint8); qd[:, :len(q)] = S["da"]
    nk = np.full(B, len(kb), np.int32); nq = np.full(B, len(q), np.int32)
    ur = None if u_right is None else np.zeros((B, cap_k), np.float32)
    if ur is not None:
        ur[:, :len(kb)] = u_right
    oc = None if occupied0 is None else np.zeros((B, cap_k), np.uint8)
    if oc is not None:
        oc[:, :len(kb)] = occupied0
    d = lambda a: to_dev(a, backend)
    dk, dn = d(kps), d(nk)
    gs, gi = m.grid_build(dk, dn, S["grid"])
    qm, km, nm = m.SearchByProjection(dk, d(desc), dn, gs, gi, d(qs.view(np.uint8).reshape(B, cap_q, 28)), d(qd), d(nq), S["grid"], mode,
                                      th_dist, u_right=d(ur), occupied0=d(oc))
    return to_host(gs), to_host(gi), to_host(qm), to_host(km), to_host(nm)


def check_sbp(lib, backend, mode, th, nnratio, check_ori, stereo=False, occupied=False, seed=0, scene_kw=None):
    S = scene(**(scene_kw or {}))
    rng = np.random.default_rng(seed)
    q = make_queries(S, mode, th, rng, stereo)
    kb = S["kb"]
    ur = None
    if stereo:
        ur = (kb["x"] - np.float32(20.0) + rng.normal(0, 6, len(kb))).astype(np.float32)
        ur[rng.random(len(kb)) < 0.3] = -1
    oc = (rng.random(len(kb)) < 0.15).astype(np.uint8) if occupied else None
    ogs, ogi = O.grid_build(kb, S["grid"])
    oq, ok, on = O.search_by_projection(kb, S["db"], q, S["da"], S["grid"], mode, TH_HIGH, nnratio, check_ori, ur, oc)
    gs, gi, qm, km, nm = run_sbp(lib, backend, S, q, mode, TH_HIGH, nnratio, check_ori, ur, oc, reps=2)
    for b in range(2):
        assert np.array_equal(gs[b], ogs) and np.array_equal(gi[b, :len(kb)], ogi[:len(kb)]), "grid CSR"
        assert nm[b] == on, (nm[b], on)
        assert np.array_equal(km[b, :len(kb)], ok), "mvpMapPoints"
        assert np.array_equal(qm[b, :len(q)], oq), "per-query match"
    assert on > (50 if scene_kw is None else 5)  # the scene really matches


SBP_CASES = [
    ("motion_model_th15_ori", MODE_BEST_ONLY, 15, 0.9, True, False, False),
    ("motion_model_th7_stereo", MODE_BEST_ONLY, 7, 0.9, True, True, False),
    ("motion_model_no_ori", MODE_BEST_ONLY, 30, 0.9, False, False, True),
    ("local_map_th1", MODE_LOCAL_MAP, 1, 0.8, True, False, False),
    ("local_map_th5_occupied_stereo", MODE_LOCAL_MAP, 5, 0.8, True, True, True),
    ("local_map_th15_wide", MODE_LOCAL_MAP, 15, 0.6, True, False, False),
]


@pytest.mark.parametrize("case", SBP_CASES[:4], ids=lambda c: c[0])
def test_emu_search_by_projection(emu_lib, case):
    _, mode, th, ratio, ori, stereo, occ = case
    check_sbp(emu_lib, "emu", mode, th, ratio, ori, stereo, occ)


@pytest.mark.gpu
@pytest.mark.parametrize("case", SBP_CASES, ids=lambda c: c[0])
def test_hip_search_by_projection(hip_lib, case):
    _, mode, th, ratio, ori, stereo, occ = case
    check_sbp(hip_lib, "hip", mode, th, ratio, ori, stereo, occ)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(12))
def test_hip_fuzz_search_by_projection(hip_lib, i):
    """Random scenes (size, feature count, inter-frame shift) x random search parameters; both modes; stereo gate / occupancy on or off."""
    rng = np.random.default_rng(500 + i)
    kw = dict(W=int(rng.integers(320, 900)), H=int(rng.integers(240, 600)), nf=int(rng.choice([300, 600, 1200])), seed=int(rng.integers(1000)),
              shift=(int(rng.integers(-9, 10)), int(rng.integers(-9, 10))))
    mode = MODE_BEST_ONLY if i % 2 else MODE_LOCAL_MAP
    check_sbp(hip_lib, "hip", mode, int(rng.choice([1, 3, 7, 15, 30])), float(rng.choice([0.6, 0.8, 0.9])), bool(rng.integers(2)),
              stereo=bool(rng.integers(2)), occupied=bool(rng.integers(2)), seed=i, scene_kw=kw)


@pytest.mark.parametrize("i", range(2))
def test_emu_fuzz_search_by_projection(emu_lib, i):
    """CPU-tier slice of the random sweep (small scenes): list-length-packed resolver batches, clashes, both modes, stereo gate / occupancy."""
    rng = np.random.default_rng(900 + i)
    kw = dict(W=int(rng.integers(320, 640)), H=int(rng.integers(240, 480)), nf=300, seed=int(rng.integers(1000)),
              shift=(int(rng.integers(-6, 7)), int(rng.integers(-6, 7))))
    mode = MODE_BEST_ONLY if i % 2 else MODE_LOCAL_MAP
    check_sbp(emu_lib, "emu", mode, int(rng.choice([3, 7, 15, 30])), float(rng.choice([0.6, 0.8, 0.9])), bool(rng.integers(2)),
              stereo=bool(rng.integers(2)), occupied=bool(rng.integers(2)), seed=i, scene_kw=kw)


def _overflow_case(lib, backend, max_queries=None):
    """> 64 candidates per query: the resolver's inline re-enumeration path."""
    S = scene()
    q = make_queries(S, MODE_LOCAL_MAP, 40, np.random.default_rng(5))
    if max_queries:
        q = q[:max_queries].copy()   # the emulator runs this path at ~0.1 s per query: a slice is enough on the CPU tier
        S = dict(S); S["da"] = S["da"][:max_queries]
    q["min_level"] = -1; q["max_level"] = -1   # no level filter -> every keypoint in a ~100-400 px window
    oq, ok, on = O.search_by_projection(S["kb"], S["db"], q, S["da"], S["grid"], MODE_LOCAL_MAP, TH_HIGH, 0.8, True)
    _, _, qm, km, nm = run_sbp(lib, backend, S, q, MODE_LOCAL_MAP, TH_HIGH, 0.8, True)
    assert nm[0] == on and np.array_equal(km[0, :len(S["kb"])], ok) and np.array_equal(qm[0, :len(q)], oq)


def test_emu_search_by_projection_candidate_overflow(emu_lib):
    _overflow_case(emu_lib, "emu", max_queries=120)


@pytest.mark.gpu
def test_hip_search_by_projection_candidate_overflow(hip_lib):
    _overflow_case(hip_lib, "hip")


def _mixed_window_case(lib, backend, max_queries=None, seed=11):
    """Window radii log-uniform in 4..320 px, level filter on for half of the queries: neighbouring queries of one wave land on different paths of
    the two-queries-per-wave candidates kernel (half-wave lists, > 32-entry lists, > 32-column windows, > 64-entry overflow)."""
    S = scene()
    rng = np.random.default_rng(seed)
    q = make_queries(S, MODE_LOCAL_MAP, 5, rng)
    if max_queries:
        q = q[:max_queries].copy()
        S = dict(S); S["da"] = S["da"][:max_queries]
    q["radius"] = np.exp(rng.uniform(np.log(4.0), np.log(320.0), len(q))).astype(np.float32)
    nolevel = rng.random(len(q)) < 0.5
    q["min_level"][nolevel] = -1; q["max_level"][nolevel] = -1
    oq, ok, on = O.search_by_projection(S["kb"], S["db"], q, S["da"], S["grid"], MODE_LOCAL_MAP, TH_HIGH, 0.8, True)
    _, _, qm, km, nm = run_sbp(lib, backend, S, q, MODE_LOCAL_MAP, TH_HIGH, 0.8, True)
    assert nm[0] == on and np.array_equal(km[0, :len(S["kb"])], ok) and np.array_equal(qm[0, :len(q)], oq)
    assert on > 20


def test_emu_search_by_projection_mixed_windows(emu_lib):
    _mixed_window_case(emu_lib, "emu", max_queries=90)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_hip_search_by_projection_mixed_windows(hip_lib, seed):
    _mixed_window_case(hip_lib, "hip", seed=seed)


def test_descriptor_distance_kats():
    # bit-hack popcount == popcount; known answers
    z, o = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    assert O.hamming(z, z) == 0 and O.hamming(z, o) == 256 and O.hamming(o, o) == 0
    one = z.copy(); one[17] = 0x10
    assert O.hamming(z, one) == 1
    rng = np.random.default_rng(0)
    for _ in range(200):
        a, b = rng.integers(0, 256, (2, 32), dtype=np.uint8)
        assert O.hamming(a, b) == int(np.unpackbits(a ^ b).sum())


def _hamming_knn(lib, backend):
    rng = np.random.default_rng(7)
    B, nq, nt = 2, 70, 333
    q = rng.integers(0, 256, (B, nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (B, nt, 32), dtype=np.uint8)
    t[:, 40] = t[:, 7]; t[:, 300] = t[:, 7]; q[:, 3] = t[:, 7]   # exact ties: lower train index must win
    m = orbhip.ORBmatcher(lib=lib)
    D = _view_u16(m.DescriptorDistance(to_dev(q, backend), to_dev(t, backend)))
    ref = np.unpackbits(q[:, :, None, :] ^ t[:, None, :, :], axis=-1).sum(-1)
    assert np.array_equal(D.astype(np.int64), ref)
    capq, capt = nq + 9, nt + 31
    qq = np.zeros((B, capq, 32), np.uint8); qq[:, :nq] = q
    tt = np.zeros((B, capt, 32), np.uint8); tt[:, :nt] = t
    nqv = np.array([nq, nq - 5], np.int32); ntv = np.array([nt, 1], np.int32)
    idx, dist = [to_host(x) for x in m.knnMatch2(to_dev(qq, backend), to_dev(nqv, backend), to_dev(tt, backend), to_dev(ntv, backend))]
    for b in range(B):
        oi, od = O.knn2(q[b, :nqv[b]], t[b, :ntv[b]])
        assert np.array_equal(idx[b, :nqv[b]], oi) and np.array_equal(dist[b, :nqv[b]], od)
        assert (idx[b, nqv[b]:] == -1).all()
    assert idx[0, 3, 0] == 7 and idx[0, 3, 1] == 40 and dist[0, 3, 0] == 0 and idx[1, 0, 1] == -1


def test_emu_hamming_and_knn2(emu_lib):
    _hamming_knn(emu_lib, "emu")


@pytest.mark.gpu
def test_hip_hamming_and_knn2(hip_lib):
    _hamming_knn(hip_lib, "hip")


def feature_vector(desc, n_nodes=100):
    """Synthetic DBoW2::FeatureVector: node id -> ascending feature indices (TemplatedVocabulary.h:1163-1170),
    nodes = a hash of a few descriptor bits so that corresponding features mostly share a node."""
    node = ((desc[:, 0] >> 4).astype(np.int64) * 7 + (desc[:, 9] >> 5).astype(np.int64) * 3 + (desc[:, 21] >> 6)) % n_nodes
    ids = np.unique(node)
    start = [0]
    feat = []
    for i in ids:
        f = np.nonzero(node == i)[0]
        feat += f.tolist()
        start.append(len(feat))
    return ids.astype(np.int32), np.array(start, np.int32), np.array(feat, np.int32)


def _bow_sides(S):
    ka, da, kb, db = S["ka"], S["da"], S["kb"], S["db"]
    kid, kst, kfe = feature_vector(da)
    fid, fst, ffe = feature_vector(db)
    okf = dict(desc=da, angle=np.ascontiguousarray(ka["angle"]), node_id=kid, node_start=kst, feat_idx=kfe, n_nodes=len(kid))
    of = dict(desc=db, angle=np.ascontiguousarray(kb["angle"]), node_id=fid, node_start=fst, feat_idx=ffe, n_nodes=len(fid))
    return okf, np.ones(len(ka), np.uint8), of


def _bow_case(lib, backend, nnratio, ori, nleft=-1):
    S = scene()
    rng = np.random.default_rng(3)
    ka, da, kb, db = S["ka"], S["da"], S["kb"], S["db"]
    kid, kst, kfe = feature_vector(da)
    fid, fst, ffe = feature_vector(db)
    fid = fid[fid % 7 != 3]                       # drop some nodes on one side: exercises the lower_bound skips
    keep = np.isin(np.arange(100), fid)
    fid2, fst2, ffe2 = feature_vector(db)
    sel = keep[fid2]
    starts = [0]; feats = []
    for k in np.nonzero(sel)[0]:
        feats += ffe2[fst2[k]:fst2[k + 1]].tolist(); starts.append(len(feats))
    fid, fst, ffe = fid2[sel], np.array(starts, np.int32), np.array(feats, np.int32)
    kvalid = (rng.random(len(ka)) < 0.8).astype(np.uint8)
    okf = dict(desc=da, angle=np.ascontiguousarray(ka["angle"]), node_id=kid, node_start=kst, feat_idx=kfe, n_nodes=len(kid))
    of = dict(desc=db, angle=np.ascontiguousarray(kb["angle"]), node_id=fid, node_start=fst, feat_idx=ffe, n_nodes=len(fid))
    om, on = O.search_by_bow(okf, kvalid, of, nnratio, ori, nleft)
    B = 2

    def slab(d, cap_f, cap_n):
        out = dict(desc=np.zeros((B, cap_f, 32), np.uint8), angle=np.zeros((B, cap_f), np.float32), node_id=np.zeros((B, cap_n), np.int32),
                   node_start=np.zeros((B, cap_n + 1), np.int32), feat_idx=np.zeros((B, cap_f), np.int32), n_nodes=np.full(B, d["n_nodes"], np.int32))
        n = len(d["desc"])
        out["desc"][:, :n] = d["desc"]; out["angle"][:, :n] = d["angle"]; out["node_id"][:, :d["n_nodes"]] = d["node_id"]
        out["node_start"][:, :d["n_nodes"] + 1] = d["node_start"]; out["feat_idx"][:, :len(d["feat_idx"])] = d["feat_idx"]
        return {k: to_dev(v, backend) for k, v in out.items()}
    fslab = slab(of, len(kb) + 2, 110)
    if nleft >= 0:
        fslab["n_left"] = to_dev(np.full(B, nleft, np.int32), backend)
    kv = np.zeros((B, len(ka) + 4), np.uint8); kv[:, :len(ka)] = kvalid
    m = orbhip.ORBmatcher(nnratio, ori, lib=lib)
    fm, nm = [to_host(x) for x in m.SearchByBoW(slab(okf, len(ka) + 4, 128), to_dev(kv, backend), fslab)]
    for b in range(B):
        assert nm[b] == on and np.array_equal(fm[b, :len(kb)], om)
    assert on > 20 or nleft == 0   # Nleft == 0: no left candidates -> bestDist1 stays 256 and nothing is accepted (:453)


@pytest.mark.parametrize("ratio,ori", [(0.7, True), (0.9, False)])
def test_emu_search_by_bow(emu_lib, ratio, ori):
    _bow_case(emu_lib, "emu", ratio, ori)


def test_emu_search_by_bow_fisheye_rig(emu_lib):
    """F.Nleft != -1 (ORBmatcher.cc:411-436, 505-540): features >= Nleft are the right camera's; separate best/second per camera, the
    right match rides on the left best passing TH_LOW and skips the ratio test (`|| true`)."""
    S = scene()
    n_mono = O.search_by_bow(*_bow_sides(S), 0.7, True)[1]
    _bow_case(emu_lib, "emu", 0.7, True, nleft=int(0.55 * len(S["kb"])))
    assert n_mono > 20


@pytest.mark.gpu
@pytest.mark.parametrize("frac", [0.55, 0.0, 1.0])
def test_hip_search_by_bow_fisheye_rig(hip_lib, frac):
    _bow_case(hip_lib, "hip", 0.7, True, nleft=int(frac * len(scene()["kb"])))


@pytest.mark.gpu
@pytest.mark.parametrize("ratio,ori", [(0.7, True), (0.9, False), (0.6, True)])
def test_hip_search_by_bow(hip_lib, ratio, ori):
    _bow_case(hip_lib, "hip", ratio, ori)


# ---- SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)  ORBmatcher.cc:984-1124 (LoopClosing.cc:697) -------------------------------------------
def _drop_nodes(ids, st, fe, keep_mask):
    starts, feats = [0], []
    for k in np.nonzero(keep_mask)[0]:
        feats += fe[st[k]:st[k + 1]].tolist(); starts.append(len(feats))
    return ids[keep_mask], np.array(starts, np.int32), np.array(feats, np.int32)


def _bow_kf_case(lib, backend, nnratio, ori, scene_kw=None, seed=5, drop=(7, 3), valid_frac=(0.8, 0.85), rig_nleft=None):
    """Both sides carry map-point validity; nodes are dropped on BOTH sides (lower_bound skips in both directions); optional rig key frames
    (features >= NLeft are folded into the validity flags like the adapter does)."""
    S = scene(**(scene_kw or {}))
    rng = np.random.default_rng(seed)
    ka, da, kb, db = S["ka"], S["da"], S["kb"], S["db"]
    id1, st1, fe1 = feature_vector(da)
    id2, st2, fe2 = feature_vector(db)
    id1, st1, fe1 = _drop_nodes(id1, st1, fe1, id1 % drop[0] != drop[1])
    id2, st2, fe2 = _drop_nodes(id2, st2, fe2, id2 % (drop[0] + 4) != 1)
    v1 = (rng.random(len(ka)) < valid_frac[0]).astype(np.uint8)
    v2 = (rng.random(len(kb)) < valid_frac[1]).astype(np.uint8)
    if rig_nleft is not None:
        v1[int(rig_nleft * len(ka)):] = 0; v2[int(rig_nleft * len(kb)):] = 0
    o1 = dict(desc=da, angle=np.ascontiguousarray(ka["angle"]), node_id=id1, node_start=st1, feat_idx=fe1, n_nodes=len(id1))
    o2 = dict(desc=db, angle=np.ascontiguousarray(kb["angle"]), node_id=id2, node_start=st2, feat_idx=fe2, n_nodes=len(id2))
    om, on = O.search_by_bow_kf(o1, v1, o2, v2, nnratio, ori)
    B = 2

    def slab(d, cap_f, cap_n):
        out = dict(desc=np.zeros((B, cap_f, 32), np.uint8), angle=np.zeros((B, cap_f), np.float32), node_id=np.zeros((B, cap_n), np.int32),
                   node_start=np.zeros((B, cap_n + 1), np.int32), feat_idx=np.zeros((B, cap_f), np.int32), n_nodes=np.full(B, d["n_nodes"], np.int32))
        n = len(d["desc"])
        out["desc"][:, :n] = d["desc"]; out["angle"][:, :n] = d["angle"]; out["node_id"][:, :d["n_nodes"]] = d["node_id"]
        out["node_start"][:, :d["n_nodes"] + 1] = d["node_start"]; out["feat_idx"][:, :len(d["feat_idx"])] = d["feat_idx"]
        return {k: to_dev(v, backend) for k, v in out.items()}
    c1, c2 = len(ka) + 4, len(kb) + 9
    V1 = np.zeros((B, c1), np.uint8); V1[:, :len(ka)] = v1
    V2 = np.ones((B, c2), np.uint8); V2[:, :len(kb)] = v2     # slack entries flagged valid: they must never be reached through the CSR
    m = orbhip.ORBmatcher(nnratio, ori, lib=lib)
    m12, nm = [to_host(x) for x in m.SearchByBoWKF(slab(o1, c1, 128), to_dev(V1, backend), slab(o2, c2, 110), to_dev(V2, backend))]
    for b in range(B):
        assert nm[b] == on, (nm[b], on)
        assert np.array_equal(m12[b, :len(ka)], om)
        assert (m12[b, len(ka):] == -1).all()
    got = om[om >= 0]
    assert len(np.unique(got)) == len(got)          # vbMatched2: a KF2 feature is taken at most once
    assert v1[om >= 0].all() and v2[got].all()      # only features holding good map points take part
    return on


BOW_KF_CASES = [
    ("ratio07_ori", 0.7, True, None, 5, (7, 3), (0.8, 0.85), None),
    ("ratio09_no_ori", 0.9, False, None, 6, (5, 0), (0.6, 0.95), None),
    ("scene2_ratio08", 0.8, True, dict(W=400, H=300, nf=500, seed=77, shift=(3, 5)), 7, (9, 2), (1.0, 1.0), None),
    ("scene3_sparse_mps", 0.75, True, dict(W=520, H=380, nf=800, seed=91, shift=(-5, 2)), 8, (4, 1), (0.3, 0.4), None),
    ("rig_keyframes", 0.7, True, None, 9, (7, 3), (0.9, 0.9), 0.6),
]


@pytest.mark.parametrize("case", BOW_KF_CASES[:3] + BOW_KF_CASES[4:], ids=lambda c: c[0])
def test_emu_search_by_bow_keyframes(emu_lib, case):
    n = _bow_kf_case(emu_lib, "emu", *case[1:])
    assert n > (20 if case[0] != "scene3_sparse_mps" else 3)


@pytest.mark.gpu
@pytest.mark.parametrize("case", BOW_KF_CASES, ids=lambda c: c[0])
def test_hip_search_by_bow_keyframes(hip_lib, case):
    n = _bow_kf_case(hip_lib, "hip", *case[1:])
    assert n > (20 if case[0] != "scene3_sparse_mps" else 3)


def test_oracle_search_by_bow_keyframes_known_answers():
    """Hand-built node: strict `bestDist1 < TH_LOW` (a distance of exactly 50 is refused, the (KeyFrame, Frame) overload accepts it),
    first minimum wins, and a KF2 feature taken by an earlier KF1 feature is invisible to later ones."""
    def bits(n):
        d = np.zeros(32, np.uint8)
        d[:n // 8] = 255
        if n % 8:
            d[n // 8] = (1 << (n % 8)) - 1
        return d
    z = bits(0)
    d1 = np.stack([z, z])                           # two KF1 features, both in node 4
    d2 = np.stack([bits(50), bits(10), bits(10)])   # KF2: distances 50, 10, 10 to z
    s1 = dict(desc=d1, angle=np.zeros(2, np.float32), node_id=np.array([4], np.int32), node_start=np.array([0, 2], np.int32),
              feat_idx=np.array([0, 1], np.int32), n_nodes=1)
    s2 = dict(desc=d2, angle=np.zeros(3, np.float32), node_id=np.array([4], np.int32), node_start=np.array([0, 3], np.int32),
              feat_idx=np.array([0, 1, 2], np.int32), n_nodes=1)
    ones = lambda n: np.ones(n, np.uint8)
    # feature 0: best 10 (idx 1, first of the equals), second 10 -> ratio test 10 < 0.9*10 fails; feature 1 likewise: no matches
    m, n = O.search_by_bow_kf(s1, ones(2), s2, ones(3), 0.9, False)
    assert n == 0 and m.tolist() == [-1, -1]
    # with KF2 feature 2 invalid: feature 0 takes idx 1 (10 < 0.9*50); feature 1 then only sees idx 0 at distance 50 -> not < TH_LOW
    m, n = O.search_by_bow_kf(s1, ones(2), s2, np.array([1, 1, 0], np.uint8), 0.9, False)
    assert n == 1 and m.tolist() == [1, -1]
    # the (KeyFrame, Frame) overload accepts distance 50 (`<= TH_LOW`, :453): same data, second feature matches idx 0
    fm, fn = O.search_by_bow(s1, ones(2), dict(s2, desc=d2[:2], angle=np.zeros(2, np.float32), node_start=np.array([0, 2], np.int32),
                                               feat_idx=np.array([0, 1], np.int32)), 0.9, False)
    assert fn == 2 and fm.tolist() == [1, 0]
