"""Stage-2 parity at BASELINE.json's stated sizes (GPU tier): the occupancy-dependent paths of the matcher kernels — two queries per wave,
windows wider than 32 grid columns, the resolver's list-length packing, >64-entry candidate lists — only show at 752x480 / 1000 keypoints
(configs[1], configs[2]) and 1280x720 / 1500 keypoints with lapping areas (configs[3]).  Every case runs a batch of DIFFERENT scenes through the
C ABI and compares each frame with the oracle on the same inputs (ORBmatcher.cc:59-255, 323-587, 2244-2509; Frame.cc:955-1133, 1281-1325)."""
import sys

import numpy as np
import pytest

import oracle_lib as O
import orbhip
from orbhip.matcher import MODE_BEST_ONLY, MODE_LOCAL_MAP, QUERY_DTYPE, TH_HIGH
from orbhip.synth import synth_image
from test_matcher_parity import make_queries, scene, to_dev, to_host

BACKEND = "hip"   # the CPU-tier twins of these tests run the same code against the emulated library ("emu")


@pytest.fixture
def emu_backend(emu_lib, monkeypatch):
    monkeypatch.setattr(sys.modules[__name__], "BACKEND", "emu")
    return emu_lib

FULL = [dict(W=752, H=480, nf=1000, seed=140 + i, shift=s) for i, s in enumerate([(6, -4), (-9, 3), (2, 11), (14, 7)])]


def _batch_sbp(lib, scenes, qs, mode, th_dist, nnratio, ori, urs, occs):
    """One orbm_search_by_projection call over len(scenes) different frames (fixed-capacity slabs, ragged counts)."""
    B = len(scenes)
    cap_k = max(len(S["kb"]) for S in scenes) + 7
    cap_q = max(len(q) for q in qs) + 5
    kps = np.zeros((B, cap_k, 7), np.float32); desc = np.zeros((B, cap_k, 32), np.uint8)
    Q = np.zeros((B, cap_q), QUERY_DTYPE); qd = np.zeros((B, cap_q, 32), np.uint8)
    nk = np.zeros(B, np.int32); nq = np.zeros(B, np.int32)
    ur = None if urs is None else np.zeros((B, cap_k), np.float32)
    oc = None if occs is None else np.zeros((B, cap_k), np.uint8)
    for b, (S, q) in enumerate(zip(scenes, qs)):
        n = len(S["kb"])
        kps[b, :n] = S["kb"].view(np.float32).reshape(-1, 7); desc[b, :n] = S["db"]; nk[b] = n
        Q[b, :len(q)] = q; qd[b, :len(q)] = S["da"]; nq[b] = len(q)
        if ur is not None:
            ur[b, :n] = urs[b]
        if oc is not None:
            oc[b, :n] = occs[b]
    d = lambda a: to_dev(a, BACKEND)
    m = orbhip.ORBmatcher(nnratio, ori, lib=lib)
    dk, dn = d(kps), d(nk)
    grid = scenes[0]["grid"]
    gs, gi = m.grid_build(dk, dn, grid)
    qm, km, nm = m.SearchByProjection(dk, d(desc), dn, gs, gi, d(Q.view(np.uint8).reshape(B, cap_q, 28)), d(qd), d(nq), grid, mode, th_dist,
                                      u_right=d(ur), occupied0=d(oc))
    return to_host(gs), to_host(gi), to_host(qm), to_host(km), to_host(nm)


SBP_FULL = [
    ("motion_model_th15", MODE_BEST_ONLY, 15, 0.9, True, False, False),       # Tracking.cc TrackWithMotionModel th = 15 (mono)
    ("motion_model_th7_stereo", MODE_BEST_ONLY, 7, 0.9, True, True, False),   # th = 7 (stereo), u_right gate
    ("motion_model_th30_retry", MODE_BEST_ONLY, 30, 0.9, True, False, True),  # the 2*th retry (Tracking.cc) with occupied keypoints
    ("local_map_th1", MODE_LOCAL_MAP, 1, 0.8, True, False, False),            # SearchLocalPoints th = 1
    ("local_map_th3_stereo_occ", MODE_LOCAL_MAP, 3, 0.8, True, True, True),   # th = 3 (RGB-D / recently relocalised: 5)
    ("local_map_th15_wide", MODE_LOCAL_MAP, 15, 0.8, True, False, True),      # th = 15 after relocalisation: windows up to 36 grid columns wide
]


@pytest.mark.parametrize("case", SBP_FULL[:2] + SBP_FULL[3:5], ids=lambda c: c[0])
def test_emu_search_by_projection_752x480_1000kp(emu_backend, case):
    test_hip_search_by_projection_752x480_1000kp(emu_backend, case, n_scenes=2)


@pytest.mark.gpu
@pytest.mark.parametrize("case", SBP_FULL, ids=lambda c: c[0])
def test_hip_search_by_projection_752x480_1000kp(hip_lib, case, n_scenes=4):
    _, mode, th, nnratio, ori, stereo, occupied = case
    scenes = [scene(**kw) for kw in FULL[:n_scenes]]
    rng = np.random.default_rng(7)
    qs, urs, occs = [], [], []
    for S in scenes:
        qs.append(make_queries(S, mode, th, rng, stereo))
        kb = S["kb"]
        u = (kb["x"] - np.float32(20.0) + rng.normal(0, 6, len(kb))).astype(np.float32)
        u[rng.random(len(kb)) < 0.3] = -1
        urs.append(u)
        occs.append((rng.random(len(kb)) < 0.15).astype(np.uint8))
    gs, gi, qm, km, nm = _batch_sbp(hip_lib, scenes, qs, mode, TH_HIGH, nnratio, ori, urs if stereo else None, occs if occupied else None)
    total = 0
    for b, (S, q) in enumerate(zip(scenes, qs)):
        kb = S["kb"]
        assert 990 <= len(kb) <= 1040 and len(q) >= 990            # the stated size, not a reduced stand-in
        ogs, ogi = O.grid_build(kb, S["grid"])
        oq, ok, on = O.search_by_projection(kb, S["db"], q, S["da"], S["grid"], mode, TH_HIGH, nnratio, ori, urs[b] if stereo else None,
                                            occs[b] if occupied else None)
        assert np.array_equal(gs[b], ogs) and np.array_equal(gi[b, :len(kb)], ogi[:len(kb)]), "grid CSR"
        assert nm[b] == on, (b, nm[b], on)
        assert np.array_equal(km[b, :len(kb)], ok), "mvpMapPoints"
        assert np.array_equal(qm[b, :len(q)], oq), "per-query match"
        total += on
    assert total > n_scenes * 150


def test_emu_compute_bow_search_by_bow_752x480_1000kp_k10_L6(emu_backend):
    test_hip_compute_bow_search_by_bow_752x480_1000kp_k10_L6(emu_backend, n_scenes=2)


@pytest.mark.gpu
def test_hip_compute_bow_search_by_bow_752x480_1000kp_k10_L6(hip_lib, n_scenes=4):
    """configs[2]'s matching step at its stated size: Frame::ComputeBoW on a k = 10, L = 6 vocabulary (the stock ORBvoc shape, 1.1 M nodes,
    levelsup = 4) on the device, then SearchByBoW(KF, F) and SearchByBoW(KF, KF) on the device CSRs, vs the oracle chain."""
    from orbhip.bow import ORBVocabulary, synth_vocabulary_fast
    scenes = [scene(**kw) for kw in FULL[:n_scenes]]
    blob = synth_vocabulary_fast(5, 10, 6, sample_desc=np.concatenate([scenes[0]["da"], scenes[1]["db"]]))
    ov = O.OracleVocabulary(blob)
    V = ORBVocabulary(blob, lib=hip_lib)
    assert (V.k, V.L) == (10, 6) and V.n_nodes > 1_000_000
    B = len(scenes)
    cap = max(max(len(S["ka"]), len(S["kb"])) for S in scenes) + 9
    rng = np.random.default_rng(21)
    d = lambda a: to_dev(a, BACKEND)

    def slabs(key_k, key_d):
        desc = np.zeros((B, cap, 32), np.uint8); ang = np.zeros((B, cap), np.float32); n = np.zeros(B, np.int32)
        for b, S in enumerate(scenes):
            m = len(S[key_k]); desc[b, :m] = S[key_d]; ang[b, :m] = S[key_k]["angle"]; n[b] = m
        return desc, ang, n
    dA, aA, nA = slabs("ka", "da")
    dB, aB, nB = slabs("kb", "db")
    rA = V.transform(d(dA), d(nA), 4)
    rB = V.transform(d(dB), d(nB), 4)
    mk = lambda r, desc, ang: dict(desc=d(desc), angle=d(ang), node_id=r["fv_node_id"], node_start=r["fv_node_start"], feat_idx=r["fv_feat_idx"],
                                   n_nodes=r["fv_n_nodes"])
    vA = np.zeros((B, cap), np.uint8); vB = np.zeros((B, cap), np.uint8)
    for b, S in enumerate(scenes):
        vA[b, :len(S["ka"])] = rng.random(len(S["ka"])) < 0.85
        vB[b, :len(S["kb"])] = rng.random(len(S["kb"])) < 0.9
    m = orbhip.ORBmatcher(0.75, True, lib=hip_lib)
    fm, nm = [to_host(x) for x in m.SearchByBoW(mk(rA, dA, aA), d(vA), mk(rB, dB, aB))]
    m12, nk = [to_host(x) for x in m.SearchByBoWKF(mk(rA, dA, aA), d(vA), mk(rB, dB, aB), d(vB))]
    side = lambda o, k, dsc: dict(desc=dsc, angle=np.ascontiguousarray(k["angle"]), node_id=o["fv_node_id"][:o["fv_n_nodes"]],
                                  node_start=o["fv_node_start"][:o["fv_n_nodes"] + 1], feat_idx=o["fv_feat_idx"][:o["fv_node_start"][o["fv_n_nodes"]]],
                                  n_nodes=o["fv_n_nodes"])
    hA = {k: to_host(v) for k, v in rA.items()}
    tot = 0
    for b, S in enumerate(scenes):
        oa, ob = ov.transform(S["da"], 4), ov.transform(S["db"], 4)
        na = len(S["ka"])
        assert np.array_equal(hA["word_id"][b, :na], oa["word_id"][:na]) and np.array_equal(hA["node_id"][b, :na], oa["node_id"][:na])
        nb = oa["bv_n"]
        assert np.array_equal(hA["bv_value"][b, :nb].view(np.uint64), oa["bv_value"][:nb].view(np.uint64))
        om, on = O.search_by_bow(side(oa, S["ka"], S["da"]), vA[b, :na], side(ob, S["kb"], S["db"]), 0.75, True)
        assert nm[b] == on and np.array_equal(fm[b, :len(S["kb"])], om), b
        ok12, onk = O.search_by_bow_kf(side(oa, S["ka"], S["da"]), vA[b, :na], side(ob, S["kb"], S["db"]), vB[b, :len(S["kb"])], 0.75, True)
        assert nk[b] == onk and np.array_equal(m12[b, :na], ok12), b
        tot += on + onk
    assert tot > 2 * n_scenes * 40


def _stereo_pair(seed, W, H):
    left = synth_image(seed, W, H)
    rng = np.random.default_rng(seed + 9)
    right = np.zeros_like(left)
    for y0, y1, dd in ((0, H // 3, 9), (H // 3, 2 * H // 3, 23), (2 * H // 3, H, 41)):
        right[y0:y1] = np.roll(left[y0:y1], -dd, axis=1)
    right = np.clip(right.astype(np.int32) + rng.integers(-2, 3, right.shape), 0, 255).astype(np.uint8)
    return left, right


@pytest.mark.gpu
def test_hip_compute_stereo_matches_752x480_1000kp(hip_lib):
    """configs[2] (EuRoC stereo) at its stated size: 2 x ORBextractor (nFeatures = 1000) + Frame::ComputeStereoMatches on a batch of 4 different
    rectified pairs; mvuRight / mvDepth bitwise."""
    import torch
    from orbhip.extractor import stereo_matches
    BF, FX = 47.90639384423901, 458.654
    W, H, NF, B = 752, 480, 1000, 4
    pairs = [_stereo_pair(300 + i, W, H) for i in range(B)]
    eL = orbhip.ORBextractor(NF, 1.2, 8, 20, 7, lib=hip_lib, max_batch=B)
    eR = orbhip.ORBextractor(NF, 1.2, 8, 20, 7, lib=hip_lib, max_batch=B)
    dL = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    dR = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    outL, outR = eL.extract_batch(dL, (0, 0)), eR.extract_batch(dR, (0, 0))
    u, dz = stereo_matches(eL, eR, outL, outR, BF / FX, BF)
    u, dz = u.cpu().numpy(), dz.cpu().numpy()
    cl = outL[2].cpu().numpy()
    for b, (left, right) in enumerate(pairs):
        oL, oR = O.OrbOracle(NF, 1.2, 8, 20, 7), O.OrbOracle(NF, 1.2, 8, 20, 7)
        _, kl, dl = oL.extract(left, 0, 0)
        _, kr, dr = oR.extract(right, 0, 0)
        assert cl[b, 0] == len(kl) >= 990
        our, odp = O.stereo_matches(oL, oR, kl, dl, kr, dr, BF / FX, BF)
        assert np.array_equal(u[b, :len(kl)].view(np.uint32), our.view(np.uint32)), "mvuRight (bitwise)"
        assert np.array_equal(dz[b, :len(kl)].view(np.uint32), odp.view(np.uint32)), "mvDepth (bitwise)"
        assert (our >= 0).sum() > 250


@pytest.mark.gpu
def test_hip_stereo_fisheye_1280x720_1500kp_lapping(hip_lib):
    """configs[3] at its stated size: two ORBextractors (1280x720, nFeatures = 1500) with lapping areas + Frame::ComputeStereoFishEyeMatches on what
    they produced (the lapping keypoints [monoIndex, N) are the only stereo candidates), batch of 3 different frames, vs the oracle chain."""
    import torch
    from orbhip.frame import ComputeStereoFishEyeMatches, FisheyeRig
    W, H, NF, B = 1280, 720, 1500, 3
    lapL, lapR = (300, 980), (250, 930)
    kb = [190.978 * 2.5, 190.973 * 2.5, W / 2.0, H / 2.0, 0.0034823894, 0.0007150348, -0.0020532361, 0.00020293673]
    sig2 = (np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2
    rig = FisheyeRig.make(kb, kb, np.eye(3), [0.1, 0.0, 0.0], [float(v) for v in sig2])
    lefts = [synth_image(700 + i, W, H) for i in range(B)]
    rights = [np.roll(l, -(10 + 6 * i), axis=1) for i, l in enumerate(lefts)]
    eL = orbhip.ORBextractor(NF, 1.2, 8, 20, 7, lib=hip_lib, max_batch=B)
    eR = orbhip.ORBextractor(NF, 1.2, 8, 20, 7, lib=hip_lib, max_batch=B)
    dL, dR = torch.from_numpy(np.stack(lefts)).cuda(), torch.from_numpy(np.stack(rights)).cuda()
    oLt, oRt = eL.extract_batch(dL, lapL), eR.extract_batch(dR, lapR)
    cl, cr = oLt[2].view(-1), oRt[2].view(-1)
    out = ComputeStereoFishEyeMatches(oLt[0], oLt[1], cl, cl[1:], oRt[0], oRt[1], cr, cr[1:], rig, lib=hip_lib, count_stride=2)
    l2r, r2l, depth, p3d, nm = [to_host(o) for o in out]
    hkl, hdl, hcl = oLt[0].cpu().numpy(), oLt[1].cpu().numpy(), oLt[2].cpu().numpy()
    tot = 0
    for b in range(B):
        oe = O.OrbOracle(NF, 1.2, 8, 20, 7)
        ml, kl, dl = oe.extract(lefts[b], *lapL)
        mr, kr, dr = oe.extract(rights[b], *lapR)
        nl, nr = len(kl), len(kr)
        assert nl >= 1490 and hcl[b, 0] == nl and hcl[b, 1] == ml and 0 < ml < nl       # stated size, and a real lapping split
        assert np.array_equal(hkl[b, :nl].view(np.uint8), kl.view(np.uint8).reshape(nl, -1)) and np.array_equal(hdl[b, :nl], dl)
        ol2r, or2l, od, op, on = O.stereo_fisheye(kl, dl, ml, kr, dr, mr, rig.as_array(), sig2)
        assert nm[b] == on and np.array_equal(l2r[b, :nl], ol2r) and np.array_equal(r2l[b, :nr], or2l), b
        assert np.allclose(depth[b, :nl], od, rtol=2e-6, atol=0) and np.allclose(p3d[b, :nl], op, rtol=2e-6, atol=1e-7), b
        assert (ol2r[:ml] == -1).all()            # monocular keypoints never take part
        tot += on
    assert tot > 0
