"""k_sbp_resolve_par (parallel fixed-point rounds) on adversarial inputs: contention that makes the dependency chain of the serial walk as long
as the frame has queries, lists deeper than the staged heads, key points shared by accepters without an observed point, frames that must take
the one-wave walk (a list longer than the candidate cache) next to frames that do not — all against the oracle's literal loops
(ORBmatcher.cc:59-255 local map, :2244-2509 motion model)."""
import sys

import numpy as np
import pytest

import oracle_lib as O
import orbhip
from orbhip.matcher import MODE_BEST_ONLY, MODE_LOCAL_MAP, Q_HAS_OBS, Q_VALID, QUERY_DTYPE
from test_matcher_parity import to_dev, to_host

BACKEND = "hip"
GRID = (0.0, 0.0, float(np.float32(64) / np.float32(640)), float(np.float32(48) / np.float32(480)))


@pytest.fixture
def emu_backend(emu_lib, monkeypatch):
    monkeypatch.setattr(sys.modules[__name__], "BACKEND", "emu")
    return emu_lib


def crowded_frame(rng, n_clusters, per_cluster, n_queries, spread=3.0, obs_frac=0.8, near_desc=False, levels=(0, 1), lattice=False):
    """Key points in tight clusters; every query of a cluster sees all of the cluster's key points (lattice: and no other cluster's)."""
    kps = np.zeros(n_clusters * per_cluster, O.KP_DTYPE)
    cx = rng.uniform(60, 580, n_clusters); cy = rng.uniform(60, 420, n_clusters)
    if lattice:
        cells = rng.permutation(24 * 18)[:n_clusters]
        cx = 40.0 + 24.0 * (cells % 24); cy = 30.0 + 24.0 * (cells // 24)
    for c in range(n_clusters):
        s = slice(c * per_cluster, (c + 1) * per_cluster)
        kps["x"][s] = (cx[c] + rng.uniform(-spread, spread, per_cluster)).astype(np.float32)
        kps["y"][s] = (cy[c] + rng.uniform(-spread, spread, per_cluster)).astype(np.float32)
    kps["size"] = 31; kps["angle"] = rng.uniform(0, 360, len(kps)).astype(np.float32); kps["response"] = 50
    kps["octave"] = rng.choice(levels, len(kps)); kps["class_id"] = -1
    order = rng.permutation(len(kps))                     # key point index order is unrelated to position
    kps = kps[order]
    base = rng.integers(0, 256, (n_clusters, 32), dtype=np.uint8)
    desc = rng.integers(0, 256, (len(kps), 32), dtype=np.uint8)
    if near_desc:                                         # descriptors of a cluster a few bits apart: many equal distances (ties -> enumeration order decides)
        cl = (order // per_cluster)
        desc = base[cl].copy()
        flip = rng.integers(0, 256, len(kps)); desc[np.arange(len(kps)), flip % 32] ^= (1 << (flip % 8)).astype(np.uint8)
    q = np.zeros(n_queries, QUERY_DTYPE)
    qc = rng.integers(0, n_clusters, n_queries)
    q["u"] = cx[qc].astype(np.float32); q["v"] = cy[qc].astype(np.float32); q["radius"] = np.float32(spread + 2.0)
    q["min_level"] = 0; q["max_level"] = 7
    q["angle"] = rng.uniform(0, 360, n_queries).astype(np.float32)
    q["flags"] = Q_VALID | Q_HAS_OBS
    q["flags"][rng.random(n_queries) >= obs_frac] &= ~np.uint32(Q_HAS_OBS)
    qd = base[qc].copy() if near_desc else rng.integers(0, 256, (n_queries, 32), dtype=np.uint8)
    return kps, desc, q, qd


def run_frames(lib, frames, mode, th_dist, nnratio, ori, rig_oracle=()):
    B = len(frames)
    cap_k = max(len(f[0]) for f in frames) + 3; cap_q = max(len(f[2]) for f in frames) + 5
    kps = np.zeros((B, cap_k, 7), np.float32); desc = np.zeros((B, cap_k, 32), np.uint8)
    Q = np.zeros((B, cap_q), QUERY_DTYPE); qd = np.zeros((B, cap_q, 32), np.uint8)
    nk = np.zeros(B, np.int32); nq = np.zeros(B, np.int32)
    for b, (k, d, q, qdd) in enumerate(frames):
        kps[b, :len(k)] = k.view(np.float32).reshape(-1, 7); desc[b, :len(k)] = d; nk[b] = len(k)
        Q[b, :len(q)] = q; qd[b, :len(q)] = qdd; nq[b] = len(q)
    dv = lambda a: to_dev(a, BACKEND)
    m = orbhip.ORBmatcher(nnratio, ori, lib=lib)
    dk, dn = dv(kps), dv(nk)
    gs, gi = m.grid_build(dk, dn, GRID)
    work = dv(np.zeros(m._L.orbm_search_workspace_bytes(B, cap_q), np.uint8))
    qm, km, nm = [to_host(x) for x in m.SearchByProjection(dk, dv(desc), dn, gs, gi, dv(Q.view(np.uint8).reshape(B, cap_q, 28)), dv(qd), dv(nq), GRID,
                                                          mode, th_dist, work=work)]
    # the per-frame flag words behind the query rows (orbm_search_workspace_bytes): 1 = k_sbp_frame left the frame to the one-wave walk
    flags = to_host(work)[B * cap_q * 128 * 4:].view(np.int32)[:B].copy()
    for b, (k, d, q, qdd) in enumerate(frames):
        if b in rig_oracle:   # the rig restatement with every key point in the left camera and no stereo links: TWIN semantics (ORBmatcher.cc:166-167, :2332)
            oq, ok, on = O.search_by_projection_rig(k, d, len(k), None, q, qdd, GRID, mode, th_dist, nnratio, ori)
        else:
            oq, ok, on = O.search_by_projection(k, d, q, qdd, GRID, mode, th_dist, nnratio, ori)
        assert nm[b] == on, (b, nm[b], on)
        assert np.array_equal(km[b, :len(k)], ok), ("mvpMapPoints", b)
        assert np.array_equal(qm[b, :len(q)], oq), ("per-query match", b)
        assert np.all(qm[b, len(q):] == -1) and np.all(km[b, len(k):] == -1)
    return nm, flags


CASES = [
    # (name, mode, th_dist, ratio, ori, clusters, per cluster, queries, obs fraction, near-equal descriptors)
    ("chain_40_deep_lists", MODE_BEST_ONLY, 255, 0.9, False, 6, 40, 400, 1.0, False),       # everything accepted: chains as long as a cluster has key points
    ("chain_ties", MODE_BEST_ONLY, 255, 0.9, True, 10, 20, 500, 0.8, True),               # equal distances: enumeration order decides; rotation cull
    ("shared_keypoints", MODE_BEST_ONLY, 255, 0.9, True, 8, 12, 300, 0.3, False),         # most accepters do not block: key points change hands
    ("local_map_ratio_levels", MODE_LOCAL_MAP, 255, 0.8, True, 10, 14, 500, 0.7, True),   # ratio test with levels on the FIRST TWO unblocked entries
    ("local_map_th118", MODE_LOCAL_MAP, 118, 0.97, True, 12, 30, 900, 0.9, False),
    ("over_1024_queries", MODE_BEST_ONLY, 255, 0.9, True, 20, 30, 2300, 0.9, False),      # more queries than the workgroup has threads
    # clusters of at most SBPF_SD key points: the kept lists are complete, so these frames MUST be resolved by the parallel rounds (flags == 0)
    ("par_chain_8", MODE_BEST_ONLY, 255, 0.9, True, 40, 8, 640, 1.0, False),              # 16 queries fight over 8 key points per cluster
    ("par_shared_8", MODE_BEST_ONLY, 255, 0.9, True, 40, 8, 640, 0.4, True),              # ... most of them without an observed point, with ties
    ("par_local_map_7", MODE_LOCAL_MAP, 255, 0.8, True, 60, 7, 900, 0.7, True),
    ("par_over_1024_queries", MODE_LOCAL_MAP, 255, 0.95, True, 150, 8, 2300, 0.8, False),   # cap_q > 2 048: beyond k_sbp_frame, the rounds-1-3 pair (flags stay 0: unused)
    # 1 024 < cap_q <= 2 048: the queries beyond the workgroup's threads keep their lists in LDS (k_sbp_frame<TAIL>)
    ("par_tail_1500", MODE_BEST_ONLY, 255, 0.9, True, 120, 8, 1500, 0.8, True),
    ("tail_deep_1300", MODE_LOCAL_MAP, 255, 0.9, True, 30, 40, 1300, 0.9, False),        # ... and read on in their workspace rows
    ("tail_2040", MODE_BEST_ONLY, 118, 0.9, True, 60, 20, 2040, 0.7, False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_emu_resolver_rounds(emu_backend, case):
    test_hip_resolver_rounds(emu_backend, case, n_frames=2)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_hip_resolver_rounds(hip_lib, case, n_frames=6):
    _, mode, th, ratio, ori, ncl, per, nqr, obs, near = case
    rng = np.random.default_rng(abs(hash(case[0])) % 10000)
    frames = [crowded_frame(rng, ncl, per, nqr, obs_frac=obs, near_desc=near, lattice=case[0].startswith("par_")) for _ in range(n_frames)]
    nm, flags = run_frames(hip_lib, frames, mode, th, ratio, ori)
    assert set(flags.tolist()) <= {0, 1}
    if case[0].startswith("par_"):
        assert flags.sum() == 0, flags
    assert nm.sum() > n_frames * (3 if case[0] == "local_map_ratio_levels" else 20), nm


def test_emu_deep_lists_and_flagged_frames(emu_backend):
    test_hip_deep_lists_and_flagged_frames(emu_backend)


@pytest.mark.gpu
def test_hip_deep_lists_and_flagged_frames(hip_lib):
    """Frame 1 holds clusters of 90 key points: its queries run out of kept candidates and read on in their workspace rows (exact; the frame stays
    with k_sbp_frame).  Clusters of 150 are more than list + row hold: the frame goes to the one-wave walk (flag 1) next to frames that stay.
    An empty frame and a frame without queries ride along.  A frame whose queries carry the
    rig's TWIN flag is not k_sbp_frame's: it raises its flag and is redone by k_sbp_candidates_flagged -> k_sbp_resolve (checked against the
    rig oracle with every key point in the left camera), next to frames that stay parallel."""
    from orbhip.matcher import Q_TWIN
    rng = np.random.default_rng(5)
    f0 = crowded_frame(rng, 5, 20, 200)
    f1 = crowded_frame(rng, 3, 90, 260, spread=4.0)
    f2 = crowded_frame(rng, 6, 10, 150)
    e = crowded_frame(rng, 1, 4, 3)
    f3 = (e[0][:0], e[1][:0], e[2], e[3])                 # no key points
    f4 = (e[0], e[1], e[2][:0], e[3][:0])                 # no queries
    _, fl = run_frames(hip_lib, [f0, f1, f2, f3, f4], MODE_BEST_ONLY, 255, 0.9, True)
    assert fl.sum() == 0, fl
    _, fl = run_frames(hip_lib, [f1, f0, f4, f3, f2], MODE_LOCAL_MAP, 255, 0.8, True)
    assert fl.sum() == 0, fl
    f5 = crowded_frame(rng, 3, 150, 300, spread=5.0, near_desc=True)   # ties: every query of a cluster wants the same key points in the same order
    for mode, ratio in ((MODE_BEST_ONLY, 0.9), (MODE_LOCAL_MAP, 1.0)):   # (ratio 1: equal first and second distances on one level still pass)
        _, fl = run_frames(hip_lib, [f0, f5, f1, f2], mode, 255, ratio, True)
        assert fl.tolist() == [0, 1, 0, 0], fl
    ft = crowded_frame(rng, 12, 8, 300, lattice=True)
    ft[2]["flags"][1::2] |= Q_TWIN                        # every second query: "right-camera twin of the previous one" (left grid: no Q_RIGHT)
    for mode, ratio in ((MODE_BEST_ONLY, 0.9), (MODE_LOCAL_MAP, 0.8)):
        _, fl = run_frames(hip_lib, [f0, ft, f2], mode, 255, ratio, True, rig_oracle=(1,))
        assert fl.tolist() == [0, 1, 0], fl


