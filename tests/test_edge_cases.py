"""Empty and degenerate inputs through the C ABI (the reference's own edge cases: frames without keypoints, windows without candidates,
key frames without shared vocabulary nodes, vocabularies hit by zero features).  Nothing may crash; counts must be zero / arrays -1."""
import numpy as np
import pytest

import orbhip
from orbhip.bow import ORBVocabulary, synth_vocabulary
from orbhip.lba import POSE_EDGE_DTYPE, pose_optimization, synth_pose_frames
from orbhip.matcher import MODE_BEST_ONLY, MODE_INIT, MODE_LOCAL_MAP, Q_VALID, QUERY_DTYPE, TRI_PAIR_DTYPE
from test_matcher_parity import to_dev, to_host
from orbhip._lib import OrbHipError as _OrbHipError

GRID = (0.0, 0.0, 64 / 640.0, 48 / 480.0)


def _run(lib, backend):
    d = lambda a: to_dev(a, backend)
    m = orbhip.ORBmatcher(0.8, True, lib=lib)
    B, ck, cq = 3, 16, 9
    rng = np.random.default_rng(0)
    kps = np.zeros((B, ck, 7), np.float32)
    kps[..., 0] = rng.uniform(20, 600, (B, ck)); kps[..., 1] = rng.uniform(20, 440, (B, ck))
    desc = rng.integers(0, 256, (B, ck, 32), dtype=np.uint8)
    nk = np.array([0, ck, 5], np.int32)              # frame 0 has no keypoints at all
    q = np.zeros((B, cq), QUERY_DTYPE)
    q["u"] = 300; q["v"] = 200; q["radius"] = 50; q["min_level"] = -1; q["max_level"] = -1; q["flags"] = Q_VALID
    q["flags"][1, 3:] = 0                                  # invalid queries
    q["u"][2] = -5000                                      # windows entirely outside the grid
    nq = np.array([cq, cq, 0], np.int32)             # frame 2 has no queries
    qd = rng.integers(0, 256, (B, cq, 32), dtype=np.uint8)
    dk, dn = d(kps), d(nk)
    gs, gi = m.grid_build(dk, dn, GRID)
    gs = to_host(gs)
    assert gs[0].max() == 0 and gs[1][-1] == ck and gs[2][-1] == 5
    gs, gi = m.grid_build(dk, dn, GRID)
    dq = d(q.view(np.uint8).reshape(B, cq, 28))
    for mode in (MODE_LOCAL_MAP, MODE_BEST_ONLY, MODE_INIT):
        qm, km, nm = [to_host(x) for x in m.SearchByProjection(dk, d(desc), dn, gs, gi, dq, d(qd), d(nq), GRID, mode, 256)]
        assert nm[0] == 0 and nm[2] == 0 and (qm[0] == -1).all() and (qm[2] == -1).all() and (km[0] == -1).all() and (km[2] == -1).all()
        assert nm[1] == (qm[1] >= 0).sum() and (qm[1, 3:] == -1).all()
    qm, qdist, nf = [to_host(x) for x in m.Fuse(dk, d(desc), dn, gs, gi, dq, d(qd), d(nq), GRID, th_dist=256)]
    assert nf[0] == 0 and nf[2] == 0 and (qm[0] == -1).all() and (qdist[0] == 256).all() and nf[1] == (qm[1] >= 0).sum()
    # BoW search / triangulation with no nodes on one side and with disjoint node sets
    side = lambda n_nodes, ids: dict(kps=d(kps), desc=d(desc), angle=d(np.zeros((B, ck), np.float32)), u_right=None,
                                     has_mp=d(np.zeros((B, ck), np.uint8)), node_id=d(np.tile(np.array(ids, np.int32), (B, 1))),
                                     node_start=d(np.tile(np.arange(5, dtype=np.int32) * 4, (B, 1))), feat_idx=d(np.tile(np.arange(ck, dtype=np.int32), (B, 1))),
                                     n_nodes=d(np.array(n_nodes, np.int32)))
    a, b = side([4, 0, 4], [1, 2, 3, 4]), side([4, 4, 4], [7, 8, 9, 10])
    fm, nm = [to_host(x) for x in m.SearchByBoW(a, d(np.ones((B, ck), np.uint8)), b)]
    assert (nm == 0).all() and (fm == -1).all()
    pairs = np.zeros(B, TRI_PAIR_DTYPE)
    m12, nm = [to_host(x) for x in m.SearchForTriangulation(a, b, d(pairs.view(np.uint8).reshape(B, -1)), False, True)]
    assert (nm == 0).all() and (m12 == -1).all()
    # knn with an empty train set
    idx, dist = [to_host(x) for x in m.knnMatch2(d(qd), d(nq), d(desc), d(np.zeros(B, np.int32)))]
    assert (idx == -1).all() and (dist == 256).all()
    # vocabulary: frames without features
    V = ORBVocabulary(synth_vocabulary(1, 4, 2), lib=lib)
    r = {k: to_host(v) for k, v in V.transform(d(desc), d(np.array([0, ck, 1], np.int32)), 1).items()}
    assert r["bv_n"][0] == 0 and r["fv_n_nodes"][0] == 0 and r["fv_node_start"][0, 0] == 0
    assert r["bv_n"][2] <= 1 and r["fv_n_nodes"][2] <= 1
    # PoseOptimization: frames with 0, 1, 2 correspondences return 0 and leave the pose untouched
    f = synth_pose_frames(seed=2, batch=3, n_pts=20, kind="mono")
    f["n_edges"][:] = [0, 1, 2]
    out, outl, ng = [to_host(x) for x in pose_optimization(d(f["poses"]), d(f["edges"].view(np.uint8).reshape(3, -1)), d(f["n_edges"]),
                                                           d(np.ascontiguousarray(f["cameras"]).view(np.uint8)), lib=lib)]
    assert (ng == 0).all() and np.array_equal(out, f["poses"]) and (outl == 0).all()


def test_emu_empty_and_degenerate_inputs(emu_lib):
    _run(emu_lib, "emu")


@pytest.mark.gpu
def test_hip_empty_and_degenerate_inputs(hip_lib):
    _run(hip_lib, "hip")


def _host_alloc_round_trip(lib):
    """orb_host_alloc / orb_host_free (include/orbhip.h): a page-locked host block is written, sent to the device and read back through the helpers."""
    import ctypes as C
    import numpy as np
    for f in (lib.orb_host_alloc, lib.orb_host_free, lib.orb_dev_alloc, lib.orb_dev_free, lib.orb_memcpy_h2d, lib.orb_memcpy_d2h, lib.orb_stream_sync):
        f.restype = C.c_int
    lib.orb_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    lib.orb_host_free.argtypes = [C.c_void_p]
    lib.orb_dev_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.orb_dev_free.argtypes = [C.c_void_p]
    lib.orb_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.orb_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.orb_stream_sync.argtypes = [C.c_void_p]
    n = 100000
    hp, hq, dp = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.orb_host_alloc(n, C.byref(hp)) == 0 and lib.orb_host_alloc(n, C.byref(hq)) == 0 and lib.orb_dev_alloc(0, n, C.byref(dp)) == 0
    assert lib.orb_host_alloc(n, None) == -3                     # ORB_E_INVALID
    src = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), (n,))
    dst = np.ctypeslib.as_array(C.cast(hq, C.POINTER(C.c_uint8)), (n,))
    src[:] = np.random.default_rng(3).integers(0, 256, n, dtype=np.uint8)
    dst[:] = 0
    assert lib.orb_memcpy_h2d(dp, hp, n, None) == 0 and lib.orb_memcpy_d2h(hq, dp, n, None) == 0 and lib.orb_stream_sync(None) == 0
    assert np.array_equal(src, dst)
    assert lib.orb_host_free(hp) == 0 and lib.orb_host_free(hq) == 0 and lib.orb_dev_free(dp) == 0 and lib.orb_host_free(None) == 0


def test_emu_host_alloc_round_trip(emu_lib):
    _host_alloc_round_trip(emu_lib)


@pytest.mark.gpu
def test_hip_host_alloc_round_trip(hip_lib):
    _host_alloc_round_trip(hip_lib)


def test_emu_resize_staging_footprints_cover_their_tiles(emu_lib):
    """orbx_create checks every tabulated k_resize2 staging footprint against the exact cv::resize index tables (coverage of both taps of every column / row of the
    tile, LDS tile capacity) and refuses the configuration otherwise.  Host code only: a sweep over image sizes, scale factors and level counts must never hit it."""
    import ctypes as C
    from orbhip._lib import OrbxConfig
    rng = np.random.default_rng(2024)
    created = 0
    for i in range(3000):
        W, H = int(rng.integers(80, 2600)), int(rng.integers(80, 1700))
        sf = float(rng.uniform(1.01, 1.3)) if i % 3 == 0 else float(rng.choice([1.1, 1.2, 1.25, 1.3])) if i % 3 == 1 else float(rng.uniform(1.3, 1.6))
        cfg = OrbxConfig(1000, sf, int(rng.integers(2, 11)), 20, 7)
        h = C.c_void_p()
        rc = emu_lib.orbx_create(C.byref(cfg), W, H, 1, 0, C.byref(h))
        if rc == 0:
            created += 1
            emu_lib.orbx_destroy(h)
        else:
            assert "footprint" not in (emu_lib.orbx_last_error(None) or b"").decode(), (W, H, sf)
    assert created > 1500


@pytest.mark.gpu
def test_hip_stage_events_are_opt_in(hip_lib):
    """orbx_enable_timing (round 6): a batch call records its five stage events only while timing is on — orbx_last_timing refuses (ORB_E_INVALID) before
    any timed call and again after timing is switched off; with it on the four stage intervals are positive and add up to the total.  The key points do
    not depend on it."""
    import torch
    from orbhip.synth import synth_image
    frames = torch.from_numpy(np.stack([synth_image(70 + i, 320, 240) for i in range(4)])).cuda()
    e = orbhip.ORBextractor(300, 1.2, 5, 20, 7, device=0, max_batch=4, lib=hip_lib)
    out0 = [t.clone() for t in e.extract_batch(frames, (0, 1000))]
    torch.cuda.synchronize()
    with pytest.raises(_OrbHipError):
        e.last_timing()
    e.enable_timing(True)
    out1 = e.extract_batch(frames, (0, 1000))
    t = e.last_timing()
    assert all(t[k] > 0 for k in ("pyramid", "fast", "octree", "describe")) and abs(t["total"] - (t["pyramid"] + t["fast"] + t["octree"] + t["describe"])) < 0.05 * t["total"] + 0.02
    assert torch.equal(out0[2], out1[2])
    for b in range(4):                                   # (the slots past a frame's count are never written)
        n = int(out0[2][b, 0])
        assert n > 50 and torch.equal(out0[0][b, :n].contiguous().view(torch.uint8), out1[0][b, :n].contiguous().view(torch.uint8)) and torch.equal(out0[1][b, :n], out1[1][b, :n])   # (bytes: class_id = -1 reads as NaN)
    e.enable_timing(False)
    e.extract_batch(frames, (0, 1000))
    with pytest.raises(_OrbHipError):
        e.last_timing()
