"""Stage-1 parity: the HIP extractor vs the oracle.

`backend` = "emu" (product kernels compiled against tests/emu: logic check on CPU, `-m "not gpu"`) or
"hip" (the real liborbhip.so through the C ABI on an MI355X, `-m gpu`).  Bar: bit-exact."""
import numpy as np
import pytest

import oracle_lib as O
import orbhip
from orbhip.synth import flat_image, low_contrast_image, synth_image

CASES = [
    # (name, image factory, nfeatures, lapping, extra ctor args)
    ("euroc_752x480", lambda: synth_image(0), 1000, (0, 1000), {}),
    ("lapping_window", lambda: synth_image(3), 1000, (300, 500), {}),
    ("no_lapping", lambda: synth_image(2), 1000, (0, 0), {}),
    ("ini_extractor_5x", lambda: synth_image(4), 5000, (0, 1000), {}),
    ("low_contrast_retries", lambda: low_contrast_image(5), 1000, (0, 1000), {}),
    ("flat_zero_keypoints", lambda: flat_image(), 1000, (0, 1000), {}),
    ("tumvi_1280x720", lambda: synth_image(6, 1280, 720), 1500, (0, 0), {}),
    ("sparse_640x480", lambda: synth_image(7, 640, 480, n_rect=10, n_disc=5), 1000, (100, 200), {}),
    ("other_config", lambda: synth_image(8, 400, 300), 500, (0, 0), dict(nlevels=4, iniThFAST=30, minThFAST=10, scaleFactor=1.5)),
    ("tumvi_512", lambda: synth_image(9, 512, 512), 1500, (0, 511), {}),
    # white noise: > 4096 FAST corners per tile -> the kernel's whole-tile fallback path; also the densest octree input
    ("noise_tile_overflow", lambda: np.random.default_rng(77).integers(0, 256, (300, 400), dtype=np.uint8), 1000, (0, 1000), dict(nlevels=4)),
]
EMU_CASES = {"noise_tile_overflow", "euroc_752x480", "lapping_window", "low_contrast_retries", "flat_zero_keypoints", "other_config", "sparse_640x480"}


def _run_case(lib, case, stage_checks=True):
    name, mk, nf, lap, kw = case
    img = mk()
    cfg = dict(scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7)
    cfg.update(kw)
    o = O.OrbOracle(nf, cfg["scaleFactor"], cfg["nlevels"], cfg["iniThFAST"], cfg["minThFAST"])
    mono, k, d = o.extract(img, *lap)
    e = orbhip.ORBextractor(nf, cfg["scaleFactor"], cfg["nlevels"], cfg["iniThFAST"], cfg["minThFAST"], lib=lib)
    m2, k2, d2 = e(img, None, lap)
    if stage_checks:
        for l in range(cfg["nlevels"]):
            assert np.array_equal(o.level_image(l), e.pyramid_level(l)), "pyramid level %d" % l
            ca = set(map(tuple, o.level_candidates(l).tolist()))
            cb = e.debug_candidates(l)
            assert len(cb) == len(ca) and set(map(tuple, cb.tolist())) == ca, "FAST candidates level %d" % l
            ka, _ = o.level_keypoints(l)
            ka = np.stack([ka["x"] - 16, ka["y"] - 16, ka["response"]], 1).astype(np.int32) if len(ka) else np.zeros((0, 3), np.int32)
            assert np.array_equal(ka, e.debug_selected(l)), "octree selection/order level %d" % l
    assert m2 == mono
    assert len(k2) == len(k)
    assert np.array_equal(k.view(np.uint8), k2.view(np.uint8)), "keypoints (bitwise)"
    assert np.array_equal(d, d2), "descriptors"
    return len(k)


@pytest.mark.parametrize("case", [c for c in CASES if c[0] in EMU_CASES], ids=lambda c: c[0])
def test_emulated_kernels_match_oracle(emu_lib, case):
    _run_case(emu_lib, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_hip_matches_oracle(hip_lib, case):
    _run_case(hip_lib, case)


def test_emulated_fast_tile_fallback_path():
    """FAST_Q2CAP=48 forces the whole-tile fallback of k_fast in almost every tile; results must not change."""
    import ctypes
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("FAST_Q2CAP=48",), tag="q2cap48")))
    for case in CASES:
        if case[0] in ("sparse_640x480", "noise_tile_overflow"):
            _run_case(lib, case)


def test_emulated_fast_row_group_compaction_path():
    """k_fast fills a wave's pre-test queue once per tile; when a wave's survivors do not fit its queue slice it goes row group by row group.
    FAST_FITS_MAX=96 sends every wave with more than 96 survivors that way (each group queues behind the score bytes parked so far), so ordinary
    tiles take that path too; results must not change."""
    import ctypes
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("FAST_FITS_MAX=96",), tag="fits96")))
    for case in CASES:
        if case[0] in ("sparse_640x480", "noise_tile_overflow", "euroc_752x480"):
            _run_case(lib, case)


def test_emulated_fast_two_cell_row_tiles():
    """Calls with >= 8 frames run k_fast on tiles of two cell rows (score map aliased onto the image tile, one blank map row between the cell rows,
    per-cell retry counters per (cell row, cell)); FAST_TALL_MIN_BATCH=1 sends the single-frame cases through them on the emulator — also combined
    with the whole-tile fallback (FAST_Q2CAP=48: the in-place score ring) and the row-group compaction (FAST_FITS_MAX=96).  Results must not change."""
    import ctypes
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("FAST_TALL_MIN_BATCH=1", "FAST_TWO_PASS_MIN_BATCH=1"), tag="tall1")))
    for case in CASES:
        if case[0] in EMU_CASES:
            _run_case(lib, case)
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("FAST_TALL_MIN_BATCH=1", "FAST_TWO_PASS_MIN_BATCH=1", "FAST_Q2CAP=48", "FAST_FITS_MAX=96"), tag="tall1_q2cap48_fits96")))
    for case in CASES:
        if case[0] in ("sparse_640x480", "noise_tile_overflow", "euroc_752x480"):
            _run_case(lib, case)


def test_emulated_fast_two_threshold_passes():
    """Batches run k_fast twice: instance <0> detects at iniThFAST and lists, per tile, the cells that hold no local maximum; instance <1> detects the
    listed cells again at min(ini, min) — as a one-cell tile when one cell is listed, else the whole tile reporting the listed cells only.
    FAST_TALL_MIN_BATCH=1 FAST_TWO_PASS_MIN_BATCH=1 send single frames that way on the emulator; a corner list of 192 / 96 entries makes tiles that pass at 20 overflow at 7
    (the in-place fallback under a partial cell mask) or overflow already at 20 (every cell listed, nothing emitted by the first pass).  Thresholds
    20/7 (the reference's), 40/5 (many empty cells) and 7/7, 5/9 (ini <= min: one pass).  Results must not change."""
    import ctypes
    import build_emu
    from orbhip import _lib
    imgs = [synth_image(31, 400, 300, n_rect=25, n_disc=10, noise=1.0), synth_image(32, 333, 251, n_rect=120, n_disc=60, noise=3.0, contrast=0.5)]
    for defines, tag in ((("FAST_TALL_MIN_BATCH=1", "FAST_TWO_PASS_MIN_BATCH=1"), "tall1"),
                         (("FAST_TALL_MIN_BATCH=1", "FAST_TWO_PASS_MIN_BATCH=1", "FAST_Q2CAP=192"), "tall1_q2cap192"),
                         (("FAST_TALL_MIN_BATCH=1", "FAST_TWO_PASS_MIN_BATCH=1", "FAST_Q2CAP=96"), "tall1_q2cap96")):
        lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=defines, tag=tag)))
        for ini, mn in ((20, 7), (40, 5), (7, 7), (5, 9)) if tag == "tall1" else ((20, 7), (40, 5)):
            for img in imgs if tag != "tall1_q2cap96" else imgs[1:]:
                o = O.OrbOracle(500, 1.2, 6, ini, mn)
                mono, k, d = o.extract(img)
                e = orbhip.ORBextractor(500, 1.2, 6, ini, mn, lib=lib)
                m2, k2, d2 = e(img)
                assert m2 == mono and len(k) == len(k2), (tag, ini, mn, len(k), len(k2))
                assert np.array_equal(k.view(np.uint8), k2.view(np.uint8)) and np.array_equal(d, d2), (tag, ini, mn)


def test_emulated_fast_second_pass_walks_the_list_with_a_stride():
    """k_fast<1> runs on a 1-D grid sized from the previous call's list and walks the retry list with a workgroup stride (a barrier between the tiles
    of one workgroup: the next tile rewrites the LDS the last phase of this one reads).  The product's grid is never smaller than the list it was
    sized for, so FAST_P1_GRID=3 forces it: three workgroups take every listed tile of the call, several each.  Results must not change."""
    import ctypes
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("FAST_TALL_MIN_BATCH=1", "FAST_TWO_PASS_MIN_BATCH=1", "FAST_P1_GRID=3", "FAST_TWO_PASS_MAX_LISTED=1.0"), tag="tall1_p1grid3_alwaystwo")))
    imgs = [synth_image(31, 400, 300, n_rect=25, n_disc=10, noise=1.0), synth_image(33, 480, 360, n_rect=60, n_disc=20, noise=2.0, contrast=0.7)]
    for ini, mn in ((20, 7), (40, 5)):
        for img in imgs:
            o = O.OrbOracle(500, 1.2, 6, ini, mn)
            mono, k, d = o.extract(img)
            e = orbhip.ORBextractor(500, 1.2, 6, ini, mn, lib=lib)
            for _ in range(2):   # (the second call sees the first one's list length)
                m2, k2, d2 = e(img)
                assert e.last_fast_passes()["two_pass"] == 1 and e.last_fast_passes()["listed"] > 6, e.last_fast_passes()
                assert m2 == mono and len(k) == len(k2), (ini, mn, len(k), len(k2))
                assert np.array_equal(k.view(np.uint8), k2.view(np.uint8)) and np.array_equal(d, d2), (ini, mn)


def test_emulated_fast_pass_policy_never_changes_results():
    """Which form a batch call takes is decided from the listed share of the handle's previous two-pass call: a sparsely textured frame (most tiles
    have an empty cell) sends the second call down the one-pass form, a textured one keeps the two passes — and the key points are the same either
    way (orbx_last_fast_passes reports the decision)."""
    import ctypes
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("FAST_TALL_MIN_BATCH=1", "FAST_TWO_PASS_MIN_BATCH=1"), tag="tall1")))
    sparse = synth_image(41, 400, 300, n_rect=6, n_disc=2, noise=0.5)
    dense = synth_image(42, 400, 300, n_rect=400, n_disc=200, noise=2.0)
    for img, expect_second in ((sparse, 0), (dense, 1)):
        o = O.OrbOracle(400, 1.2, 5, 20, 7)
        mono, k, d = o.extract(img)
        e = orbhip.ORBextractor(400, 1.2, 5, 20, 7, lib=lib)
        modes = []
        for _ in range(3):
            m2, k2, d2 = e(img)
            modes.append(e.last_fast_passes())
            assert m2 == mono and np.array_equal(k.view(np.uint8), k2.view(np.uint8)) and np.array_equal(d, d2)
        assert modes[0]["two_pass"] == 1 and 0 < modes[0]["tiles"]
        assert modes[1]["two_pass"] == expect_second, modes


def test_emulated_octree_lds_key_cache_path():
    """The octree keeps a level's candidates in an LDS cache for small batches (OCT_KEYCAP keys; bigger levels and big batches read them from
    global memory).  A 1500-key cache is hit by the small levels and missed by the big ones, so both paths run inside one extraction.
    Results must not change."""
    import ctypes
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("OCT_KEYCAP=1500",), tag="keycap1500")))
    for case in CASES:
        if case[0] in ("euroc_752x480", "noise_tile_overflow"):
            _run_case(lib, case)


def test_emulated_octree_1024_threads_per_problem():
    """The single-frame entry point runs k_octree with 1 024 threads per (frame, level) problem on the GPU (OCT_T_SINGLE; the emulated builds of this
    tier keep 256, the batch instantiation, because the emulator pays per work-item): the 1 024-thread instantiation on the densest octree input
    and on a lapping window.  Results must not change."""
    import ctypes
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu.build(defines=("OCT_T_SINGLE=1024",), tag="octt1024")))
    for case in CASES:
        if case[0] in ("noise_tile_overflow", "other_config"):
            _run_case(lib, case)


def test_empty_image_returns_minus_one(emu_lib):
    e = orbhip.ORBextractor(1000, 1.2, 8, 20, 7, lib=emu_lib)
    mono, k, d = e(np.zeros((0, 0), np.uint8))
    assert mono == -1 and len(k) == 0 and d.shape == (0, 32)


def test_getters_match_oracle(emu_lib):
    e = orbhip.ORBextractor(1000, 1.2, 8, 20, 7, lib=emu_lib)
    t = O.OrbOracle(1000, 1.2, 8, 20, 7).tables()
    assert e.GetLevels() == 8 and e.GetScaleFactor() == np.float32(1.2)
    assert np.array_equal(e.GetScaleFactors(), t["scale"]) and np.array_equal(e.GetInverseScaleFactors(), t["inv_scale"])
    assert np.array_equal(e.GetScaleSigmaSquares(), t["sigma2"]) and np.array_equal(e.GetInverseScaleSigmaSquares(), t["inv_sigma2"])
    assert np.array_equal(e.features_per_level(), t["nfeat"])


def test_too_small_image_is_rejected(emu_lib):
    e = orbhip.ORBextractor(1000, 1.2, 8, 20, 7, lib=emu_lib)
    with pytest.raises(orbhip.OrbHipError):
        e(np.zeros((60, 80), np.uint8))  # level 7 would have no FAST cell (reference divides by zero there)


def test_bordered_pyramid_matches_reference_layout(emu_lib):
    img = synth_image(12, 320, 240, n_rect=60, n_disc=30)
    o = O.OrbOracle(300)
    o.extract(img, 0, 0)
    e = orbhip.ORBextractor(300, 1.2, 8, 20, 7, lib=emu_lib)
    e(img)
    for l in (0, 3, 7):
        assert np.array_equal(o.level_bordered(l), e.pyramid_level(l, border=19))


@pytest.mark.gpu
@pytest.mark.parametrize("B", [6, 10, 66])   # one-row tiles / two-row tiles in one pass / two-row tiles in two passes (k_fast)
def test_hip_batch_matches_oracle_and_is_deterministic(hip_lib, B):
    import torch
    imgs = np.stack([synth_image(20 + i) for i in range(B - 2)] + [flat_image(), low_contrast_image(31)])
    e = orbhip.ORBextractor(1000, 1.2, 8, 20, 7, lib=hip_lib)
    dev = torch.from_numpy(imgs).cuda()
    outs = []
    for rep in range(2):
        kps, desc, counts = e.extract_batch(dev, (0, 1000))
        torch.cuda.synchronize()
        outs.append((kps.cpu().numpy().copy(), desc.cpu().numpy().copy(), counts.cpu().numpy().copy()))
    o = O.OrbOracle(1000)
    for b in range(B):
        mono, k, d = o.extract(imgs[b], 0, 1000)
        n = outs[0][2][b, 0]
        assert n == len(k) and outs[0][2][b, 1] == mono
        assert np.array_equal(outs[0][0][b, :n].view(np.uint8).reshape(-1), k.view(np.uint8).reshape(-1))
        assert np.array_equal(outs[0][1][b, :n], d)
        assert np.array_equal(outs[0][0][b, :n].view(np.uint8), outs[1][0][b, :n].view(np.uint8)) and np.array_equal(outs[0][1][b, :n], outs[1][1][b, :n])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sparse_1280x720", "textured_752x480"])
def test_hip_fast_pass_policy_never_changes_results(hip_lib, kind):
    """A handle keeps the two-pass form of k_fast while few tiles are listed for the second pass and drops to one pass otherwise (probing again every
    16th call).  Twenty consecutive batch calls on the same frames: every call's output is the oracle's, whatever form it ran in; sparse frames end up
    in the one-pass form, textured ones keep the two passes."""
    import torch
    if kind == "sparse_1280x720":
        W, H, nf = 1280, 720, 1500
        imgs = np.stack([synth_image(300 + i, W, H, n_rect=40, n_disc=10) for i in range(8)] * 8)   # 64 frames: the two-pass form starts there
    else:
        W, H, nf = 752, 480, 1000
        imgs = np.stack([synth_image(310 + i, W, H, n_rect=900, n_disc=400) for i in range(8)] * 8)
    e = orbhip.ORBextractor(nf, 1.2, 8, 20, 7, lib=hip_lib)
    dev = torch.from_numpy(imgs).cuda()
    o = O.OrbOracle(nf)
    ref = [o.extract(imgs[b]) for b in (0, 63)]
    modes = []
    for call in range(20):
        kps, desc, counts = [t.cpu().numpy() for t in e.extract_batch(dev, (0, 0))]
        modes.append(e.last_fast_passes()["two_pass"])
        for b, (mono, k, d) in zip((0, 63), ref):
            n = counts[b, 0]
            assert n == len(k) and counts[b, 1] == mono, (call, b)
            assert np.array_equal(kps[b, :n].view(np.uint8).reshape(-1), k.view(np.uint8).reshape(-1)) and np.array_equal(desc[b, :n], d), (call, b)
    assert modes[0] == 1
    if kind == "sparse_1280x720":
        assert modes[2] == 0 and modes[16] == 1, modes   # one pass once the listed share is known; the 17th call probes
    else:
        assert all(modes), modes


@pytest.mark.gpu
def test_hip_full_size_batch_properties(hip_lib):
    """BASELINE-size batch: size-independent properties (counts, bounds, frame independence under permutation)."""
    import torch
    B = 64
    base = [synth_image(100 + i) for i in range(8)]
    imgs = np.stack([np.roll(base[i % 8], (3 * (i // 8), 5 * (i // 8)), (0, 1)) for i in range(B)])
    e = orbhip.ORBextractor(1000, 1.2, 8, 20, 7, lib=hip_lib)
    dev = torch.from_numpy(imgs).cuda()
    kps, desc, counts = [t.cpu().numpy().copy() for t in e.extract_batch(dev, (0, 0))]
    perm = np.random.default_rng(0).permutation(B)
    kps2, desc2, counts2 = [t.cpu().numpy().copy() for t in e.extract_batch(torch.from_numpy(imgs[perm]).cuda(), (0, 0))]
    assert np.array_equal(counts[perm], counts2)
    for j, b in enumerate(perm):
        n = counts[b, 0]
        assert 990 <= n <= 1032 and counts[b, 1] == n
        assert np.array_equal(kps[b, :n].view(np.uint8), kps2[j, :n].view(np.uint8)) and np.array_equal(desc[b, :n], desc2[j, :n])
        k = kps[b, :n]
        lvl = k[:, 5].view(np.int32)
        assert (np.diff(lvl) >= 0).all() and lvl.min() >= 0 and lvl.max() <= 7
        assert (k[:, 0] >= 19).all() and (k[:, 0] <= 752 - 19).all() and (k[:, 1] >= 19).all() and (k[:, 1] <= 480 - 19).all()
        assert ((k[:, 3] >= 0) & (k[:, 3] < 360)).all()
    o = O.OrbOracle(1000)
    for b in (0, 17, 63):
        mono, k, d = o.extract(imgs[b], 0, 0)
        assert np.array_equal(kps[b, :len(k)].view(np.uint8).reshape(-1), k.view(np.uint8).reshape(-1)) and np.array_equal(desc[b, :len(k)], d)


def test_emulated_batch_frame_order_and_heavy_octree_pass():
    """Batches hand their workgroups out heaviest frame first (k_frame_order: the previous call's FAST candidate counts; it also clears the
    counters) and give (frame, level) problems above OCT_HEAVY_MIN candidates to a second k_octree launch.  OCT_HEAVY_MIN=300 makes ordinary
    levels "heavy" on small images.  Four frames of very different density (noise, flat, textured, sparse), three calls on one handle: the first
    runs on cleared counters (identity order), the second ordered, the third also with the heavy launch — every frame of every call equals the
    oracle's single-frame result, and a different batch afterwards (stale order and counts) does too."""
    import ctypes as C
    import build_emu
    from orbhip import _lib
    lib = _lib.bind(C.CDLL(build_emu.build(defines=("OCT_HEAVY_MIN=300",), tag="octheavy300")))
    lib.orbx_last_schedule.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    W, H, nf = 240, 200, 300
    rng = np.random.default_rng(9)
    frames = np.stack([rng.integers(0, 256, (H, W), dtype=np.uint8), flat_image(W, H, 90), synth_image(61, W, H, n_rect=60, n_disc=30),
                       synth_image(62, W, H, n_rect=6, n_disc=3)])
    e = orbhip.ORBextractor(nf, 1.2, 5, 20, 7, lib=lib)
    h = e._handle(W, H, max_batch=4)
    cap = lib.orbx_max_keypoints(h)
    o = O.OrbOracle(nf, 1.2, 5, 20, 7)

    def run(fr):
        B = len(fr)
        kps = np.zeros((B, cap), O.KP_DTYPE); desc = np.zeros((B, cap, 32), np.uint8); cnt = np.zeros((B, 2), np.int32)
        rc = lib.orbx_extract_batch_dev(h, _lib.ptr(fr), B, W * H, W, 0, 1000, _lib.ptr(kps), _lib.ptr(desc), cap, _lib.ptr(cnt), None)
        assert rc == 0
        for b in range(B):
            mono, k, d = o.extract(fr[b], 0, 1000)
            n = cnt[b, 0]
            assert n == len(k) and cnt[b, 1] == mono, (b, n, len(k))
            assert np.array_equal(kps[b, :n].view(np.uint8), k.view(np.uint8)) and np.array_equal(desc[b, :n], d), b
        a, b_ = C.c_int(-1), C.c_int(-1)
        assert lib.orbx_last_schedule(h, C.byref(a), C.byref(b_)) == 0
        return a.value, b_.value
    plans = [run(frames) for _ in range(3)]
    assert plans[0] == (1, 0) and plans[1][0] == 1 and plans[2] == (1, 1), plans
    assert run(np.ascontiguousarray(frames[::-1][:3])) [0] == 1          # another batch size and content on stale counts
    assert run(frames)[0] == 1                                           # (whether the heavy launch runs here depends on when the host read the word)
