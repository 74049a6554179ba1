#!/usr/bin/env python3
"""bench.py — hot-path throughput on MI355X.

One "step" = one pass of the hot path over one batch of synthetic frames already resident in HBM:
ORBextractor (pyramid, FAST, octree, orientation + blur + rBRIEF) -> Frame::UndistortKeyPoints -> AssignFeaturesToGrid ->
ORBmatcher::SearchByProjection (motion model, th = 15, TH_HIGH, rotation histogram) of every frame against its partner frame's points.
Workload at N=1: the configuration BASELINE.json's metric is quoted on — 752x480, nFeatures = 1000, 8 levels, scale 1.2, FAST 20/7
(EuRoC.yaml values).  Frames are independent units: with N>1 every rank owns its own batch (weak scaling, no data-path collective) and
`value` = frames all ranks processed / max-over-ranks time.

Prints ONE JSON line on rank 0 (see the driver contract): `value` = extract+match frames/s over exactly --steps steps, `roofline` = the
step's dominant kernel (HIP-event timed on the launch stream inside the C ABI), `metric_components` = the other parts of BASELINE's
composite metric (extract only; LocalBA linearisations / LM iterations per second), `cpu_baseline` = the oracle (reference algorithm
restated) doing the same per-frame work on this box's host cores (rank 0, N=1 only).  At N>1 any failure — including a failing RCCL
collective in the exchange legs — is fatal (non-zero exit); at N=1 an auxiliary leg that fails is reported in `extra` and never costs the
headline line.
"""
import argparse
import json
import os
import re
import sys
import traceback
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

W, H, NFEAT = 752, 480, 1000
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured-achievable)
try:
    BASELINE_METRIC = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")))["metric"]
except Exception:   # noqa: BLE001
    BASELINE_METRIC = "frames/sec ORB extract+match (752\u00d7480, 1000 kp) + LocalBA iters/sec; 1/2/4/8 GPU"


def level_sizes(W, H, n=8, sf=1.2):
    s, out = np.float32(1.0), []
    for i in range(n):
        inv = np.float32(1.0) / s
        out.append((int(np.rint(np.float32(W) * inv)), int(np.rint(np.float32(H) * inv))))
        s = np.float32(s * np.float64(np.float32(sf)))
    return out


def algorithmic_bytes(W, H, N):
    """per-frame algorithmic bytes of the whole extract, and of each kernel."""
    lv = level_sizes(W, H)
    P = [w * h for w, h in lv]
    p_ge1 = sum(P[1:])
    whole = P[0] + 2 * p_ge1 + N * (961 + 512 + 60)
    per_kernel = {
        "pyramid": P[0] + sum(P[1:-1]) + p_ge1,      # read levels 0..6 once, write levels 1..7 once
        "fast": sum(P) + 4 * 10 * N,                   # read every level once + ~10N packed candidates out
        "octree": 4 * 10 * N * 2 + 8 * N,              # candidates in (twice: assign + select), selection out
        "describe": N * (961 + 512 + 60),              # 31x31 orientation patch + 512 rBRIEF samples in, 28 B key point + 32 B descriptor out
    }
    return whole, per_kernel


def describe_patch_bytes(N):
    """What k_describe2 really stages per key point: the 43x43 neighbourhood the 7x7 blur of the 31x31 patch reads (+ the 60 bytes out).  Reported
    next to the SURVEY figure (`algorithmic_bytes()["describe"]`), never used for a roofline fraction."""
    return N * (43 * 43 + 60)


_C_TOKEN = re.compile(r'"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\'|//[^\n]*|/\*.*?\*/', re.S)


def source_sha():
    """sha256 over the kernel sources a PMC pass belongs to (profiles/pmc_latest.json carries the same figure).  Comments and white space
    are taken out first (string / character literals are kept as they are), so a comment-only edit of a kernel file does not orphan the
    counters that were measured on the same code."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "awesome-orb-slam3-3dvisioncraft-version_amd", "csrc")
    for f in ("orbx_extractor.hip", "orbm_matcher.hip", "orbf_frame.hip"):
        src = open(os.path.join(d, f), "r", encoding="utf-8", errors="replace").read()
        src = _C_TOKEN.sub(lambda m: " " if m.group(0)[0] == "/" else m.group(0), src)
        h.update(" ".join(src.split()).encode())
        h.update(b"\0")
    return h.hexdigest()[:16]


def match_algorithmic_bytes(N, Nq):
    """SURVEY.md §8(d) A_match per matched frame pair, and the split over the stage-2 kernels used for a per-kernel roofline."""
    whole = 2 * N * (32 + 28) + Nq * (32 + 24) + Nq * 8
    per_kernel = {
        "undistort_grid": N * 28 * 2 + (64 * 48 + 1) * 4 + N * 4,  # keypoint records in, undistorted records + CSR out
        "sbp_frame": Nq * (28 + 32) + N * (28 + 32) + (64 * 48 + 1) * 4 + N * 4 + N * 4 + Nq * 4,   # projection records + descriptors, the frame's records + descriptors + CSR once, mvpMapPoints + per-query result out
        "sbp_fallback": 8,                                          # one flag word per frame in, nothing out (unless the frame is flagged)
    }
    return whole, per_kernel


# timed stage -> the kernel that is the stage (round 4: UndistortKeyPoints + AssignFeaturesToGrid are one launch, the projection search is one workgroup
# per frame; "sbp_fallback" = the gated k_sbp_candidates_flagged + k_sbp_resolve launches behind it, a per-frame flag check unless a frame needs them)
KNAMES = {"pyramid": "k_resize2", "fast": "k_fast", "octree": "k_octree", "describe": "k_describe2", "undistort_grid": "k_undistort_grid",
          "sbp_frame": "k_sbp_frame", "sbp_fallback": "k_sbp_resolve"}
SHIFT = (6, -4)   # frame 2j+1 = frame 2j moved by (dx, dy) px + sensor noise: consecutive views of one scene
CAM_EUROC = (458.654, 457.296, 367.215, 248.375, (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05))   # EuRoC.yaml:9-20 (cam0, radtan)


def camera_for(w, h):
    """EuRoC's calibration with the intrinsics scaled to the frame size (it is quoted at 752x480; identical floats there)."""
    fx, fy, cx, cy, dist = CAM_EUROC
    return (fx * w / 752.0, fy * h / 480.0, cx * w / 752.0, cy * h / 480.0, dist)


def _synth_one(spec):
    """One base scene by kind (SURVEY 8(d) "Synthetic inputs"): textured = the corner-rich generator, sparse = 10 rectangles and 5 discs on the smooth
    background, low_contrast = the textured scene at 12 % contrast (forces the minThFAST retries), flat = constant grey (zero key points), noise =
    uniform random bytes (every cell full of corners)."""
    from orbhip.synth import flat_image, low_contrast_image, synth_image
    kind, seed, w, h = spec
    if kind == "textured":
        return synth_image(seed, w, h)
    if kind == "sparse":
        return synth_image(seed, w, h, n_rect=10, n_disc=5)
    if kind == "low_contrast":
        return low_contrast_image(seed, w, h)
    if kind == "flat":
        return flat_image(w, h, 40 + seed % 160)
    if kind == "noise":
        return np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)
    raise ValueError(kind)


def synth_many(specs, workers=1):
    """[(kind, seed, w, h)] -> list of images.  `workers` > 1 forks that many children (0.15 s per textured 752x480 scene on one core): only to be used
    BEFORE this process initialises HIP.  Plain os.fork + a file per child under /dev/shm + os._exit: no multiprocessing pool — a pool stops its workers
    with SIGTERM and lets them run their exit handlers, and under rocprofv3 (whose tool library every child inherits) a worker then hangs in the
    profiler's signal handler and takes the parent's join with it (round 5: a --pmc pass of this script sat there until the box's time limit)."""
    if workers <= 1 or len(specs) < 8:
        return [_synth_one(sp) for sp in specs]
    import shutil
    import tempfile
    n = min(workers, len(specs))
    need = sum(sp[2] * sp[3] for sp in specs) + (1 << 20)
    shm_ok = os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 2 * need      # (a container's /dev/shm can be 64 MB: then the default temp dir)
    tmp = tempfile.mkdtemp(prefix="orbhip_synth_", dir="/dev/shm" if shm_ok else None)
    pids = []
    try:
        for k in range(n):
            pid = os.fork()
            if pid == 0:
                code = 1
                try:
                    part = [_synth_one(sp) for sp in specs[k::n]]
                    np.save(os.path.join(tmp, "part%d.npy" % k), np.stack(part))
                    code = 0
                finally:
                    os._exit(code)          # no exit handlers, no inherited profiler finalisation
            pids.append(pid)
        failed = [pid for pid in pids if os.waitpid(pid, 0)[1] != 0]
        if failed:
            return [_synth_one(sp) for sp in specs]      # a child died (memory, a signal): do it here
        out = [None] * len(specs)
        for k in range(n):
            part = np.load(os.path.join(tmp, "part%d.npy" % k))
            for j, img in enumerate(part):
                out[k + j * n] = img
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _pair_up(base, B, rng):
    """Frames 2j, 2j+1 = base scene j mod len(base) (rolled when the bases are used more than once), and the same moved by SHIFT px + sensor noise:
    consecutive views of one scene."""
    frames = []
    for j in range((B + 1) // 2):
        k = j // len(base)
        a = np.roll(base[j % len(base)], (7 * k, 13 * k), (0, 1))
        b = np.clip(np.roll(a, (SHIFT[1], SHIFT[0]), (0, 1)).astype(np.int16) + rng.integers(-3, 4, a.shape), 0, 255).astype(np.uint8)
        frames += [a, b]
    return np.stack(frames[:B])


def make_batch(B, seed0=0, unique=16, w=None, h=None, workers=1):
    """B frames = B/2 pairs (a scene, the scene moved by SHIFT + noise) over `unique` distinct base scenes (seeds seed0 ..).  The headline uses
    unique = B/2: every pair its own seeded scene."""
    w, h = w or W, h or H
    base = synth_many([("textured", seed0 + i, w, h) for i in range(min(unique, max(1, B // 2)))], workers)
    return _pair_up(base, B, np.random.default_rng(seed0 + 12345))


MIXED_SHARES = (("textured", 0.60), ("sparse", 0.20), ("low_contrast", 0.10), ("flat", 0.05), ("noise", 0.05))


def make_mixed_batch(B, seed0=0, w=None, h=None, workers=1):
    """The mixed batch of the round-4 verdict: B/2 pairs whose base scenes are 60 % textured, 20 % sparse, 10 % low-contrast, 5 % flat, 5 % uniform
    noise — every scene its own seed, the pairs shuffled — so that neighbouring workgroups see different amounts of work.  -> (frames, kinds per frame)"""
    w, h = w or W, h or H
    npairs = (B + 1) // 2
    kinds = []
    for kind, share in MIXED_SHARES:
        kinds += [kind] * int(round(share * npairs))
    kinds = (kinds + ["textured"] * npairs)[:npairs]
    rng = np.random.default_rng(seed0 + 777)
    order = rng.permutation(npairs)
    kinds = [kinds[i] for i in order]
    base = synth_many([(k, seed0 + i, w, h) for i, k in enumerate(kinds)], workers)
    frames = _pair_up(base, B, np.random.default_rng(seed0 + 12345))
    return frames, [kinds[i // 2] for i in range(B)]


def grow_batch_on_device(d_frames, B):
    """A batch of B frames from a smaller resident one: copy i = frame i mod B0 rolled by (5, 9) px x (i div B0) — different images
    (key points move with the content, cells and borders do not), built on the device so that a 4096-frame batch costs no host time."""
    import torch
    B0 = d_frames.shape[0]
    if B <= B0:
        return d_frames[:B].contiguous()
    parts = [d_frames] + [torch.roll(d_frames, shifts=(5 * k, 9 * k), dims=(1, 2)) for k in range(1, -(-B // B0))]
    return torch.cat(parts)[:B].contiguous()


def build_match_queries(kps, counts, scale, cap):
    """Motion-model queries (ORBmatcher.cc:2265-2331, mono: neither forward nor backward, th=15): every keypoint of the partner
    frame becomes a projected map point at its position moved by the known inter-frame shift."""
    from orbhip.matcher import QUERY_DTYPE, Q_VALID, Q_HAS_OBS
    B = kps.shape[0]
    q = np.zeros((B, cap), QUERY_DTYPE)
    src = np.arange(B) ^ 1
    src[src >= B] = B - 1
    sgn = np.where(np.arange(B) % 2 == 1, 1.0, -1.0).astype(np.float32)   # odd frames see even frames' points moved by +SHIFT
    for b in range(B):
        n = counts[src[b], 0]
        k = kps[src[b], :n]
        lvl = k[:, 5].view(np.int32)
        q["u"][b, :n] = k[:, 0] + sgn[b] * np.float32(SHIFT[0]); q["v"][b, :n] = k[:, 1] + sgn[b] * np.float32(SHIFT[1])
        q["radius"][b, :n] = np.float32(15.0) * scale[lvl]
        q["min_level"][b, :n] = lvl - 1; q["max_level"][b, :n] = lvl + 1
        q["angle"][b, :n] = k[:, 3]
        q["flags"][b, :n] = Q_VALID | Q_HAS_OBS
    nq = counts[src, 0].astype(np.int32).copy()
    return q, nq, src


class StepPipeline:
    """The benchmark step on one GPU: ORBextractor -> Frame::UndistortKeyPoints (EuRoC calibration) -> AssignFeaturesToGrid ->
    ORBmatcher::SearchByProjection (motion model, th = 15, TH_HIGH, rotation histogram) of every frame against its partner frame's points.
    With `streams` >= 2 the match kernels of step i run on a second HIP stream next to the extraction of step i+1 (double-buffered extractor
    outputs, an event pair per buffer set orders producer and consumer); with >= 3, `streams - 1` extractor handles alternate on their own
    streams.  The same object is what tests/test_bench_config_parity.py checks against the oracle frame by frame."""
    LAP = (0, 1000)

    def __init__(self, d_frames, w, h, nfeat, device_index, streams=3, frames_host=None):
        """d_frames: one resident batch [B,H,W] u8, or a LIST of them = input sets the steps rotate over (step i reads set i mod len: consecutive
        steps never see the same images; every set has its own prepared projection records).  frames_host: the same on the host (for the checker)."""
        import torch
        import orbhip
        from orbhip.frame import Camera, FrameOps
        self.torch, self.orbhip = torch, orbhip
        sets = list(d_frames) if isinstance(d_frames, (list, tuple)) else [d_frames]
        hosts = (list(frames_host) if isinstance(frames_host, (list, tuple)) else [frames_host]) if frames_host is not None else [None] * len(sets)
        self.w, self.h, self.nfeat, self.di, self.streams = w, h, nfeat, device_index, streams
        self.dev = sets[0].device
        self.B = sets[0].shape[0]
        assert all(s_.shape == sets[0].shape for s_ in sets) and len(hosts) == len(sets)
        self.cam = camera_for(w, h)
        self.ex = orbhip.ORBextractor(nfeat, 1.2, 8, 20, 7, device=device_index, max_batch=self.B)
        self.m = orbhip.ORBmatcher(0.9, True)
        self.fo = FrameOps(Camera.make(*self.cam), w, h)
        self.grid = self.fo.grid
        # first pass per input set: the projection records of every frame's partner (the "last frame" of the motion model) — prepared once, resident in
        # HBM, like a map
        self.sets = []
        self.out = None
        for d_set, host in zip(sets, hosts):
            self.out = self.ex.extract_batch(d_set, self.LAP, out=self.out)
            self.cap = self.out[0].shape[1]
            un = self.fo.UndistortKeyPoints(self.out[0], self.out[2].view(-1), count_stride=2)
            torch.cuda.synchronize()
            counts0 = self.out[2].cpu().numpy()
            q, nq, src = build_match_queries(un.cpu().numpy(), counts0, self.ex.GetScaleFactors(), self.cap)
            self.sets.append(dict(d_frames=d_set, frames_host=host, q=q, nq=nq, src=src,
                                  d_q=torch.from_numpy(q.view(np.uint8).reshape(self.B, self.cap, 28)).to(self.dev), d_nq=torch.from_numpy(nq).to(self.dev),
                                  d_qdesc=self.out[1][torch.from_numpy(src).to(self.dev)].contiguous()))
            self.un = un
        self._use_set(0)
        self.work = torch.empty(self.m._L.orbm_search_workspace_bytes(self.B, self.cap), dtype=torch.uint8, device=self.dev)
        self.res = self.gbuf = None
        self._stream_state = None

    def _use_set(self, i):
        """Make input set i the current one: what the next extraction / match reads and what snapshot() / check_against_oracle() refer to."""
        self.cur = i
        st = self.sets[i]
        self.d_frames, self.frames_host, self.q, self.nq, self.src = st["d_frames"], st["frames_host"], st["q"], st["nq"], st["src"]
        self.d_q, self.d_nq, self.d_qdesc = st["d_q"], st["d_nq"], st["d_qdesc"]

    def _match(self, out):
        cnt = out[2].view(-1)
        self.un, gs, gi = self.fo.UndistortAndGrid(out[0], cnt, count_stride=2, out=None if self.gbuf is None else (self.un, self.gbuf[0], self.gbuf[1]))
        self.gbuf = (gs, gi)
        self.res = self.m.SearchByProjection(self.un, out[1], cnt, self.gbuf[0], self.gbuf[1], self.d_q, self.d_qdesc, self.d_nq, self.grid, 1, 100,
                                             count_stride=2, work=self.work, out=self.res)

    def sequential_step(self):
        self.out = self.ex.extract_batch(self.d_frames, self.LAP, out=self.out)
        self._match(self.out)

    def capture_graph(self):
        """The sequential step (extraction + match on ONE stream) captured once as a HIP graph and replayed: what a caller with few frames per call does
        about launch overhead (a step is ~20 launches; at B <= 16 they cost more than the kernels).  The captured call keeps its FAST pass form
        (include/orbhip.h, orbx_last_fast_passes).  -> False when the capture is refused."""
        torch = self.torch
        assert len(self.sets) == 1
        self.sequential_step()                       # every buffer of the step exists before the capture
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    self.sequential_step()
            torch.cuda.synchronize()
            self._graph = g
            return True
        except Exception as err:   # noqa: BLE001
            self._graph, self._graph_error = None, "%s: %s" % (type(err).__name__, err)
            torch.cuda.synchronize()
            return False

    def graph_step(self):
        self._graph.replay()

    def kernel_times(self, warm=4):
        """Per-kernel device times (HIP events recorded on the launch stream inside the C ABI) from one untimed, sequential step — call before
        the extra streams exist."""
        torch = self.torch
        for _ in range(warm):
            self.sequential_step()
        torch.cuda.synchronize()
        kern = {}
        self.m.enable_timing(True)
        self.ex.enable_timing(True)
        self.out = self.ex.extract_batch(self.d_frames, self.LAP, out=self.out)
        self._match(self.out)
        torch.cuda.synchronize()
        for k, v in self.ex.last_timing().items():
            kern[k if k != "total" else "extract_total"] = v
        self.ex.enable_timing(False)
        mt = self.m.last_timing()      # (grid_build, sbp_candidates, sbp_resolve) = the three event intervals of the matcher's launches
        kern["undistort_grid"], kern["sbp_frame"], kern["sbp_fallback"] = mt["grid_build"], mt["sbp_candidates"], mt["sbp_resolve"]
        self.m.enable_timing(False)
        return kern

    def start_streams(self):
        torch, orbhip = self.torch, self.orbhip
        st = self.streams
        sA = torch.cuda.current_stream(self.dev)
        sB = torch.cuda.Stream(self.dev) if st >= 2 else sA
        nbuf = 1 if st == 1 else max(2, st - 1)             # buffer sets in rotation (>= 3 streams: one extractor handle + stream each)
        sX = [sA] + [torch.cuda.Stream(self.dev) if st >= 3 else sA for _ in range(nbuf - 1)]          # extraction stream of buffer set k
        exs = [self.ex] + [orbhip.ORBextractor(self.nfeat, 1.2, 8, 20, 7, device=self.di, max_batch=self.B) if st >= 3 else self.ex for _ in range(nbuf - 1)]
        self._stream_state = dict(sB=sB, nbuf=nbuf, sX=sX, exs=exs, outs=[self.out] + [None] * (nbuf - 1), evA=[torch.cuda.Event() for _ in range(nbuf)],
                                  evB=[torch.cuda.Event() for _ in range(nbuf)], evB_set=[False] * nbuf, step_no=0)

    def step(self):
        torch = self.torch
        S = self._stream_state
        k = S["step_no"] % S["nbuf"]
        if len(self.sets) > 1:
            self._use_set(S["step_no"] % len(self.sets))
        S["step_no"] += 1
        with torch.cuda.stream(S["sX"][k]):
            if self.streams >= 2 and S["evB_set"][k]:
                S["sX"][k].wait_event(S["evB"][k])                   # the match that read this buffer set two steps ago has finished
            S["outs"][k] = S["exs"][k].extract_batch(self.d_frames, self.LAP, out=S["outs"][k])
            self.out = S["outs"][k]
            if self.streams >= 2:
                S["evA"][k].record(S["sX"][k])
        with torch.cuda.stream(S["sB"]):
            if self.streams >= 2:
                S["sB"].wait_event(S["evA"][k])
            self._match(self.out)
            if self.streams >= 2:
                S["evB"][k].record(S["sB"])
                S["evB_set"][k] = True

    # ---- host-fed form of the same step (Tracking::GrabImageMonocular receives host images, src/Tracking.cc:507-560): the input sets live in pinned
    # host memory, the resident device sets become staging buffers, every result of a step goes back to pinned host memory
    def start_host_fed(self):
        torch = self.torch
        S = self._stream_state
        assert S is not None and self.streams >= 3 and all(s_["frames_host"] is not None for s_ in self.sets)
        ns = len(self.sets)
        # The copy streams are HIGH-PRIORITY streams: HIP multiplexes its streams onto a few hardware queues (4 per priority by default) and hands them
        # out round robin — as plain streams the H2D stream landed on the queue of an extraction stream and every second copy waited for a whole
        # extraction (4.1 instead of 3.4 ms per step, profiles/r06_host_fed_queues.txt); raising GPU_MAX_HW_QUEUES instead un-shares the two extraction
        # streams as well, which costs the resident step 12 % (1.51 -> 1.72 ms).  Queues of another priority are never shared with the kernels' streams.
        prio = int(os.environ.get("ORBHIP_COPY_STREAM_PRIORITY", "-1"))
        self._hf = dict(sH=torch.cuda.Stream(self.dev, priority=prio), sD=torch.cuda.Stream(self.dev, priority=prio), pin=[torch.from_numpy(s_["frames_host"]).pin_memory() for s_ in self.sets],
                        evH=[torch.cuda.Event() for _ in range(ns)], evX=[torch.cuda.Event() for _ in range(ns)], evX_set=[False] * ns,
                        evD=[torch.cuda.Event() for _ in range(S["nbuf"])], evD_set=[False] * S["nbuf"], host_out=[None] * S["nbuf"], last=None, prev_k=None)
        torch.cuda.synchronize()

    def host_fed_step(self):
        """H2D of this step's B images (copy stream) -> extraction (its handle's stream) -> undistort + grid + SearchByProjection (match stream) -> D2H of
        every result (second copy stream); events order the four, nothing waits on the host: the H2D of step i+1 runs under the kernels of step i."""
        torch = self.torch
        S, F = self._stream_state, self._hf
        j, k = S["step_no"] % len(self.sets), S["step_no"] % S["nbuf"]
        self._use_set(j)
        S["step_no"] += 1
        with torch.cuda.stream(F["sH"]):
            if F["evX_set"][j]:
                F["sH"].wait_event(F["evX"][j])                      # the extraction that last read this staging buffer has finished
            self.d_frames.copy_(F["pin"][j], non_blocking=True)
            F["evH"][j].record(F["sH"])
        with torch.cuda.stream(S["sX"][k]):
            S["sX"][k].wait_event(F["evH"][j])
            if S["evB_set"][k]:
                S["sX"][k].wait_event(S["evB"][k])
            if F["evD_set"][k]:
                S["sX"][k].wait_event(F["evD"][k])                   # this buffer set's previous results are on the host
            if not F.get("skip_kernels"):                            # (experiment knob, tools/exp.py host_fed)
                S["outs"][k] = S["exs"][k].extract_batch(self.d_frames, self.LAP, out=S["outs"][k])
            self.out = S["outs"][k]
            S["evA"][k].record(S["sX"][k])
            F["evX"][j].record(S["sX"][k]); F["evX_set"][j] = True
        with torch.cuda.stream(S["sB"]):
            S["sB"].wait_event(S["evA"][k])
            if F["prev_k"] is not None:
                S["sB"].wait_event(F["evD"][F["prev_k"]])            # (the match outputs are single-buffered: the previous step's are on the host)
            if not F.get("skip_kernels"):
                self._match(self.out)
            S["evB"][k].record(S["sB"]); S["evB_set"][k] = True
        dev_res = (self.out[0], self.out[1], self.out[2], self.un, self.res[0], self.res[1], self.res[2])
        if F["host_out"][k] is None:
            F["host_out"][k] = [torch.empty(t_.shape, dtype=t_.dtype).pin_memory() for t_ in dev_res]
        with torch.cuda.stream(F["sD"]):
            F["sD"].wait_event(S["evB"][k])
            for h_, d_ in zip(F["host_out"][k], dev_res if not F.get("skip_d2h") else ()):
                h_.copy_(d_, non_blocking=True)
            F["evD"][k].record(F["sD"]); F["evD_set"][k] = True
        F["last"], F["prev_k"] = (k, j), k

    def host_fed_bytes(self):
        """(H2D bytes, D2H bytes) of one host-fed step."""
        F = self._hf
        ho = next(h_ for h_ in F["host_out"] if h_ is not None)
        return int(F["pin"][0].numel()), int(sum(t_.numel() * t_.element_size() for t_ in ho))

    def host_fed_copy_only(self, steps, direction):
        """The step's copies alone (no kernels): `steps` x H2D of an input set, or D2H of a result set — the PCIe side of the overlap figure."""
        torch = self.torch
        F = self._hf
        ho = next(h_ for h_ in F["host_out"] if h_ is not None)
        dev_res = (self.out[0], self.out[1], self.out[2], self.un, self.res[0], self.res[1], self.res[2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            if direction == "h2d":
                with torch.cuda.stream(F["sH"]):
                    self.sets[i % len(self.sets)]["d_frames"].copy_(F["pin"][i % len(self.sets)], non_blocking=True)
            else:
                with torch.cuda.stream(F["sD"]):
                    for h_, d_ in zip(ho, dev_res):
                        h_.copy_(d_, non_blocking=True)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def check_host_fed_against_oracle(self, sel=None, nthreads=None):
        """The last completed host-fed step, read from the PINNED HOST result buffers, against the oracle."""
        import bench_check
        self.torch.cuda.synchronize()
        F = self._hf
        k, j = F["last"]
        sel = np.arange(self.B) if sel is None else np.asarray(sel)
        ho = [t_.numpy()[sel] for t_ in F["host_out"][k]]
        snap = dict(kps=ho[0], desc=ho[1], counts=ho[2], un=ho[3], q_match=ho[4], kp_match=ho[5], nm=ho[6])
        st = self.sets[j]
        cam9 = np.array(list(self.cam[:4]) + list(self.cam[4]) + [0.0], np.float32)
        qd = st["d_qdesc"][self.torch.as_tensor(sel, device=self.dev, dtype=self.torch.long)].cpu().numpy()
        return bench_check.compare_step(st["frames_host"][sel], snap, st["q"][sel].copy(), qd, st["nq"][sel].copy(), cam9, np.array(self.grid, np.float32), self.nfeat,
                                        lap=self.LAP, nthreads=nthreads)

    def extract_only_step(self):
        if len(self.sets) > 1:
            self._use_set((self.cur + 1) % len(self.sets))
        self.out = self.ex.extract_batch(self.d_frames, self.LAP, out=self.out)

    def snapshot(self, sel=None):
        """Host copies of the last completed step's outputs (after a device synchronize) for the frames `sel` (default: all)."""
        torch = self.torch
        torch.cuda.synchronize()
        idx = slice(None) if sel is None else torch.as_tensor(np.asarray(sel), device=self.dev, dtype=torch.long)
        g = lambda t: t[idx].cpu().numpy()
        return dict(kps=g(self.out[0]), desc=g(self.out[1]), counts=g(self.out[2]), un=g(self.un), q_match=g(self.res[0]), kp_match=g(self.res[1]),
                    nm=g(self.res[2]))

    def check_against_oracle(self, sel=None, frames_host=None, nthreads=None):
        """The last completed step against the oracle on the frames `sel` (default all): -> (frames compared, mismatch descriptions, totals).
        Test / measurement-hygiene infrastructure: called after timed regions only."""
        import bench_check
        sel = np.arange(self.B) if sel is None else np.asarray(sel)
        snap = self.snapshot(sel)
        if frames_host is None:
            frames_host = self.frames_host
        fr = frames_host[sel] if frames_host is not None else self.d_frames[self.torch.as_tensor(sel, device=self.dev, dtype=self.torch.long)].cpu().numpy()
        cam9 = np.array(list(self.cam[:4]) + list(self.cam[4]) + [0.0], np.float32)
        qd = self.d_qdesc[self.torch.as_tensor(sel, device=self.dev, dtype=self.torch.long)].cpu().numpy()
        return bench_check.compare_step(fr, snap, self.q[sel].copy(), qd, self.nq[sel].copy(), cam9, np.array(self.grid, np.float32), self.nfeat,
                                        lap=self.LAP, nthreads=nthreads)


def measure_traffic_live(kernel_base, batch, size, nfeat, timeout_s=150):
    """HBM bytes per launch of `kernel_base` (template instances summed: a batch runs each once) measured NOW, for the build and box this line comes from:
    two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md) over a short single-stream run of
    this script's own step, nothing else traced.  -> (bytes corrected, bytes raw, note) or (None, None, reason).
    Correction as the guide prescribes for gfx950: FETCH_SIZE x 2 for wide coalesced reads (so raw <= true <= corrected), WRITE_SIZE as reported."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="orbhip_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline", "--headline-only", "--no-parity-check",
           "--streams", "1", "--no-pmc", "--input-sets", "1", "--batch", str(batch), "--size", size, "--nfeatures", str(nfeat)]
    got = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            r = subprocess.run([exe, "--pmc", ctr, "-d", out, "-o", "pmc", "--"] + cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                               timeout=timeout_s)
            dbs = glob.glob(os.path.join(out, "**", "*results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, None, "rocprofv3 --pmc %s failed (rc %d)" % (ctr, r.returncode)
            tot = 0.0
            con = sqlite3.connect(dbs[0])
            for name, mean in con.execute("select kernel_name, avg(value) from counters_collection where counter_name = ? group by kernel_name", (ctr,)):
                b = re.sub(r"^void\s+", "", name).split("(")[0]
                b = re.sub(r"<.*>$", "", b)
                if b == kernel_base:
                    tot += mean * 1024.0           # the counters are reported in KB per dispatch
            con.close()
            got[ctr] = tot
        if got["FETCH_SIZE"] <= 0:
            return None, None, "no %s dispatches in the PMC pass" % kernel_base
        return 2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"], got["FETCH_SIZE"] + got["WRITE_SIZE"], "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of a 3-step single-stream run of this step), FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE"
    except Exception as err:   # noqa: BLE001
        return None, None, "live PMC pass failed: %s: %s" % (type(err).__name__, err)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def usable_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU boxes show 256 hardware threads but
    run the container under a 16-CPU quota: threads beyond the quota only add throttling).  -> (cores, quota or None)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            quota = float(a) / float(b)
    except Exception:   # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p_
        except Exception:   # noqa: BLE001
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(frames, q, qdesc, nq, cam9, grid4, budget_s=24.0):
    """Oracle (reference algorithm restated, g++ -O3) on this host in native threads (oracle/bench_oracle.cpp): one extractor per thread,
    one frame per thread at a time (the reference extracts one image on one thread, Frame.cc:111-114), each frame = ORBextractor +
    UndistortKeyPoints + AssignFeaturesToGrid + motion-model SearchByProjection with the same prepared projection records as the GPU step."""
    import oracle_lib as O
    cores, quota = usable_cores()
    run = lambda nt, per, match: O.bench_extract_match_mt(frames, nt, per, cam9, grid4, q, qdesc, nq, 100, 0.9, True, do_match=match, nfeatures=NFEAT)
    s1, _, _ = run(1, 6, True)          # warm + calibrate
    fps1 = 6 / s1
    n1 = max(8, int(fps1 * budget_s / 6))
    s1, _, _ = run(1, n1, True)
    fps1 = n1 / s1
    s1x, _, _ = run(1, n1, False)
    fps1x = n1 / s1x
    # thread-count sweep: one frame per thread at a time; the best aggregate is the baseline, the per-core figure is reported next to it
    sweep, best = [], None
    for nt in sorted({max(1, cores // 2), cores, min(2 * cores, os.cpu_count() or cores)}):
        per_thread = max(4, int(fps1 * budget_s / 6))
        sN, kp, mt = run(nt, per_thread, True)
        fpsN = nt * per_thread / sN
        sweep.append({"threads": nt, "frames_per_s": round(fpsN, 1), "per_thread": round(fpsN / nt, 2)})
        if best is None or fpsN > best[0]:
            best = (fpsN, nt, per_thread, kp / (nt * per_thread), mt / (nt * per_thread))
    fpsN, nt, per_thread, meankp, meanmt = best
    return {"value": round(fpsN, 2), "unit": "frames/s", "cores": nt, "kind": "port",
            "per_core": round(fpsN / nt, 2), "host_threads": os.cpu_count(), "cpu_quota": quota, "single_thread": round(fps1, 2), "single_thread_extract_only": round(fps1x, 2), "sweep": sweep,
            "sample": "extract + UndistortKeyPoints + grid + SearchByProjection per frame; best of a thread-count sweep around the %d CPUs this container may use: %d native "
                      "threads x %d frames (round-robin over %d frames of the same synthetic %dx%d batch), oracle = reference algorithm restated, "
                      "g++ -O3, per-thread malloc arenas; single thread: %.2f frames/s over %d frames; mean %.1f keypoints, %.1f matches per frame"
                      % (cores, nt, per_thread, len(frames), W, H, fps1, n1, meankp, meanmt)}


def resolve_world(gpus, environ, visible_devices=None):
    """What `--gpus N` means for this process.  -> ("run", world) when this process is a rank (or the single-GPU run), ("launch", N) when it must
    start the N ranks itself; raises SystemExit(2) with a message when --gpus contradicts the launcher's WORLD_SIZE or asks for more GPUs than the
    node shows.  The driver starts N>1 as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (WORLD_SIZE = N: "run"); a bare
    `python bench.py --gpus N` (WORLD_SIZE unset) launches the same thing itself — it never silently measures one GPU and prints n_gpus = 1."""
    env_world = environ.get("WORLD_SIZE")
    if env_world is not None:
        world = int(env_world)
        if gpus is not None and gpus != world:
            raise SystemExit("bench.py: --gpus %d contradicts WORLD_SIZE=%d set by the launcher; refusing to run" % (gpus, world))
        return "run", world
    n = 1 if gpus is None else gpus
    if n < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if n == 1:
        return "run", 1
    if visible_devices is not None and environ.get("ORBHIP_BENCH_ONE_DEVICE") != "1" and visible_devices < n:
        raise SystemExit("bench.py: --gpus %d but this node shows %d GPU(s); refusing to run (ORBHIP_BENCH_ONE_DEVICE=1 puts every rank on device 0: "
                         "a dry run of the N>1 code path, not a measurement)" % (n, visible_devices))
    return "launch", n


def self_launch(n, argv):
    """Replace this process by `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port P bench.py <argv>`
    (one rank per GPU over RCCL; the same command line the driver uses)."""
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("MASTER_PORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush(); sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs of this node = ranks (default: WORLD_SIZE if a launcher set it, else 1).  Without a launcher, "
                                                            "N > 1 starts the N ranks itself (torch.distributed.run, 127.0.0.1); a --gpus that contradicts WORLD_SIZE is refused")
    ap.add_argument("--launch-check", action="store_true", help="resolve --gpus / WORLD_SIZE, start the ranks, form the process group, print "
                                                                 '{"n_gpus": world, "ranks": [...]} on rank 0 and exit — no GPU touched (CPU-tier test of the launch logic)')
    ap.add_argument("--steps", type=int, default=50)   # 50 x 2.4 ms: the un-overlapped match of the last step (0.5 ms) is 0.4 % of the timed region (1 % at 20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=3, choices=(1, 2, 3, 4, 5),
                    help="2: the match kernels of step i run on a second HIP stream next to the extraction of step i+1 (double-buffered extractor outputs); "
                         "3: additionally two extractor handles alternate on two streams (the extractions of consecutive steps overlap); 4 / 5: three / four "
                         "handles in rotation (measured on MI355X: 2.33-2.36 ms per step against 2.35 with 3 — nothing left to overlap); 1: everything in one stream")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of exactly --steps steps each; value = the median region")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with two rocprofv3 --pmc passes of a short sub-run (N = 1 only; ~20 s)")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the oracle comparison of the last timed step (profiling runs)")
    ap.add_argument("--all-legs", action="store_true", help="N > 1: run every auxiliary leg on every rank (default: the LBA and 1280x720 legs + the exchange leg)")
    ap.add_argument("--exchange", default="rccl", choices=("rccl", "peer", "both"),
                    help="N > 1: how the exchange leg all-gathers the frame blocks — RCCL (one all_gather_into_tensor), explicit peer copies over IPC "
                         "handles (orbd_allgather_frames_peer: one pull per peer and slab, a stream per peer), or both; the peer form has never run "
                         "on more than one GPU (DESIGN section 5), so the driver's default stays RCCL and a failing peer leg is reported, not fatal")
    ap.add_argument("--headline-only", action="store_true", help="skip the extract+match and LBA legs")
    ap.add_argument("--lba-windows", type=int, default=256, help="LBA windows per GPU per step (linearisations/s against windows per launch on MI355X: 4 -> 67 k, 16 -> 78 k, 64 -> 83 k, 256 -> 91 k)")
    ap.add_argument("--lm-windows", type=int, default=256, help="LBA windows per GPU per step in the full-LM leg")
    ap.add_argument("--input-sets", type=int, default=3, help="resident input sets the steps rotate over (step i reads set i mod N; every set = --batch frames "
                                                               "from --batch/2 distinct seeded scenes with its own projection records): consecutive steps never see the same images")
    ap.add_argument("--scenes", type=int, default=0, help="distinct base scenes per input set (0 = --batch/2: every frame pair its own scene; round 4 used 16)")
    ap.add_argument("--size", default="752x480", help="frame size WxH (the headline is 752x480; 1280x720 is BASELINE configs[3]'s frame shape)")
    ap.add_argument("--nfeatures", type=int, default=1000, help="ORBextractor nFeatures (1500 with --size 1280x720)")
    args = ap.parse_args()
    global W, H, NFEAT
    W, H = [int(v) for v in args.size.lower().split("x")]
    NFEAT = args.nfeatures

    ndev = None
    if "WORLD_SIZE" not in os.environ and (args.gpus or 1) > 1 and not args.launch_check:
        import torch
        ndev = torch.cuda.device_count()
    action, world = resolve_world(args.gpus, os.environ, ndev)
    if action == "launch":
        self_launch(world, sys.argv[1:])        # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        import torch.distributed as dist
        ranks = [rank]
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(os.environ.get("ORBHIP_BENCH_BACKEND", "gloo"))
            got = [None] * world
            dist.all_gather_object(got, (rank, local_rank))
            ranks = got
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"n_gpus": world, "ranks": ranks, "launch_check": True}))
        return

    # ---- host inputs first (a forked worker pool: before this process initialises HIP).  Headline: --input-sets resident sets of B frames, every frame
    # pair its own seeded scene (B/2 distinct scenes per set; the ranks use disjoint seeds)
    B = args.batch
    t_in = time.perf_counter()
    workers = max(1, min(16, usable_cores()[0] // max(1, world)))
    scenes = args.scenes or max(1, B // 2)
    host_sets = [make_batch(B, seed0=1000 * rank + 100000 * s_, unique=scenes, workers=workers) for s_ in range(max(1, args.input_sets))]
    frames = host_sets[0]
    aux_legs = not args.headline_only and (world == 1 or args.all_legs)
    host_mixed = make_mixed_batch(B, seed0=7000 + 1000 * rank, workers=workers) if aux_legs else None
    host_16 = make_batch(B, seed0=1000 * rank, unique=16, workers=workers) if aux_legs else None
    host_1280 = make_batch(256, seed0=5000 + 1000 * rank, unique=32, w=1280, h=720, workers=workers) if not args.headline_only else None
    input_seconds = time.perf_counter() - t_in

    import torch
    import torch.distributed as dist
    import orbhip

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # ORBHIP_BENCH_BACKEND=gloo + ORBHIP_BENCH_ONE_DEVICE=1: run the N>1 code path with all ranks on ONE GPU (how the multi-rank legs
        # were exercised on the 1-GPU development box); the driver's runs use the defaults = RCCL, one GPU per rank
        backend = os.environ.get("ORBHIP_BENCH_BACKEND", "nccl")
        if os.environ.get("ORBHIP_BENCH_ONE_DEVICE") == "1":
            local_rank = 0
        elif local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d has no GPU (local rank %d, %d visible); refusing to run" % (rank, local_rank, torch.cuda.device_count()))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    d_sets = [torch.from_numpy(f_).to(dev) for f_ in host_sets]
    d_frames = d_sets[0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    extra = {}

    def guard(name, fn):
        """Auxiliary legs: a failure is reported in the line (`extra.<leg>_error`) and never costs the headline.  At N>1 a leg that fails at the same
        point on every rank (a bug, a size limit) leaves the ranks in lock-step: they have passed the same barriers; a failure on ONE rank leaves the
        others in a barrier or collective, and RCCL's watchdog ends the run — with or without this guard."""
        try:
            return fn()
        except Exception as err:   # noqa: BLE001
            extra[name + "_error"] = "%s: %s" % (type(err).__name__, err)
            sys.stderr.write(traceback.format_exc())
            return None

    def timed(fn, steps, repeats):
        """`repeats` timed regions of exactly `steps` steps each, every one bracketed by barrier + synchronize on both sides -> list of seconds
        (max over ranks is taken by the caller)."""
        dts = []
        for _ in range(repeats):
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            barrier()
            dts.append(time.perf_counter() - t0)
        return dts

    def rank_max(vals):
        if world == 1:
            return list(vals)
        t = torch.tensor(list(vals), dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    # ---- the step: ORBextractor -> UndistortKeyPoints (EuRoC calibration) -> AssignFeaturesToGrid -> SearchByProjection (motion model)
    P = StepPipeline(d_sets, W, H, NFEAT, local_rank, streams=args.streams, frames_host=host_sets)
    ex, m, grid, cap = P.ex, P.m, P.grid, P.cap
    q, nq, d_qdesc = P.sets[0]["q"], P.sets[0]["nq"], P.sets[0]["d_qdesc"]          # set 0's projection records (the CPU baseline's sample)
    CAM = P.cam
    kern = P.kernel_times()          # a few sequential passes (warm kernels and clocks), then the per-kernel timing pass — before the extra streams exist
    P.start_streams()
    for _ in range(args.warmup):
        P.step()
    # the headline: `--repeats` timed regions of exactly --steps steps; `value` is the MEDIAN region (min / max reported next to it)
    dts_rank = timed(P.step, args.steps, args.repeats)
    dts = rank_max(dts_rank)
    dt = float(np.median(dts))
    per_rank_fps = [B * args.steps / float(np.median(dts_rank))]
    if world > 1:       # every rank's own median region (the barriers make them nearly equal; a slow GPU shows in the spread of the step's kernels instead)
        tg = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(tg, torch.tensor(per_rank_fps, dtype=torch.float64, device=dev))
        per_rank_fps = [float(t_.item()) for t_ in tg]
    torch.cuda.synchronize()
    out, res = P.out, P.res
    counts = out[2].cpu().numpy()
    nm = res[2].cpu().numpy()
    extra["step"] = {"mean_matches_per_frame": float(nm.mean()), "queries_per_frame": float(nq.mean()), "mean_keypoints": float(counts[:, 0].mean()),
                     "match_only_ms": round(kern["undistort_grid"] + kern["sbp_frame"] + kern["sbp_fallback"], 4)}
    # ---- parity of THIS run: the last timed step's outputs (all three streams, the handle whose turn it was) against the oracle, after the
    # timed region: every frame at N=1, 64 per rank at N>1.  A mismatch makes the line invalid: reported and the exit code is 3.
    parity = {"checked_frames": 0, "mismatches": None}
    if not args.no_parity_check:
        sel = np.arange(B) if world == 1 else np.unique(np.linspace(0, B - 1, min(B, 64)).astype(np.int64))
        tpc = time.perf_counter()
        n_chk, bad, tot = P.check_against_oracle(sel)           # (the input set the last timed step read)
        nbad = rank_max([float(len(bad))])[0]
        parity = {"checked_frames": int(n_chk) * world, "mismatches": int(nbad), "first_mismatches": bad[:4], "seconds": round(time.perf_counter() - tpc, 2),
                  "keypoints_compared": int(tot["keypoints"]), "matches_compared": int(tot["matches"]),
                  "what": "last timed step (streams=%d) vs oracle per frame: {N, monoIndex}, key point records, descriptors, undistorted records, "
                          "per-query match, mvpMapPoints, nmatches — bitwise" % args.streams}

    # ---- metric component: ORBextractor alone (BASELINE configs[1]), same batch, exactly --steps steps
    for _ in range(min(args.warmup, 2)):
        P.extract_only_step()
    dts_extract = rank_max(timed(P.extract_only_step, args.steps, max(1, min(args.repeats, 3))))
    dt_extract = float(np.median(dts_extract))
    out = P.out

    def leg_host_api():
        # the single-image host-buffer entry point (ORBextractor::operator() drop-in): PCIe + sync inclusive
        ex1 = orbhip.ORBextractor(NFEAT, 1.2, 8, 20, 7, device=local_rank)
        ex1(frames[0], None, (0, 1000))
        th = time.perf_counter()
        nh = 100
        for i in range(nh):
            ex1(frames[i % B], None, (0, 1000))
        extra["host_api"] = {"frames_per_s": round(nh / (time.perf_counter() - th), 1),
                             "what": "orbx_extract: one %dx%d host image per call, H2D + 4 kernels + D2H, synchronous (never `value`)" % (W, H)}

    def leg_host_fed():
        # the same step fed from host memory: B frames per step pinned on the host -> H2D on a copy stream under the previous step's kernels -> the
        # step -> every result back to pinned host memory on a second copy stream.  PCIe-bound by construction (B x W x H bytes in per step)
        P.start_host_fed()
        for _ in range(max(40, args.warmup)):      # (the first ~30 steps run at half speed: pinned result buffers are faulted in, the copy queues warm up)
            P.host_fed_step()
        dth = float(np.median(rank_max(timed(P.host_fed_step, args.steps, max(1, min(args.repeats, 3))))))
        h2d_b, d2h_b = P.host_fed_bytes()
        n_chk, bad, tot = P.check_host_fed_against_oracle(np.arange(B) if world == 1 else np.unique(np.linspace(0, B - 1, min(B, 64)).astype(np.int64)))
        t_h2d = P.host_fed_copy_only(args.steps, "h2d") / args.steps
        t_d2h = P.host_fed_copy_only(args.steps, "d2h") / args.steps
        fps = B * args.steps / dth
        pcie_fps, comp_fps = B / max(t_h2d, t_d2h), B * args.steps / dt
        extra["host_fed"] = {"frames_per_s": round(fps * world, 1), "ms_per_step": round(dth / args.steps * 1e3, 4),
                             "h2d_GBps": round(h2d_b * args.steps / dth / 1e9, 2), "d2h_GBps": round(d2h_b * args.steps / dth / 1e9, 2),
                             "h2d_bytes_per_step": h2d_b, "d2h_bytes_per_step": d2h_b,
                             "copies_alone": {"h2d_ms_per_step": round(t_h2d * 1e3, 4), "h2d_GBps": round(h2d_b / t_h2d / 1e9, 2),
                                              "d2h_ms_per_step": round(t_d2h * 1e3, 4), "d2h_GBps": round(d2h_b / t_d2h / 1e9, 2)},
                             "kernels_alone_frames_per_s": round(comp_fps, 1), "pcie_alone_frames_per_s": round(pcie_fps, 1),
                             "overlap_efficiency": round(fps / min(pcie_fps, comp_fps), 4),
                             "parity": {"checked_frames": int(n_chk), "mismatches": len(bad), "first_mismatches": bad[:4], "source": "pinned host result buffers of the last step"},
                             "what": "the headline's step with its %d frames per step read from PINNED HOST memory and every result (key points, descriptors, counts, "
                                     "undistorted key points, matches) written back to pinned host memory: H2D (copy stream) / kernels (3 streams) / D2H (second copy "
                                     "stream) ordered by events only; overlap_efficiency = frames/s over min(copies alone, kernels alone).  Never `value`." % B}

    def leg_host_api_cv():
        # the call the reference actually makes: ORBextractor::operator()(cv::InputArray, ..., cv::OutputArray, vector<int>&) through the header-only
        # adapter, incl. mvImagePyramid on the host (19-px bordered levels, one pinned slab) — a C++ program (tools/extractor_cv_latency.cpp, built
        # by tools/build_lib.sh against the mock cv:: of tests/cpp/mock_orbslam3), run on two of this run's own frames
        import subprocess
        import tempfile
        exe = os.path.join(ROOT, "tools", "bin", "extractor_cv_latency")
        if not os.path.exists(exe):
            raise RuntimeError("tools/bin/extractor_cv_latency is not built (tools/build_lib.sh)")
        with tempfile.TemporaryDirectory() as td:
            fl, fr = os.path.join(td, "left.raw"), os.path.join(td, "right.raw")
            frames[0].tofile(fl)
            frames[1 % B].tofile(fr)
            env = dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(local_rank)))
            out_ = subprocess.run([exe, fl, str(W), str(H), str(NFEAT), fr], capture_output=True, text=True, timeout=300, env=env)
        if out_.returncode != 0:
            raise RuntimeError("extractor_cv_latency rc %d: %s %s" % (out_.returncode, out_.stdout[-300:], out_.stderr[-300:]))
        r_ = json.loads(out_.stdout.strip().splitlines()[-1])
        r_["what"] = ("orbslam3_hip::ORBextractor::operator() (reference signature, Frame.cc:488-495) per %dx%d host image, synchronous, PCIe inclusive: with "
                      "mvImagePyramid brought to the host in the reference's bordered layout / without (integration/Frame_hip.cc linked); "
                      "Frame::ComputeStereoMatches through the adapter on the two calls' device-resident outputs / through the upload path" % (W, H))
        extra["host_api_cv"] = r_

    def leg_lba():
        # ---- extra leg 2: LocalBundleAdjustment linearisations (C5-size windows: 100 KF / 20k landmarks)
        from orbhip.lba import LbaWindows, synth_window
        nwin = args.lba_windows
        wins, cams = [], None
        for i in range(min(nwin, 2)):
            w, cams = synth_window(100 + i + 10 * rank, 100, 20, 20000, 8, "mono")
            wins.append(w)
        wins = [wins[i % len(wins)] for i in range(nwin)]
        Lw = LbaWindows(wins, cams, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))
        outs = ("Hpp", "bp", "Hll", "bl", "Hpl", "chi2")
        for _ in range(2):
            Lw.build_system(outs)
        barrier()
        lsteps = max(3, args.steps // 2)
        t2 = time.perf_counter()
        for _ in range(lsteps):
            Lw.build_system(outs)
        barrier()
        dtl = time.perf_counter() - t2
        E = float(np.mean([len(w["edges"]) for w in wins]))
        lba_bytes = E * (28 + 144) + 20000 * (24 + 72 + 24) + 80 * (56 + 288 + 48)    # SURVEY.md §8(d) A_lba with the realised E
        extra["lba"] = {"linearizations_per_s": round(nwin * lsteps / dtl, 1), "ms_per_step": round(dtl / lsteps * 1e3, 4), "windows_per_step": nwin,
                        "edges_per_window": E, "algorithmic_GBps": round(lba_bytes * nwin * lsteps / dtl / 1e9, 2),
                        "hbm_frac": round(lba_bytes * nwin * lsteps / dtl / 1e9 / HBM_PEAK_GBS, 5),
                        "what": "BlockSolver::buildSystem equivalent (residuals, Huber, Jacobians, Hpp/Hll/Hpl/b) for 100-KF/20k-landmark windows"}
        # full LM iterations: optimizer.optimize(5) per window = linearise + Schur + Cholesky + update + rho test, GPU resident
        if args.lm_windows != nwin:   # the LM leg batches more windows: its dense Cholesky is one workgroup per window
            Lw = LbaWindows([wins[i % len(wins)] for i in range(args.lm_windows)], cams, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))
        p0, x0 = Lw.d["poses"].clone(), Lw.d["points"].clone()
        Lw.optimize(5)
        barrier()
        osteps = 2
        t3 = time.perf_counter()
        for _ in range(osteps):
            Lw.d["poses"].copy_(p0); Lw.d["points"].copy_(x0)
            stats = Lw.optimize(5)
        barrier()
        dto = time.perf_counter() - t3
        extra["lba"]["lm_iterations_per_s"] = round(float(stats[:, 0].sum()) * osteps / dto, 1)
        extra["lba"]["lm_ms_per_optimize5_batch"] = round(dto / osteps * 1e3, 3)
        extra["lba"]["lm_trials_per_window"] = float(stats[:, 3].mean())
        extra["lba"]["lm_windows_per_step"] = args.lm_windows
        # one window per call: what LocalMapping's single LocalBundleAdjustment sees (latency: the dense Cholesky runs as one launch per phase)
        L1 = LbaWindows([wins[0]], cams, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))
        q0, y0 = L1.d["poses"].clone(), L1.d["points"].clone()
        L1.optimize(5)
        barrier()
        t4 = time.perf_counter()
        for _ in range(5):
            L1.d["poses"].copy_(q0); L1.d["points"].copy_(y0)
            L1.optimize(5)
        barrier()
        extra["lba"]["lm_single_window_ms_per_optimize5"] = round((time.perf_counter() - t4) / 5 * 1e3, 3)
        del L1
        # the same LM on windows of the other edge families (the monocular windows above run the monocular-pinhole kernel instantiation):
        # stereo / RGB-D pinhole maps (monocular + stereo edges: the pinhole instantiation) and fisheye maps (the generic kernels)
        for fam, kind in (("stereo_pinhole", "stereo"), ("fisheye", "kb8")):
            nw = max(8, args.lm_windows // 4)
            fw, fcams = [], None
            for i in range(2):
                w_, fcams = synth_window(300 + i + 10 * rank, 100, 20, 20000, 8, kind)
                fw.append(w_)
            Lf = LbaWindows([fw[i % 2] for i in range(nw)], fcams, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))
            fp0, fx0 = Lf.d["poses"].clone(), Lf.d["points"].clone()
            Lf.optimize(5)
            barrier()
            tf = time.perf_counter()
            Lf.d["poses"].copy_(fp0); Lf.d["points"].copy_(fx0)
            fst = Lf.optimize(5)
            barrier()
            dtf = time.perf_counter() - tf
            extra["lba"]["lm_iterations_per_s_" + fam] = round(float(fst[:, 0].sum()) / dtf, 1)
            extra["lba"]["lm_windows_" + fam] = nw
            del Lf
        if world == 1 and not args.no_cpu_baseline:   # the oracle's LM (reference algorithm restated, dense Schur/Cholesky) on one host core
            import oracle_lib as O
            from orbhip.lba import HUBER_MONO, HUBER_STEREO
            tc = time.perf_counter()
            _, _, ost = O.lba_optimize(wins[0], cams, (HUBER_MONO, HUBER_STEREO), 2)
            extra["lba"]["cpu_port_lm_iterations_per_s_1core"] = round(float(ost[0]) / (time.perf_counter() - tc), 2)

    def leg_pose_optimization():
        # ---- extra leg 3: Optimizer::PoseOptimization, one workgroup per frame, 4 x optimize(10) in a single launch
        from orbhip.lba import pose_optimization, synth_pose_frames
        pf = synth_pose_frames(seed=40 + rank, batch=64, n_pts=400, kind="stereo")
        PB = 2048
        rep = lambda a: torch.from_numpy(np.ascontiguousarray(np.concatenate([a] * (PB // 64)))).to(dev)
        pP, pE, pN = rep(pf["poses"]), rep(pf["edges"].view(np.uint8).reshape(64, -1)), rep(pf["n_edges"])
        pC = torch.from_numpy(np.ascontiguousarray(pf["cameras"]).view(np.uint8)).to(dev)
        pose_optimization(pP, pE, pN, pC, pinhole=True)   # stereo / monocular edges on pinhole cameras
        barrier()
        psteps = 3
        t4 = time.perf_counter()
        for _ in range(psteps):
            po = pose_optimization(pP, pE, pN, pC, pinhole=True)
        barrier()
        dtp = time.perf_counter() - t4
        extra["pose_optimization"] = {"frames_per_s": round(PB * psteps / dtp, 1), "ms_per_batch": round(dtp / psteps * 1e3, 3), "frames_per_batch": PB,
                                      "edges_per_frame": float(pf["n_edges"].mean()), "mean_inliers": float(po[2].float().mean().item()),
                                      "what": "Optimizer::PoseOptimization (4 rounds x LM optimize(10), outlier re-classification) per frame"}
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib as O
            tc = time.perf_counter()
            for b in range(16):
                O.pose_optimize(pf["poses"][b], pf["edges"][b, :pf["n_edges"][b]], pf["cameras"])
            extra["pose_optimization"]["cpu_port_frames_per_s_1core"] = round(16 / (time.perf_counter() - tc), 1)

    def leg_inertial_ba():
        # ---- extra leg 3b (SURVEY N4 tail): Optimizer::LocalInertialBA windows, one workgroup per window, optimize(10) in a single launch
        from orbhip.inertial import InertialWindows, synth_inertial_window
        iw = [synth_inertial_window(60 + i + 10 * rank, n_opt=10, n_fixed_vis=6, n_pts=1200, max_obs=8, kind="stereo") for i in range(2)]
        IB = 512
        td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        IWn = InertialWindows([iw[i % 2] for i in range(IB)], td)
        kf0, pt0 = IWn.d["kfs"].clone(), IWn.d["points"].clone()
        IWn.optimize(1.0, 10)
        barrier()
        t4b = time.perf_counter()
        for _ in range(2):
            IWn.d["kfs"].copy_(kf0); IWn.d["points"].copy_(pt0)
            ist = IWn.optimize(1.0, 10)
        barrier()
        dti = (time.perf_counter() - t4b) / 2
        ist = ist.cpu().numpy()
        extra["inertial_ba"] = {"windows_per_s": round(IB / dti, 1), "lm_iterations_per_s": round(float(ist[:, 0].sum()) / dti, 1),
                                "ms_per_batch": round(dti * 1e3, 3), "windows_per_batch": IB, "edges_per_window": float(np.mean([len(w["edges"]) for w in iw])),
                                "opt_keyframes": 10, "points_per_window": 1200, "chi2_drop": float((ist[:, 1] / ist[:, 4]).mean()),
                                "what": "Optimizer::LocalInertialBA optimize(10): EdgeInertial/GyroRW/AccRW + EdgeMono/EdgeStereo, LM + Schur + Cholesky"}
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib as O
            from orbhip.lba import HUBER_MONO, HUBER_STEREO
            tc = time.perf_counter()
            _, _, ost = O.inertial_optimize(iw[0], (HUBER_MONO, HUBER_STEREO), 1.0, 10)
            extra["inertial_ba"]["cpu_port_windows_per_s_1core"] = round(1.0 / (time.perf_counter() - tc), 2)

    def leg_pose_inertial():
        # ---- extra leg 3c: Optimizer::PoseInertialOptimizationLastKeyFrame, one wave per frame, 4 x 10 Gauss-Newton in a single launch
        from orbhip.inertial import pose_inertial_optimization_last_keyframe, synth_inertial_frame
        from orbhip.lba import POSE_EDGE_DTYPE
        pfs = [synth_inertial_frame(80 + i + 8 * rank, 300, "stereo") for i in range(8)]
        PIB = 4096
        capE = max(len(f["edges"]) for f in pfs)
        pe = np.zeros((PIB, capE), POSE_EDGE_DTYPE); pn = np.zeros(PIB, np.int32)
        for b2 in range(PIB):
            f = pfs[b2 % 8]
            pe[b2, :len(f["edges"])] = f["edges"]; pn[b2] = len(f["edges"])
        pfr = np.concatenate([pfs[b2 % 8]["frame"] for b2 in range(PIB)]); pkf = np.concatenate([pfs[b2 % 8]["keyframe"] for b2 in range(PIB)])
        pim = np.concatenate([pfs[b2 % 8]["imu"] for b2 in range(PIB)])
        tdv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        from orbhip.inertial import PoseInertialBatch, synth_prior
        rates = {}
        for name, pri in (("last_keyframe", None), ("last_frame", np.concatenate([synth_prior(pfs[b2 % 8]["keyframe"][0], b2 % 8) for b2 in range(PIB)]))):
            PBt = PoseInertialBatch(pfr, pkf, pfs[0]["rig"], pe, pn, pim, tdv, priors=pri)
            PBt.run()
            barrier()
            t4c = time.perf_counter()
            for _ in range(3):
                pg = PBt.run()
            barrier()
            rates[name] = (PIB / ((time.perf_counter() - t4c) / 3), float(pg.float().mean().item()))
        extra["pose_inertial"] = {"last_keyframe_frames_per_s": round(rates["last_keyframe"][0], 1), "last_frame_frames_per_s": round(rates["last_frame"][0], 1),
                                  "frames_per_batch": PIB, "edges_per_frame": float(pn.mean()), "mean_inliers": rates["last_keyframe"][1],
                                  "what": "Optimizer::PoseInertialOptimizationLastKeyFrame / ...LastFrame (4 x 10 Gauss-Newton over 15 / 30 unknowns, re-classification, "
                                          "prior Hessian / marginalisation) per frame, device resident"}
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib as O
            tc = time.perf_counter()
            for b2 in range(8):
                O.pose_inertial_kf(pfs[b2]["frame"], pfs[b2]["keyframe"], pfs[b2]["rig"], pfs[b2]["edges"], pfs[b2]["imu"])
            extra["pose_inertial"]["cpu_port_frames_per_s_1core"] = round(8 / (time.perf_counter() - tc), 1)

    def leg_bow():
        # ---- extra leg 4 (SURVEY N2 + M6, BASELINE configs[2] shape): Frame::ComputeBoW on a k=10, L=6 vocabulary (the stock ORBvoc shape,
        #      synthetic node descriptors) followed by SearchByBoW of every frame pair, all on the device CSRs
        from orbhip.bow import ORBVocabulary, synth_vocabulary_fast
        kps, desc = out[0], out[1]
        counts_b = out[2].cpu().numpy()      # (the batch `out` holds: the input set the last extraction read)
        voc = ORBVocabulary(synth_vocabulary_fast(5, 10, 6, sample_desc=desc[0, :int(counts_b[0, 0])].cpu().numpy()), device=dev.index or 0)
        nfeat = out[2][:, 0].contiguous()
        bw = voc.transform(desc, nfeat, 4)
        barrier()
        t5 = time.perf_counter()
        for _ in range(3):
            bw = voc.transform(desc, nfeat, 4)
        barrier()
        dtb = (time.perf_counter() - t5) / 3
        ang = kps[:, :, 3].contiguous()
        sideA = dict(desc=desc[0::2].contiguous(), angle=ang[0::2].contiguous(), node_id=bw["fv_node_id"][0::2].contiguous(),
                     node_start=bw["fv_node_start"][0::2].contiguous(), feat_idx=bw["fv_feat_idx"][0::2].contiguous(), n_nodes=bw["fv_n_nodes"][0::2].contiguous())
        sideB = dict(desc=desc[1::2].contiguous(), angle=ang[1::2].contiguous(), node_id=bw["fv_node_id"][1::2].contiguous(),
                     node_start=bw["fv_node_start"][1::2].contiguous(), feat_idx=bw["fv_feat_idx"][1::2].contiguous(), n_nodes=bw["fv_n_nodes"][1::2].contiguous())
        kvalid = torch.ones((B // 2, desc.shape[1]), dtype=torch.uint8, device=dev)
        mb = orbhip.ORBmatcher(0.7, True)
        fm, nmb = mb.SearchByBoW(sideA, kvalid, sideB)
        barrier()
        t6 = time.perf_counter()
        for _ in range(3):
            fm, nmb = mb.SearchByBoW(sideA, kvalid, sideB)
        barrier()
        dts = (time.perf_counter() - t6) / 3
        extra["bow"] = {"compute_bow_frames_per_s": round(B / dtb, 1), "compute_bow_ms_per_batch": round(dtb * 1e3, 3),
                        "search_by_bow_pairs_per_s": round((B // 2) / dts, 1), "search_by_bow_ms_per_batch": round(dts * 1e3, 3),
                        "mean_matches_per_pair": float(nmb.float().mean().item()), "mean_words_per_frame": float(bw["bv_n"].float().mean().item()),
                        "vocabulary": "synthetic k=10 L=6 (%d nodes, %d words), levelsup=4" % (voc.n_nodes, voc.n_words),
                        "extract_bow_match_frames_per_s": round(B / (dt_extract / args.steps + dtb + dts), 1)}

    def leg_stereo():
        nonlocal out
        # ---- extra leg 5 (row M9, BASELINE configs[2] is a stereo sequence): rectified stereo = extraction of the right image of every frame
        #      (a 12..40 px horizontal-disparity copy of the left one) + Frame::ComputeStereoMatches on the two pyramids
        from orbhip.extractor import stereo_matches
        exR = orbhip.ORBextractor(NFEAT, 1.2, 8, 20, 7, device=local_rank, max_batch=B)
        disp = 12 + (torch.arange(B, device=dev) % 8) * 4
        d_right = torch.stack([torch.roll(d_frames[i], shifts=-int(disp[i]), dims=1) for i in range(B)]).contiguous()
        outR = exR.extract_batch(d_right, (0, 1000))
        outL = ex.extract_batch(d_frames, (0, 1000), out=out)
        ur, dp = stereo_matches(ex, exR, outL, outR, 0.11, 47.9)
        barrier()
        t7 = time.perf_counter()
        for _ in range(3):
            ur, dp = stereo_matches(ex, exR, outL, outR, 0.11, 47.9)
        barrier()
        dtsm = (time.perf_counter() - t7) / 3
        t8 = time.perf_counter()
        for _ in range(3):
            outR = exR.extract_batch(d_right, (0, 1000), out=outR)
            outL = ex.extract_batch(d_frames, (0, 1000), out=outL)
            ur, dp = stereo_matches(ex, exR, outL, outR, 0.11, 47.9)
        barrier()
        dtst = (time.perf_counter() - t8) / 3
        nst = (ur[:, :NFEAT] > 0).sum(1).float().mean().item()
        extra["stereo"] = {"compute_stereo_matches_ms_per_batch": round(dtsm * 1e3, 3), "stereo_frames_per_s": round(B / dtst, 1),
                           "mean_stereo_points_per_frame": round(nst, 1),
                           "what": "2 x ORBextractor + Frame::ComputeStereoMatches per stereo frame (bf = 47.9, baseline 0.11 m)"}

    def leg_fisheye_stereo():
        # ---- extra leg 6 (BASELINE configs[3] shape: fisheye stereo, 1280x720, nFeatures = 1500): two extractors with lapping areas +
        #      Frame::ComputeStereoFishEyeMatches (2-NN, ratio test, KannalaBrandt8::TriangulateMatches); images are synthetic, so the
        #      triangulation gates see arbitrary geometry — the leg measures the arithmetic, not a calibration
        from orbhip.frame import ComputeStereoFishEyeMatches, FisheyeRig
        FB, FW, FH, FN = 128, 1280, 720, 1500
        from orbhip.synth import synth_image
        base = [torch.from_numpy(synth_image(900 + i + 16 * rank, FW, FH)).to(dev) for i in range(8)]
        fL = torch.stack([base[i % 8] for i in range(FB)]).contiguous()
        fR = torch.stack([torch.roll(base[i % 8], shifts=-(8 + 2 * (i % 8)), dims=1) for i in range(FB)]).contiguous()
        eL = orbhip.ORBextractor(FN, 1.2, 8, 20, 7, device=local_rank, max_batch=FB)
        eR = orbhip.ORBextractor(FN, 1.2, 8, 20, 7, device=local_rank, max_batch=FB)
        lap = (300, 980)
        kbp = [190.978 * 2.5, 190.973 * 2.5, FW / 2.0, FH / 2.0, 0.0034823894, 0.0007150348, -0.0020532361, 0.00020293673]
        rigF = FisheyeRig.make(kbp, kbp, np.eye(3), [0.1, 0.0, 0.0], [float(v) for v in (np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2])
        oL, oR = eL.extract_batch(fL, lap), eR.extract_batch(fR, lap)

        def fish():
            cl, cr = oL[2].view(-1), oR[2].view(-1)
            return ComputeStereoFishEyeMatches(oL[0], oL[1], cl, cl[1:], oR[0], oR[1], cr, cr[1:], rigF, count_stride=2)
        fm = fish()
        barrier()
        t9 = time.perf_counter()
        for _ in range(3):
            fm = fish()
        barrier()
        dtf = (time.perf_counter() - t9) / 3
        t10 = time.perf_counter()
        for _ in range(3):
            oL = eL.extract_batch(fL, lap, out=oL); oR = eR.extract_batch(fR, lap, out=oR)
            fm = fish()
        barrier()
        dtff = (time.perf_counter() - t10) / 3
        cn = oL[2].cpu().numpy()
        extra["fisheye_stereo"] = {"stereo_frames_per_s": round(FB / dtff, 1), "compute_stereo_fisheye_matches_ms_per_batch": round(dtf * 1e3, 3),
                                   "frames_per_batch": FB, "size": "%dx%d" % (FW, FH), "nfeatures": FN, "mean_keypoints_left": float(cn[:, 0].mean()),
                                   "mean_lapping_keypoints_left": float((cn[:, 0] - cn[:, 1]).mean()), "mean_matches": float(fm[4].float().mean().item()),
                                   "what": "2 x ORBextractor (lapping area 300..980) + Frame::ComputeStereoFishEyeMatches per fisheye stereo frame"}

    def leg_size_1280x720():
        # ---- north_star's second frame size: 1280x720, nFeatures = 1500 (TUM_512.yaml:62's feature count on BASELINE configs[3]'s frame shape),
        #      the SAME step (extract + undistort + grid + SearchByProjection on 3 streams) and the extraction alone, with their roofline fractions
        FW, FH, FN, FB = 1280, 720, 1500, 256
        fr = host_1280                       # 256 frames over 32 distinct seeded scenes (generated before HIP came up)
        dfr = torch.from_numpy(fr).to(dev)
        P2 = StepPipeline(dfr, FW, FH, FN, local_rank, streams=args.streams, frames_host=fr)
        k2 = P2.kernel_times(warm=2)
        P2.start_streams()
        for _ in range(3):
            P2.step()
        st = max(5, args.steps // 2)
        d2 = float(np.median(rank_max(timed(P2.step, st, 3))))
        chk = None
        if not args.no_parity_check:
            sel2 = np.unique(np.linspace(0, FB - 1, 32).astype(np.int64))
            n2, bad2, _ = P2.check_against_oracle(sel2, frames_host=fr)
            chk = {"checked_frames": int(n2), "mismatches": len(bad2), "first_mismatches": bad2[:2]}
        d2x = float(np.median(rank_max(timed(P2.extract_only_step, st, 3))))
        cnt2 = P2.out[2].cpu().numpy()
        we, pk2 = algorithmic_bytes(FW, FH, FN)
        wm, pkm2 = match_algorithmic_bytes(FN, FN)
        pk2.update(pkm2)
        dom2 = max(KNAMES, key=lambda k: k2.get(k, 0.0))
        fps2, fps2x = world * FB * st / d2, world * FB * st / d2x
        extra["size_1280x720"] = {"extract_match_frames_per_s": round(fps2, 1), "extract_frames_per_s": round(fps2x, 1), "ms_per_step": round(d2 / st * 1e3, 4),
                                  "frames_per_gpu_per_step": FB, "nfeatures": FN, "steps": st, "mean_keypoints": float(cnt2[:, 0].mean()),
                                  "mean_matches": float(P2.res[2].float().mean().item()), "fast_passes": P2.ex.last_fast_passes(),
                                  "whole_step_algorithmic_bytes_per_frame": we + wm,
                                  "whole_step_frac": round((we + wm) * fps2 / world / 1e9 / HBM_PEAK_GBS, 5),
                                  "whole_extract_frac": round(we * fps2x / world / 1e9 / HBM_PEAK_GBS, 5),
                                  "dominant_kernel": KNAMES[dom2], "dominant_kernel_ms": round(k2[dom2], 4),
                                  "dominant_kernel_frac": round(pk2[dom2] * FB / (k2[dom2] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                  "kernel_ms": {k: round(v, 4) for k, v in k2.items()},
                                  "per_kernel_frac": {KNAMES[k]: round(pk2[k] * FB / (k2[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) for k in KNAMES if k2.get(k, 0) > 0},
                                  "parity": chk}
        del P2

    def leg_batch_sweep():
        # ---- B in {64, 512, 4096} per GPU, plus the sizes a camera rig delivers per call (2, 8, 16).  The 512 input frames (185 MB) fit the 256 MiB Infinity Cache and are re-read every step;
        #      4096 frames (1.48 GB of input, 4.6 GB of pyramid per handle) cannot — the rate must not depend on that residency.
        sweep = {}
        for SB in (2, 8, 16, 64, 4096):
            dfs = grow_batch_on_device(torch.cat(d_sets) if SB > B else d_frames, SB)      # 4096 = the resident sets' distinct frames, then rolled copies
            st = max(3, args.steps // 4) if SB > B else args.steps
            # A call of few frames leaves most of the machine idle between its ~14 dependent launches: calls of up to 128 frames are pipelined over
            # more extractor handles / streams (the step is the same; 3 is the headline's count) — every count tried is reported, the best one is the entry
            by_streams, best = {}, None
            for nst in ((args.streams,) if SB > 128 else tuple(sorted({args.streams, 4, 5, 6}))):
                Pt = StepPipeline(dfs, W, H, NFEAT, local_rank, streams=nst)
                Pt.kernel_times(warm=1)
                Pt.start_streams()
                for _ in range(4):
                    Pt.step()
                dt_ = rank_max(timed(Pt.step, st, 3))
                by_streams[str(nst)] = round(world * SB * st / float(np.median(dt_)), 1)
                if best is None or float(np.median(dt_)) < float(np.median(best[1])):
                    if best is not None:
                        del best
                    best = (Pt, dt_, nst)
                else:
                    del Pt
                torch.cuda.empty_cache()
            Ps, ds, nst_best = best
            ent = {"frames_per_s": round(world * SB * st / float(np.median(ds)), 1), "ms_per_step": round(float(np.median(ds)) / st * 1e3, 4), "steps": st,
                   "frames_per_s_min": round(world * SB * st / max(ds), 1), "frames_per_s_max": round(world * SB * st / min(ds), 1),
                   "hip_streams": nst_best, "frames_per_s_by_hip_streams": by_streams,
                   "input_bytes": int(SB) * W * H, "fast_passes": Ps.ex.last_fast_passes()}
            if not args.no_parity_check:
                sels = np.unique(np.linspace(0, SB - 1, min(SB, 32)).astype(np.int64))
                ns, bads, _ = Ps.check_against_oracle(sels)
                ent["parity"] = {"checked_frames": int(ns), "mismatches": len(bads), "first_mismatches": bads[:2]}
            if SB <= 64:
                # the sizes a rig delivers per call (2-16 images): the same step as ONE graph launch (one stream), next to the eager three-stream figure
                Pg = StepPipeline(dfs, W, H, NFEAT, local_rank, streams=1)
                if Pg.capture_graph():
                    for _ in range(3):
                        Pg.graph_step()
                    dg = rank_max(timed(Pg.graph_step, st, 3))
                    ent["graph_replay"] = {"frames_per_s": round(world * SB * st / float(np.median(dg)), 1), "ms_per_step": round(float(np.median(dg)) / st * 1e3, 4)}
                    if not args.no_parity_check:
                        ng, badg, _ = Pg.check_against_oracle(sels)
                        ent["graph_replay"]["parity"] = {"checked_frames": int(ng), "mismatches": len(badg), "first_mismatches": badg[:2]}
                    # ... and K such graphs (K handles, K streams, the same frames) replayed round robin: one launch per call on the host AND calls that
                    # overlap on the device — what a rig's cameras, each with its own extractor, give the machine
                    KG = 4
                    Pk = [Pg] + [StepPipeline(dfs, W, H, NFEAT, local_rank, streams=1) for _ in range(KG - 1)]
                    if all(p_.capture_graph() for p_ in Pk[1:]):
                        sk = [torch.cuda.Stream(dfs.device) for _ in range(KG)]

                        def round_robin():
                            for p_, s_ in zip(Pk, sk):
                                with torch.cuda.stream(s_):
                                    p_.graph_step()
                        for _ in range(3):
                            round_robin()
                        torch.cuda.synchronize()
                        dk = rank_max(timed(round_robin, max(1, st // KG), 3))
                        nst = max(1, st // KG) * KG
                        ent["graph_replay"]["x%d_streams" % KG] = {"frames_per_s": round(world * SB * nst / float(np.median(dk)), 1),
                                                                   "ms_per_call": round(float(np.median(dk)) / nst * 1e3, 4)}
                        if not args.no_parity_check:
                            torch.cuda.synchronize()
                            nk, badk, _ = Pk[-1].check_against_oracle(sels)
                            ent["graph_replay"]["x%d_streams" % KG]["parity"] = {"checked_frames": int(nk), "mismatches": len(badk), "first_mismatches": badk[:2]}
                    del Pk
                else:
                    ent["graph_replay"] = {"error": Pg._graph_error}
                del Pg
            sweep[str(SB)] = ent
            del Ps, dfs, best
            torch.cuda.empty_cache()
        sweep[str(B)] = {"frames_per_s": round(world * B * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 4), "steps": args.steps,
                         "input_bytes": int(B) * W * H, "note": "the headline"}
        extra["batch_sweep"] = sweep

    def leg_scene_diversity():
        # ---- round-4 verdict, weak #2: the headline now reads --input-sets x B frames, every pair its own scene.  This leg times round 4's input (ONE
        #      resident set, 512 frames rolled out of 16 base scenes) with the same step, so the line shows what the input change alone did to the rate
        d16 = torch.from_numpy(host_16).to(dev)
        P16 = StepPipeline(d16, W, H, NFEAT, local_rank, streams=args.streams, frames_host=host_16)
        P16.kernel_times(warm=2)
        P16.start_streams()
        for _ in range(3):
            P16.step()
        d16s = rank_max(timed(P16.step, args.steps, 3))
        fps16 = world * B * args.steps / float(np.median(d16s))
        ent = {"headline_distinct_scenes": int(scenes * len(d_sets)), "headline_input_sets": len(d_sets), "headline_frames_per_s": round(world * B * args.steps / dt, 1),
               "r04_input_16_scenes_one_set_frames_per_s": round(fps16, 1), "ratio_headline_over_r04_input": round((world * B * args.steps / dt) / fps16, 4),
               "r04_input_fast_passes": P16.ex.last_fast_passes()}
        if not args.no_parity_check:
            sel16 = np.unique(np.linspace(0, B - 1, 32).astype(np.int64))
            n16, bad16, _ = P16.check_against_oracle(sel16)
            ent["r04_input_parity"] = {"checked_frames": int(n16), "mismatches": len(bad16), "first_mismatches": bad16[:2]}
        extra["scene_diversity"] = ent
        del P16, d16

    def leg_mixed_batch():
        # ---- round-4 verdict, item 3(b): B frames = 60 % textured, 20 % sparse, 10 % low-contrast, 5 % flat, 5 % uniform-noise scenes, shuffled — workgroups of one
        #      launch see very different amounts of work (empty cells and minThFAST retries next to full ones, octree levels far below / far above their quota)
        fm, kinds = host_mixed
        dm = torch.from_numpy(fm).to(dev)
        Pm = StepPipeline(dm, W, H, NFEAT, local_rank, streams=args.streams, frames_host=fm)
        km = Pm.kernel_times(warm=2)
        Pm.start_streams()
        for _ in range(3):
            Pm.step()
        dms = rank_max(timed(Pm.step, args.steps, 3))
        cm = Pm.out[2].cpu().numpy()[:, 0]
        nmm = Pm.res[2].cpu().numpy()
        kinds_a = np.array(kinds)
        ent = {"frames_per_s": round(world * B * args.steps / float(np.median(dms)), 1), "ms_per_step": round(float(np.median(dms)) / args.steps * 1e3, 4),
               "frames_per_s_min": round(world * B * args.steps / max(dms), 1), "frames_per_s_max": round(world * B * args.steps / min(dms), 1),
               "shares": dict(MIXED_SHARES), "fast_passes": Pm.ex.last_fast_passes(), "kernel_ms": {k: round(v, 4) for k, v in km.items()},
               "mean_keypoints_by_kind": {k: round(float(cm[kinds_a == k].mean()), 1) for k in sorted(set(kinds))},
               "mean_matches_by_kind": {k: round(float(nmm[kinds_a == k].mean()), 1) for k in sorted(set(kinds))},
               "frames_by_kind": {k: int((kinds_a == k).sum()) for k in sorted(set(kinds))}}
        if not args.no_parity_check:
            nmx, badm, totm = Pm.check_against_oracle(None)
            ent["parity"] = {"checked_frames": int(nmx), "mismatches": len(badm), "first_mismatches": badm[:2], "keypoints_compared": int(totm["keypoints"]),
                             "matches_compared": int(totm["matches"])}
        extra["mixed_batch"] = ent
        del Pm, dm

    def leg_exchange():
        # N > 1 only: the two exchange steps of the path on RCCL — descriptor blocks for cross-rank matching, and the
        # landmark-sharded LBA linearisation (all-reduce of the pose-side system, all-gather of the pose blocks).  The headline has no collective in
        # its data path: a failure here is reported as `extra.exchange_error` next to it.
        from orbhip import dist as D
        from orbhip.lba import LbaWindows, synth_window
        barrier()
        D.allgather_frame_blocks(out[0], out[1], out[2])
        barrier()
        tx = time.perf_counter()
        for _ in range(3):
            ak, ad, ac = D.allgather_frame_blocks(out[0], out[1], out[2])
        barrier()
        dtx = (time.perf_counter() - tx) / 3
        blk = out[0].shape[1] * 60 + 8
        # what every rank holds of ITSELF in the gathered slabs must be its own local slabs, bit for bit (class_id = -1 reads as NaN: compare bit patterns)
        own = bool(torch.equal(ak[rank * B:(rank + 1) * B].view(torch.int32), out[0].view(torch.int32)) and torch.equal(ad[rank * B:(rank + 1) * B], out[1])
                   and torch.equal(ac[rank * B:(rank + 1) * B], out[2]))
        own_all = rank_max([0.0 if own else 1.0])[0] == 0.0
        extra["exchange"] = {"allgather_frame_blocks_ms": round(dtx * 1e3, 3), "bytes_per_rank": int(B * blk), "own_block_equals_local_slabs_on_every_rank": own_all,
                             "GBps_into_each_rank": round((world - 1) * B * blk / dtx / 1e9, 2),
                             "frames_visible_to_each_rank": int(ak.shape[0]), "what": "one all_gather_into_tensor of per-frame blocks [desc|kps|count]",
                             "rccl_ranks": world if dist.get_backend() == "nccl" else 0, "world": world, "backend": dist.get_backend()}
        if args.exchange in ("peer", "both"):
            # the RCCL-free form: IPC handles once, then one pull per peer and slab.  Never run on more than one GPU before: a failure
            # here is reported in the line and does not fail the run (every rank takes the same branch: the handle exchange is a collective)
            try:
                px = D.PeerExchange(B, out[0].shape[1], dev)
                px.kps.copy_(out[0]); px.desc.copy_(out[1]); px.counts.copy_(out[2])
                barrier()
                px.allgather()
                barrier()
                tp = time.perf_counter()
                for _ in range(3):
                    pk, pd, pc = px.allgather()
                    barrier()
                dtp = (time.perf_counter() - tp) / 3
                same = bool(torch.equal(pk.view(torch.int32), ak.view(torch.int32)) and torch.equal(pd, ad) and torch.equal(pc, ac))   # (bit patterns: class_id = -1 reads as NaN)
                same = rank_max([0.0 if same else 1.0])[0] == 0.0                       # ... on every rank
                extra["exchange"]["peer"] = {"allgather_frame_blocks_ms": round(dtp * 1e3, 3), "GBps_into_each_rank": round((world - 1) * B * blk / dtp / 1e9, 2),
                                             "equal_to_rccl_result": same, "what": "orbd_allgather_frames_peer: hipMemcpyAsync pull per peer and slab over IPC handles, "
                                                                                   "one stream per peer (timed with a barrier per call)"}
                barrier()
                px.close()
            except Exception as err:   # noqa: BLE001
                extra["exchange"]["peer_error"] = "%s: %s" % (type(err).__name__, err)
        # ---- the consumer of the all-gather: every rank matches frames of ITS OWN against the same-numbered frames of rank (rank + 1) % world, read out of
        #      the gathered slabs — knnMatch(k = 2) (Frame.cc:1300) and SearchByBoW(KF, KF) (ORBmatcher.cc:984-1124).  Checked against ONE process: this
        #      rank regenerates the peer's first frames from their seeds, extracts them itself and matches against those
        from orbhip.bow import ORBVocabulary, synth_vocabulary_fast
        P._use_set(0)
        ox = P.ex.extract_batch(P.d_frames, P.LAP)
        barrier()
        gk, gd, gc = D.allgather_frame_blocks(ox[0], ox[1], ox[2])
        S = min(B, 64)
        capx = ox[0].shape[1]
        first = lambda t: t.reshape((world, B) + tuple(t.shape[1:]))[:, :S].reshape((world * S,) + tuple(t.shape[1:])).contiguous()
        sample = ox[1][0, :900].contiguous()
        dist.broadcast(sample, 0)                                  # one vocabulary for all ranks (replicated, like ORBvoc in every ORB-SLAM3 process)
        V = ORBVocabulary(synth_vocabulary_fast(5, 10, 4, sample_desc=sample.cpu().numpy()), device=local_rank)
        mm = orbhip.ORBmatcher(0.75, True)
        mine = (ox[0][:S].contiguous(), ox[1][:S].contiguous(), ox[2][:S].contiguous())
        gath = (first(gk), first(gd), first(gc))
        got = D.cross_rank_match(*mine, *gath, mm, V, rank=rank, world=world)
        barrier()
        tm = time.perf_counter()
        for _ in range(3):
            got = D.cross_rank_match(*mine, *gath, mm, V, rank=rank, world=world)
        barrier()
        dtm = (time.perf_counter() - tm) / 3
        peer, R = (rank + 1) % world, min(8, S)
        ref_frames = make_batch(R, seed0=1000 * peer, unique=max(1, R // 2))          # == the peer's frames 0..R-1 of input set 0 (make_batch is prefix-stable)
        exr = orbhip.ORBextractor(NFEAT, 1.2, 8, 20, 7, device=local_rank, max_batch=R)
        rk, rd, rc_ = exr.extract_batch(torch.from_numpy(ref_frames).to(dev), P.LAP)
        pad = lambda t: torch.cat([t, t.new_zeros((R, capx - t.shape[1]) + tuple(t.shape[2:]))], 1) if t.shape[1] < capx else t
        loc = [torch.cat([a[:R], pad(b_)]) for a, b_ in zip(mine[:2], (rk, rd))] + [torch.cat([mine[2][:R], rc_])]
        ref = D.cross_rank_match(mine[0][:R], mine[1][:R], mine[2][:R], loc[0], loc[1], loc[2], mm, V, rank=0, world=2)
        same = all(bool(torch.equal(got[k_][:R], ref[k_])) for k_ in ("knn_idx", "knn_dist", "bow_m12", "bow_nmatches"))
        same = rank_max([0.0 if same else 1.0])[0] == 0.0
        nn = got["knn_dist"][..., 0]
        extra["cross_rank_match"] = {"frame_pairs_per_rank": int(S), "ms": round(dtm * 1e3, 3), "frame_pairs_per_s": round(world * S / dtm, 1),
                                     "mean_bow_matches": float(got["bow_nmatches"].float().mean().item()),
                                     "mean_nearest_distance": float(nn[nn < 256].float().mean().item()),
                                     "equal_to_single_process_on_every_rank": same, "single_process_pairs_checked": int(R),
                                     "what": "frame i of this rank against frame i of rank (rank + 1) %% world out of the all-gathered slabs: orbm_knn2 + "
                                             "ComputeBoW (k = 10, L = 4 vocabulary) of both sides + SearchByBoW(KF, KF); the first %d pairs also against the peer's frames "
                                             "regenerated and extracted in THIS process" % R}
        w, cams = synth_window(77, 100, 20, 20000, 8, "mono")   # the same window on every rank, landmarks sharded
        llo, lhi = D.shard(len(w["points"]), rank, world)
        wl = D.shard_window_by_landmark(w, llo, lhi)
        Ls = LbaWindows([wl], cams, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))
        nfree = int((w["pose_hidx"] >= 0).sum())
        plo, phi = D.shard(nfree, rank, world)
        free_idx = torch.from_numpy(np.nonzero(w["pose_hidx"] >= 0)[0][plo:phi]).to(dev)

        def lba_step():
            o = Ls.build_system(("Hpp", "bp", "Hll", "bl", "Hpl", "chi2"))
            Hs, bs = D.allreduce_pose_system(o["Hpp"][0, :nfree], o["bp"][0, :nfree])
            pad = torch.zeros((-(-nfree // world), 7), dtype=torch.float64, device=dev)
            mine = Ls.d["poses"][0][free_idx]
            pad[:mine.shape[0]] = mine
            return Hs, D.allgather_pose_blocks(pad)
        lba_step()
        barrier()
        ty = time.perf_counter()
        for _ in range(5):
            Hs, allp = lba_step()
        barrier()
        dty = (time.perf_counter() - ty) / 5
        extra["lba_sharded"] = {"ms_per_linearization": round(dty * 1e3, 3), "linearizations_per_s": round(1.0 / dty, 1),
                                "edges_this_rank": int(len(wl["edges"])), "landmarks_this_rank": int(lhi - llo),
                                "what": "ONE 100-KF / 20k-landmark window, landmarks sharded over %d ranks: local build + all-reduce of H_pp/b_p "
                                        "(%d doubles) + all-gather of the pose blocks" % (world, nfree * 42)}
        # ---- the sharded LM step (lba_optimize_sharded): the same window, optimize(5) with the reduced camera system all-reduced per lambda trial,
        #      next to the whole window optimised on this rank alone — time reported as it is (one window does not scale: the collectives are latency)
        to_d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        one = LbaWindows([w], cams, to_d)
        p0 = one.d["poses"].clone(); x0 = one.d["points"].clone()
        one.optimize(5)
        barrier()
        t1 = time.perf_counter()
        for _ in range(2):
            one.d["poses"].copy_(p0); one.d["points"].copy_(x0)
            st1 = one.optimize(5)
        barrier()
        dt1 = (time.perf_counter() - t1) / 2
        sh = LbaWindows([wl], cams, to_d)
        ps0 = sh.d["poses"].clone(); xs0 = sh.d["points"].clone()
        sh.optimize_sharded(5)
        barrier()
        t2 = time.perf_counter()
        for _ in range(2):
            sh.d["poses"].copy_(ps0); sh.d["points"].copy_(xs0)
            st2 = sh.optimize_sharded(5)
        barrier()
        dt2 = (time.perf_counter() - t2) / 2
        dpose = float((one.d["poses"][0] - sh.d["poses"][0]).abs().max().item())
        dpts = float((one.d["points"][0, llo:lhi] - sh.d["points"][0, :lhi - llo]).abs().max().item())
        dmax = rank_max([dpose, dpts])
        extra["lm_sharded"] = {"ms_per_optimize5_sharded": round(dt2 * 1e3, 3), "ms_per_optimize5_one_rank": round(dt1 * 1e3, 3),
                               "iterations": [float(st1[0, 0]), float(st2[0, 0])], "lambda_trials": [float(st1[0, 3]), float(st2[0, 3])],
                               "chi2": [float(st1[0, 1]), float(st2[0, 1])], "max_abs_pose_difference": dmax[0], "max_abs_point_difference": dmax[1],
                               "allreduce_calls": sh.reduce_stats["calls"], "allreduce_doubles": sh.reduce_stats["doubles"],
                               "what": "ONE 100-KF / 20k-landmark window, optimize(5): landmarks sharded over %d ranks, per lambda trial one all-reduce of the reduced "
                                       "camera system (%d doubles) + 3 scalars, per linearisation one of H_pp | b_p; every rank factorises — vs the whole window on one "
                                       "rank" % (world, (nfree * 6) ** 2 + nfree * 6)}

    if not args.headline_only:
        legs = (("host_fed", leg_host_fed), ("host_api", leg_host_api), ("host_api_cv", leg_host_api_cv), ("lba", leg_lba), ("pose_optimization", leg_pose_optimization), ("inertial_ba", leg_inertial_ba),
                ("pose_inertial", leg_pose_inertial), ("bow", leg_bow), ("stereo", leg_stereo), ("fisheye_stereo", leg_fisheye_stereo), ("size_1280x720", leg_size_1280x720),
                ("scene_diversity", leg_scene_diversity), ("mixed_batch", leg_mixed_batch), ("batch_sweep", leg_batch_sweep))
        # N>1: the legs the multi-GPU line is read for (the metric's LBA component, north_star's second frame size); the per-GPU side figures are the
        # N=1 line's business and would only add run time and failure surface to a scaling run (--all-legs runs them anyway)
        keep = None if (world == 1 or args.all_legs) else ("lba", "size_1280x720")
        for name, fn in legs:
            if keep is None or name in keep:
                guard(name, fn)
    if world > 1 and (not args.headline_only or args.exchange != "rccl"):
        guard("exchange", leg_exchange)
    lba_ms = extra.get("lba", {}).get("ms_per_step", 0.0)
    if world > 1:   # max over ranks of every timed region (all ranks ran the same number of steps between the same barriers)
        t = torch.tensor([dt, dt_extract, lba_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_extract, lba_ms = float(t[0].item()), float(t[1].item()), float(t[2].item())
        if "lba" in extra and lba_ms > 0:
            extra["lba"]["linearizations_per_s"] = round(world * args.lba_windows / (lba_ms * 1e-3), 1)
        if "lba" in extra and "lm_iterations_per_s" in extra["lba"]:
            t2 = torch.tensor([extra["lba"]["lm_iterations_per_s"]], dtype=torch.float64, device=dev)
            dist.all_reduce(t2, op=dist.ReduceOp.MIN)
            extra["lba"]["lm_iterations_per_s"] = round(world * float(t2[0].item()), 1)

    if rank == 0:
        Nq = float(nq.mean())
        Nk = float(counts[:, 0].mean())
        whole_ext, pk = algorithmic_bytes(W, H, NFEAT)
        whole_match, pkm = match_algorithmic_bytes(NFEAT, NFEAT)    # SURVEY §8(d) figures are quoted at N = N_q = nFeatures
        pk.update(pkm)
        names = KNAMES
        dom = max(names, key=lambda k: kern.get(k, 0.0))
        ach = pk[dom] * B / (kern[dom] * 1e-3) / 1e9 if kern.get(dom, 0) > 0 else 0.0
        fps = world * B * args.steps / dt
        fps_extract = world * B * args.steps / dt_extract
        # HBM bytes per launch of the dominant kernel from the PMC passes committed with this build (profiles/pmc_latest.json): used only if
        # that file was produced from exactly these kernel sources and this workload, otherwise null (never a stale number)
        traffic, traffic_note = None, "no PMC pass for this build/workload"
        sq_info = {}
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            if pm.get("source_sha") != source_sha():
                traffic_note = "profiles/pmc_latest.json (%s) was collected on different kernel sources" % pm.get("tag")
            elif pm.get("workload") != [W, H, NFEAT, B]:
                traffic_note = "profiles/pmc_latest.json (%s) ran another workload" % pm.get("tag")
            else:
                traffic = pm["kernels"][names[dom]]["traffic_corrected"]
                sq_info = {k: pm["kernels"][names[dom]][k] for k in ("valu_insts_per_wave", "waves_per_simd", "simd_cycles_per_valu_inst", "lds_conflict_frac")
                           if k in pm["kernels"][names[dom]]}
                traffic_note = "profiles/%s_pmc.csv: FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, separate --pmc passes" % pm.get("tag")
                # the kernel's issue rate against the MEASURED VALU roof (tools/ubench/valu_rate.hip -> profiles/valu_rates.csv: 2.15 SIMD cycles per wave64
                # instruction for add / sub / logic / mov / right shifts / f32 add-mul-fma, 4.2 for every other 32-bit, packed, dot, DPP and 64-bit form,
                # 8.1 for rcp / sqrt / sin), priced on the kernel's static instruction mix (tools/valu_mix.py -> profiles/valu_mix.json)
                vm = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
                if vm.get("source_sha") == source_sha() and "simd_cycles_per_valu_inst" in sq_info:
                    inst = {"k_fast": "k_fast<0>", "k_sbp_frame": "k_sbp_frame<1>" if cap > 1024 else "k_sbp_frame<0>"}.get(names[dom], names[dom])
                    floor = vm["kernels"][inst]["floor_cycles_per_inst"]
                    sq_info["valu_floor_cycles_per_inst_static_mix"] = floor
                    sq_info["valu_issue_frac_of_measured_peak"] = round(floor / sq_info["simd_cycles_per_valu_inst"], 4)
                    sq_info["valu_mix"] = {k: vm["kernels"][inst][k] for k in ("valu", "full_rate", "half_rate", "quarter_rate")}
                    lw = vm["kernels"][inst].get("loop_weighted")
                    if lw:   # the same with every instruction weighted by its loop depth (tools/valu_mix.py): prologues no longer count like the hot loops
                        sq_info["valu_floor_cycles_per_inst_loop_weighted"] = lw["floor_cycles_per_inst"]
                        sq_info["valu_issue_frac_loop_weighted_floor"] = round(lw["floor_cycles_per_inst"] / sq_info["simd_cycles_per_valu_inst"], 4)
                        sq_info["valu_mix_loop_weighted"] = {k: lw[k] for k in ("loop_weight", "full_rate_share", "half_rate_share", "quarter_rate_share")}
        except Exception:   # noqa: BLE001
            pass
        traffic_raw = None
        if world == 1 and not args.no_pmc and not args.headline_only:
            # measured by THIS run (the pmc_latest.json figure above stays as the fallback and as the source of the SQ columns)
            t_corr, t_raw, t_note = measure_traffic_live(names[dom], B, "%dx%d" % (W, H), NFEAT)
            if t_corr is not None:
                traffic, traffic_raw, traffic_note = t_corr, t_raw, t_note
            else:
                traffic_note = "%s; fallback: %s" % (t_note, traffic_note)
        is_headline = (W, H, NFEAT) == (752, 480, 1000)
        parity_note = ("; the last timed step's %d frames bit-exact vs the CPU oracle" % parity["checked_frames"]) if parity["checked_frames"] > 0 and parity["mismatches"] == 0 else ""
        res = {
            "metric": BASELINE_METRIC if is_headline else "frames/sec ORB extract+match (%dx%d, %d kp)" % (W, H, NFEAT),
            "value": round(fps, 1), "unit": "frames/s",
            "value_is": "ORB extract+match frames/s: ORBextractor + UndistortKeyPoints + AssignFeaturesToGrid + SearchByProjection per frame, "
                        "median of %d timed regions of exactly --steps steps each" % args.repeats,
            "value_min": round(world * B * args.steps / max(dts), 1), "value_max": round(world * B * args.steps / min(dts), 1), "repeats": args.repeats,
            "region_ms": [round(v * 1e3, 3) for v in dts],
            "metric_components": {"orb_extract_match_frames_per_s": round(fps, 1),
                                  "orb_extract_frames_per_s": round(fps_extract, 1),
                                  "local_ba_linearizations_per_s": extra.get("lba", {}).get("linearizations_per_s"),
                                  "local_ba_lm_iterations_per_s": extra.get("lba", {}).get("lm_iterations_per_s")},
            "n_gpus": world, "rccl_ranks": (world if world > 1 and dist.get_backend() == "nccl" else (1 if world == 1 else 0)),
            "per_rank": {"frames_per_s": [round(v, 1) for v in per_rank_fps], "min": round(min(per_rank_fps), 1), "max": round(max(per_rank_fps), 1),
                         "what": "every rank's own median timed region (frames it processed / its own clock between the same barriers)"},
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "synthetic %dx%d grayscale batch, ORB extract (nFeatures=%d, 8 levels, 1.2, FAST 20/7) + SearchByProjection match "
                                   "(motion model, th=15) against the partner frame's %d points%s" % (W, H, NFEAT, int(round(Nq)), parity_note),
                       "parity_checked_frames": parity["checked_frames"], "parity_mismatches": parity["mismatches"], "parity": parity,
                       "frames_per_gpu_per_step": B, "input_sets": len(d_sets), "distinct_scenes_per_set": int(scenes), "input_seconds": round(input_seconds, 2),
                       "inputs": "%d resident input sets x %d frames per GPU, rotated step by step; every frame pair its own seeded scene (%d distinct scenes per set), "
                                 "partner = the scene moved by (%d, %d) px + sensor noise" % (len(d_sets), B, int(scenes), SHIFT[0], SHIFT[1]),
                       "mean_keypoints": Nk, "mean_queries": Nq, "mean_matches": float(nm.mean()),
                       "parallelism": "frames sharded, %d rank(s), no collective" % world, "world": world, "hip_streams": args.streams,
                       "backend": (dist.get_backend() if world > 1 else None)},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_raw": traffic_raw, "traffic_note": traffic_note,
                         "traffic_ratio": (round(traffic / (pk[dom] * B), 3) if traffic else None),        # corrected HBM bytes / algorithmic bytes of the dominant kernel
                         "describe_patch_bytes_per_frame": describe_patch_bytes(NFEAT),                       # what k_describe2 stages (43x43 per key point); the fraction uses SURVEY's 1 533 B
                         # SQ counters of the same profile: the kernel's occupancy and the SIMD cycles one wave-level VALU instruction costs it, next to the
                         # floor its instruction mix allows at the MEASURED issue rates (valu_issue_frac_of_measured_peak = floor / measured: 1 = the SIMDs
                         # issue back to back, the kernel's HBM fraction then follows from its instruction count alone)
                         "occupancy_and_issue": sq_info or None,
                         "algorithmic_bytes_per_launch": pk[dom] * B, "kernel_ms": round(kern[dom], 4),
                         # a batch runs k_fast as two launches — instance <0> detects at iniThFAST, instance <1> again the cells that came back empty
                         # (ORBextractor.cc:812-828) —, pyramid = the 7 k_resize2 launches: kernel_ms, the rocprofv3 figure (profiles/*_kernel_stats.csv,
                         # "sum of instances per batch") and traffic are per batch = the sum over the stage's launches
                         "launches_per_batch": {"k_fast": "k_fast<0> + k_fast<1>", "k_resize2": 7}.get(names[dom], 1),
                         "whole_step_algorithmic_bytes_per_frame": whole_ext + whole_match,
                         "whole_step_frac": round((whole_ext + whole_match) * fps / world / 1e9 / HBM_PEAK_GBS, 5),
                         "whole_extract_frac": round(whole_ext * fps_extract / world / 1e9 / HBM_PEAK_GBS, 5),
                         "per_kernel_frac": {names[k]: round(pk[k] * B / (kern[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) for k in names if kern.get(k, 0) > 0}},
            "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
            "extra": extra,
        }
        if world == 1 and not args.no_cpu_baseline:
            ns = min(64, B)
            res["cpu_baseline"] = cpu_baseline(frames[:ns], q[:ns].copy(), d_qdesc[:ns].cpu().numpy(), nq[:ns].copy(),
                                               np.array(list(CAM[:4]) + list(CAM[4]) + [0.0], np.float32), np.array(grid, np.float32))
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if parity["mismatches"]:
        sys.stderr.write("bench.py: the timed step's outputs differ from the oracle: %s\n" % parity.get("first_mismatches"))
        sys.exit(3)


if __name__ == "__main__":
    main()
