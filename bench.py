#!/usr/bin/env python3
"""bench.py — hot-path throughput on MI355X.

One "step" = one pass of the hot path over one batch of synthetic frames already resident in HBM.
Workload at N=1: BASELINE.json configs[1] — synthetic 752x480 grayscale batch, ORBextractor, nFeatures=1000,
8 levels, scale 1.2, FAST 20/7 (EuRoC.yaml values).  Frames are independent units: with N>1 every rank owns
its own batch (weak scaling, no data-path collective) and `value` = frames all ranks processed / max-over-ranks time.

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` (dominant kernel, HIP-event timed) and
`cpu_baseline` (the oracle = reference algorithm restated, timed on this box's host cores; rank 0, N=1 only).
"""
import argparse
import json
import os
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
        "describe": N * (43 * 43 + 60),                # 43x43 neighbourhood in, 28 B keypoint + 32 B descriptor out
    }
    return whole, per_kernel


SHIFT = (6, -4)   # frame 2j+1 = frame 2j moved by (dx, dy) px + sensor noise: consecutive views of one scene


def make_batch(B, seed0=0, unique=16):
    from orbhip.synth import synth_image
    base = [synth_image(seed0 + i, W, H) for i in range(min(unique, max(1, B // 2)))]
    rng = np.random.default_rng(seed0 + 12345)
    frames = []
    for j in range((B + 1) // 2):
        k = j // len(base)
        a = np.roll(base[j % len(base)], (7 * k, 13 * k), (0, 1))
        b = np.clip(np.roll(a, (SHIFT[1], SHIFT[0]), (0, 1)).astype(np.int16) + rng.integers(-3, 4, a.shape), 0, 255).astype(np.uint8)
        frames += [a, b]
    return np.stack(frames[:B])


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


def cpu_baseline(frames, budget_s=10.0):
    """Oracle (reference algorithm restated, g++ -O3) on this host in native threads: one extractor per thread, one
    frame per thread at a time (the reference extracts one image on one thread, Frame.cc:111-114)."""
    import oracle_lib as O
    cores = os.cpu_count() or 1
    s1, _ = O.bench_extract_mt(frames, 1, 8)          # warm + calibrate
    fps1 = 8 / s1
    n1 = max(8, int(fps1 * budget_s / 3))
    s1, _ = O.bench_extract_mt(frames, 1, n1)
    fps1 = n1 / s1
    # thread-count sweep (allocator / page-fault contention makes "all hardware threads" slower than fewer on big hosts)
    best = None
    for nt in sorted({max(1, cores // 8), max(1, cores // 4), max(1, cores // 2), cores}):
        per_thread = max(4, int(fps1 * budget_s / 6))
        sN, kp = O.bench_extract_mt(frames, nt, per_thread)
        fpsN = nt * per_thread / sN
        if best is None or fpsN > best[0]:
            best = (fpsN, nt, per_thread, kp / (nt * per_thread))
    fpsN, nt, per_thread, meankp = best
    return {"value": round(fpsN, 2), "unit": "frames/s", "cores": nt, "kind": "port",
            "sample": "best of a thread-count sweep on a %d-thread host: %d native threads x %d frames (round-robin over %d frames of "
                      "the same synthetic 752x480 batch), oracle = reference algorithm restated, g++ -O3; single thread: %.2f frames/s "
                      "over %d frames; mean %.1f keypoints/frame" % (cores, nt, per_thread, len(frames), fps1, n1, meankp)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the extract+match and LBA legs")
    ap.add_argument("--lba-windows", type=int, default=16, help="LBA windows per GPU per step")
    ap.add_argument("--lm-windows", type=int, default=256, help="LBA windows per GPU per step in the full-LM leg")
    ap.add_argument("--size", default="752x480", help="frame size WxH (the headline is 752x480; 1280x720 is BASELINE configs[3]'s frame shape)")
    ap.add_argument("--nfeatures", type=int, default=1000, help="ORBextractor nFeatures (1500 with --size 1280x720)")
    args = ap.parse_args()
    global W, H, NFEAT
    W, H = [int(v) for v in args.size.lower().split("x")]
    NFEAT = args.nfeatures

    import torch
    import torch.distributed as dist
    import orbhip

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # ORBHIP_BENCH_BACKEND=gloo + ORBHIP_BENCH_ONE_DEVICE=1: run the N>1 code path with all ranks on ONE GPU (how the multi-rank legs
        # were exercised on the 1-GPU development box); the driver's runs use the defaults = RCCL, one GPU per rank
        backend = os.environ.get("ORBHIP_BENCH_BACKEND", "nccl")
        if os.environ.get("ORBHIP_BENCH_ONE_DEVICE") == "1":
            local_rank = 0
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    B = args.batch
    frames = make_batch(B, seed0=1000 * rank)
    d_frames = torch.from_numpy(frames).to(dev)
    ex = orbhip.ORBextractor(NFEAT, 1.2, 8, 20, 7, device=local_rank, max_batch=B)
    out = None

    def step():
        nonlocal out
        out = ex.extract_batch(d_frames, (0, 1000), out=out)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    kern = {k: 0.0 for k in ("pyramid", "fast", "octree", "describe", "total")}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # per-kernel device times (HIP events recorded on the launch stream inside the C ABI) — from one more, untimed step
    step()
    torch.cuda.synchronize()
    for k, v in ex.last_timing().items():
        kern[k] = v
    counts = out[2].cpu().numpy()

    extra = {}
    try:
        # ---- extra leg 0: the single-image host-buffer entry point (ORBextractor::operator() drop-in): PCIe + sync inclusive
        if not args.headline_only:
            ex1 = orbhip.ORBextractor(NFEAT, 1.2, 8, 20, 7, device=local_rank)
            ex1(frames[0], None, (0, 1000))
            th = time.perf_counter()
            nh = 100
            for i in range(nh):
                ex1(frames[i % B], None, (0, 1000))
            extra["host_api"] = {"frames_per_s": round(nh / (time.perf_counter() - th), 1),
                                 "what": "orbx_extract: one 752x480 host image per call, H2D 361 kB + 4 kernels + D2H 60 kB, synchronous (never `value`)"}
        # ---- extra leg 1: extract + match (grid build + motion-model SearchByProjection against the partner frame)
        if not args.headline_only:
            m = orbhip.ORBmatcher(0.9, True)
            cap = out[0].shape[1]
            # the Frame constructor's steps between extractor and matcher (EuRoC calibration): UndistortKeyPoints -> AssignFeaturesToGrid
            from orbhip.frame import Camera, FrameOps
            fo = FrameOps(Camera.make(458.654, 457.296, 367.215, 248.375, (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)), W, H)
            un = fo.UndistortKeyPoints(out[0], out[2].view(-1), count_stride=2)
            q, nq, src = build_match_queries(un.cpu().numpy(), counts, ex.GetScaleFactors(), cap)
            d_q = torch.from_numpy(q.view(np.uint8).reshape(B, cap, 28)).to(dev)
            d_nq = torch.from_numpy(nq).to(dev)
            d_qdesc = out[1][torch.from_numpy(src).to(dev)].contiguous()      # partner descriptors (prepared once, resident in HBM)
            grid = fo.grid
            work = torch.empty(m._L.orbm_search_workspace_bytes(B, cap), dtype=torch.uint8, device=dev)
            res = None

            def step_match():
                nonlocal out, res, un
                out = ex.extract_batch(d_frames, (0, 1000), out=out)
                cnt = out[2].view(-1)
                un = fo.UndistortKeyPoints(out[0], cnt, count_stride=2, out=un)
                gs, gi = m.grid_build(un, cnt, grid, count_stride=2)
                res = m.SearchByProjection(un, out[1], cnt, gs, gi, d_q, d_qdesc, d_nq, grid, 1, 100, count_stride=2, work=work)
            for _ in range(2):
                step_match()
            barrier()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t1 = time.perf_counter()
            msteps = max(3, args.steps // 4)
            for i in range(msteps):
                if i == msteps - 1:
                    out = ex.extract_batch(d_frames, (0, 1000), out=out)
                    ev0.record()
                    cnt = out[2].view(-1)
                    un = fo.UndistortKeyPoints(out[0], cnt, count_stride=2, out=un)
                    gs, gi = m.grid_build(un, cnt, grid, count_stride=2)
                    res = m.SearchByProjection(un, out[1], cnt, gs, gi, d_q, d_qdesc, d_nq, grid, 1, 100, count_stride=2, work=work)
                    ev1.record()
                else:
                    step_match()
            barrier()
            dtm = time.perf_counter() - t1
            nm = res[2].cpu().numpy()
            extra["extract_match"] = {"frames_per_s": round(B * msteps / dtm, 1), "ms_per_step": round(dtm / msteps * 1e3, 4),
                                      "match_only_ms": round(ev0.elapsed_time(ev1), 4), "mean_matches_per_frame": float(nm.mean()),
                                      "queries_per_frame": float(nq.mean()), "search": "UndistortKeyPoints (EuRoC k1,k2,p1,p2) + AssignFeaturesToGrid + SearchByProjection motion model th=15, TH_HIGH, rot. histogram"}
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
            if world == 1 and not args.no_cpu_baseline:   # the oracle's LM (reference algorithm restated, dense Schur/Cholesky) on one host core
                import oracle_lib as O
                from orbhip.lba import HUBER_MONO, HUBER_STEREO
                tc = time.perf_counter()
                _, _, ost = O.lba_optimize(wins[0], cams, (HUBER_MONO, HUBER_STEREO), 2)
                extra["lba"]["cpu_port_lm_iterations_per_s_1core"] = round(float(ost[0]) / (time.perf_counter() - tc), 2)
            # ---- extra leg 3: Optimizer::PoseOptimization, one workgroup per frame, 4 x optimize(10) in a single launch
            from orbhip.lba import pose_optimization, synth_pose_frames
            pf = synth_pose_frames(seed=40 + rank, batch=64, n_pts=400, kind="stereo")
            PB = 2048
            rep = lambda a: torch.from_numpy(np.ascontiguousarray(np.concatenate([a] * (PB // 64)))).to(dev)
            pP, pE, pN = rep(pf["poses"]), rep(pf["edges"].view(np.uint8).reshape(64, -1)), rep(pf["n_edges"])
            pC = torch.from_numpy(np.ascontiguousarray(pf["cameras"]).view(np.uint8)).to(dev)
            pose_optimization(pP, pE, pN, pC)
            barrier()
            psteps = 3
            t4 = time.perf_counter()
            for _ in range(psteps):
                po = pose_optimization(pP, pE, pN, pC)
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
            # ---- extra leg 3b (SURVEY N4 tail): Optimizer::LocalInertialBA windows, one workgroup per window, optimize(10) in a single launch
            try:
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
            except Exception as err:   # noqa: BLE001
                extra["inertial_ba_error"] = "%s: %s" % (type(err).__name__, err)
                sys.stderr.write(traceback.format_exc())
            # ---- extra leg 3c: Optimizer::PoseInertialOptimizationLastKeyFrame, one wave per frame, 4 x 10 Gauss-Newton in a single launch
            try:
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
            except Exception as err:   # noqa: BLE001
                extra["pose_inertial_error"] = "%s: %s" % (type(err).__name__, err)
                sys.stderr.write(traceback.format_exc())
            # ---- extra leg 4 (SURVEY N2 + M6, BASELINE configs[2] shape): Frame::ComputeBoW on a k=10, L=6 vocabulary (the stock ORBvoc shape,
            #      synthetic node descriptors) followed by SearchByBoW of every frame pair, all on the device CSRs
            from orbhip.bow import ORBVocabulary, synth_vocabulary_fast
            kps, desc = out[0], out[1]
            voc = ORBVocabulary(synth_vocabulary_fast(5, 10, 6, sample_desc=desc[0, :int(counts[0, 0])].cpu().numpy()), device=dev.index or 0)
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
                            "extract_bow_match_frames_per_s": round(B / (dt / args.steps + dtb + dts), 1)}
        # ---- extra leg 5 (row M9, BASELINE configs[2] is a stereo sequence): rectified stereo = extraction of the right image of every frame
        #      (a 12..40 px horizontal-disparity copy of the left one) + Frame::ComputeStereoMatches on the two pyramids
        if not args.headline_only:
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
        # ---- extra leg 6 (BASELINE configs[3] shape: fisheye stereo, 1280x720, nFeatures = 1500): two extractors with lapping areas +
        #      Frame::ComputeStereoFishEyeMatches (2-NN, ratio test, KannalaBrandt8::TriangulateMatches); images are synthetic, so the
        #      triangulation gates see arbitrary geometry — the leg measures the arithmetic, not a calibration
        if not args.headline_only:
            try:
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
            except Exception as err:   # noqa: BLE001
                extra["fisheye_stereo_error"] = "%s: %s" % (type(err).__name__, err)
                sys.stderr.write(traceback.format_exc())
    except Exception as err:   # an extra leg must never cost the headline line
        import traceback
        extra["error"] = "%s: %s" % (type(err).__name__, err)
        sys.stderr.write(traceback.format_exc())
    # ---- N > 1 only: the two exchange steps of the path on RCCL — descriptor blocks for cross-rank matching, and the
    #      landmark-sharded LBA linearisation (all-reduce of the pose-side system, all-gather of the pose blocks)
    if world > 1 and not args.headline_only:
        try:
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
            extra["exchange"] = {"allgather_frame_blocks_ms": round(dtx * 1e3, 3), "bytes_per_rank": int(B * blk),
                                 "GBps_into_each_rank": round((world - 1) * B * blk / dtx / 1e9, 2),
                                 "frames_visible_to_each_rank": int(ak.shape[0]), "what": "one all_gather_into_tensor of per-frame blocks [desc|kps|count]"}
            w, cams = synth_window(77, 100, 20, 20000, 8, "mono")   # the same window on every rank, landmarks sharded
            llo, lhi = D.shard(len(w["points"]), rank, world)
            e = w["edges"]
            wl = dict(w, edges=e[(e["point"] >= llo) & (e["point"] < lhi)].copy())
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
        except Exception as err:
            import traceback
            extra["exchange_error"] = "%s: %s" % (type(err).__name__, err)
            sys.stderr.write(traceback.format_exc())
    if world > 1:
        t = torch.tensor([dt] + [extra.get("extract_match", {}).get("ms_per_step", 0.0), extra.get("lba", {}).get("ms_per_step", 0.0)],
                         dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        if "extract_match" in extra and float(t[1].item()) > 0:
            extra["extract_match"]["frames_per_s"] = round(world * B / (float(t[1].item()) * 1e-3), 1)
        if "lba" in extra and float(t[2].item()) > 0:
            extra["lba"]["linearizations_per_s"] = round(world * args.lba_windows / (float(t[2].item()) * 1e-3), 1)

    if rank == 0:
        whole, per_kernel = algorithmic_bytes(W, H, NFEAT)
        dom = max(("pyramid", "fast", "octree", "describe"), key=lambda k: kern[k])
        ach = per_kernel[dom] * B / (kern[dom] * 1e-3) / 1e9 if kern[dom] > 0 else 0.0
        fps = world * B * args.steps / dt
        traffic = None  # HBM bytes per launch of the dominant kernel from the last committed PMC pass (profiles/pmc_latest.json)
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            if B == 512 and (W, H, NFEAT) == (752, 480, 1000):  # the PMC pass ran the default workload
                traffic = pm["kernels"]["k_" + dom]["traffic_corrected"]
        except Exception:
            pass
        res = {
            # BASELINE.json's metric string; `value` is its first, per-frame component on configs[1] (ORBextractor only, the configuration the
            # contract names for one GPU); the other components of the composite metric are in `metric_components`
            "metric": BASELINE_METRIC if (W, H, NFEAT) == (752, 480, 1000) else "frames/sec ORB extract (%dx%d, %d kp)" % (W, H, NFEAT),
            "value": round(fps, 1), "unit": "frames/s", "value_is": "ORBextractor frames/s (configs[1])",
            "metric_components": {"orb_extract_frames_per_s": round(fps, 1),
                                  "orb_extract_match_frames_per_s": extra.get("extract_match", {}).get("frames_per_s"),
                                  "local_ba_linearizations_per_s": extra.get("lba", {}).get("linearizations_per_s"),
                                  "local_ba_lm_iterations_per_s": extra.get("lba", {}).get("lm_iterations_per_s")},
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic %dx%d grayscale batch, ORBextractor only, nFeatures=%d, 8 levels, " % (W, H, NFEAT) +
                                   "bit-exact vs CPU oracle", "frames_per_gpu_per_step": B, "mean_keypoints": float(counts[:, 0].mean()),
                       "parallelism": "frames sharded, %d rank(s), no collective" % world},
            "roofline": {"bound": "hbm", "kernel": "k_" + dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": per_kernel[dom] * B, "kernel_ms": round(kern[dom], 4),
                         "whole_extract_frac": round(whole * fps / world / 1e9 / HBM_PEAK_GBS, 5)},
            "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
            "extra": extra,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(frames[:64])
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
