"""N>1 path on CPU: world_size-2 gloo processes (127.0.0.1) running the PRODUCT kernels — the emulated build of the product sources
(tests/emu: liborbhip_emu.so, same C ABI) — on their shards: frame sharding, the one-collective all-gather of frame blocks, the cross-rank matching that
consumes the gathered slabs (BFMatcher knnMatch(2), Frame.cc:1300, and SearchByBoW(KF, KF), ORBmatcher.cc:984-1124), the landmark-sharded LBA
linearisation (all-reduce of the pose-side system) and the landmark-sharded Levenberg-Marquardt step (lba_optimize_sharded: all-reduce of the reduced
camera system, block_solver.hpp:381-432) — each against the same thing computed in ONE process, and against the oracle."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, emu_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import orbhip
    from orbhip import _lib
    from orbhip import dist as D
    from orbhip.bow import ORBVocabulary, synth_vocabulary
    from orbhip.lba import HUBER_MONO, HUBER_STEREO, LbaWindows, synth_window
    from orbhip.synth import synth_image
    lib = _lib.bind(ctypes.CDLL(emu_path))          # the product sources compiled for the CPU emulator (device memory == host memory)
    try:
        # ---- frames shard with no data-path collective: every rank extracts ITS frames with the product extractor; one all-gather brings every
        #      rank all blocks
        nframes, cap = 4, 340
        lo, hi = D.shard(nframes)
        assert (lo, hi) == ((0, 2), (2, 4))[rank]
        img = lambda f: synth_image(50 + f, 320, 240, n_rect=80, n_disc=40)
        ex = orbhip.ORBextractor(300, 1.2, 8, 20, 7, lib=lib)

        def slabs(frames):
            k_ = np.zeros((len(frames), cap, 7), np.float32); d_ = np.zeros((len(frames), cap, 32), np.uint8); c_ = np.zeros((len(frames), 2), np.int32)
            for i, f in enumerate(frames):
                mono, k, d = ex(img(f), None, (0, 0))
                k_[i, :len(k)] = k.view(np.float32).reshape(-1, 7); d_[i, :len(k)] = d
                c_[i] = (len(k), mono)
            return k_, d_, c_
        kps, desc, cnt = slabs(range(lo, hi))
        ak, ad, ac = [t.numpy() for t in D.allgather_frame_blocks(torch.from_numpy(kps), torch.from_numpy(desc), torch.from_numpy(cnt))]
        assert ak.shape == (nframes, cap, 7) and ac[:, 0].min() > 100
        o = O.OrbOracle(300)
        for f in range(nframes):   # every rank now holds frame f exactly as its owner extracted it — and as the oracle extracts it
            mono, k, d = o.extract(img(f), 0, 0)
            n = int(ac[f, 0])
            assert n == len(k) and int(ac[f, 1]) == mono
            assert np.array_equal(ak[f, :n].view(np.uint8).reshape(-1), k.view(np.uint8).reshape(-1)) and np.array_equal(ad[f, :n], d)
        # ---- the consumer of the all-gather: this rank's frames against the next rank's, out of the gathered slabs == the same in ONE process
        #      (all four frames extracted here) == the oracle's loops
        m = orbhip.ORBmatcher(0.75, True, lib=lib)
        blob = synth_vocabulary(3, 10, 3, sample_desc=ad[0, :int(ac[0, 0])])
        V = ORBVocabulary(blob, lib=lib)
        got = D.cross_rank_match(kps, desc, cnt, ak, ad, ac, m, V)
        lk, ld, lc = slabs(range(nframes))           # single process: everything local
        ref = D.cross_rank_match(lk[lo:hi], ld[lo:hi], lc[lo:hi], lk, ld, lc, m, V, rank=rank, world=world)
        for key in ("knn_idx", "knn_dist", "bow_m12", "bow_nmatches"):
            assert np.array_equal(got[key], ref[key]), key
        ov = O.OracleVocabulary(blob)
        peer0 = ((rank + 1) % world) * (hi - lo)
        nmatched = 0
        for i in range(hi - lo):
            na, nb = int(cnt[i, 0]), int(ac[peer0 + i, 0])
            oi, od = O.knn2(desc[i, :na], ad[peer0 + i, :nb])
            assert np.array_equal(got["knn_idx"][i, :na], oi) and np.array_equal(got["knn_dist"][i, :na], od)
            side = lambda t, k_, d_, n_: dict(desc=d_[:n_], angle=np.ascontiguousarray(k_[:n_, 3]), node_id=t["fv_node_id"][:t["fv_n_nodes"]],
                                              node_start=t["fv_node_start"][:t["fv_n_nodes"] + 1], feat_idx=t["fv_feat_idx"][:t["fv_node_start"][t["fv_n_nodes"]]],
                                              n_nodes=t["fv_n_nodes"])
            ta, tb = ov.transform(desc[i, :na], 4), ov.transform(ad[peer0 + i, :nb], 4)
            om, on = O.search_by_bow_kf(side(ta, kps[i], desc[i], na), np.ones(na, np.uint8), side(tb, ak[peer0 + i], ad[peer0 + i], nb), np.ones(nb, np.uint8),
                                        0.75, True)
            assert on == got["bow_nmatches"][i] and np.array_equal(got["bow_m12"][i, :na], om)
            nmatched += on
        assert nmatched > 0
        # ---- LBA sharded by landmark: the PRODUCT linearises this rank's landmarks; all-reduce of the H_pp / b_p partials == the product on the whole
        #      window in one process (== the oracle); landmark blocks are purely local; pose all-gather
        w, cams = synth_window(3, 10, 2, 240, 6, "mono")
        huber = (HUBER_MONO, HUBER_STEREO)
        full = {k_: np.asarray(v_) for k_, v_ in LbaWindows([w], cams, lib=lib, huber=huber).build_system(("Hpp", "bp", "Hll", "bl")).items()}
        ofull = O.lba_build_system(w, cams, huber)
        nf = ofull["nfree"]
        llo, lhi = D.shard(len(w["points"]))
        wl = D.shard_window_by_landmark(w, llo, lhi)
        part = {k_: np.asarray(v_) for k_, v_ in LbaWindows([wl], cams, lib=lib, huber=huber).build_system(("Hpp", "bp", "Hll", "bl")).items()}
        Hs, bs = D.allreduce_pose_system(torch.from_numpy(part["Hpp"][0, :nf].copy()), torch.from_numpy(part["bp"][0, :nf].copy()))
        assert np.allclose(Hs.numpy(), full["Hpp"][0, :nf], rtol=1e-12, atol=1e-9) and np.allclose(bs.numpy(), full["bp"][0, :nf], rtol=1e-12, atol=1e-9)
        assert np.allclose(Hs.numpy(), ofull["Hpp"][:nf], rtol=1e-10, atol=1e-9)
        assert np.array_equal(part["Hll"][0, :lhi - llo], full["Hll"][0, llo:lhi])   # landmark blocks are purely local: bit-identical
        plo, phi = D.shard(len(w["poses"]))
        allp = D.allgather_pose_blocks(torch.from_numpy(w["poses"][plo:phi].copy()))
        assert np.array_equal(allp.numpy(), w["poses"])
        # ---- the sharded LM step: every rank linearises and Schur-eliminates ITS landmarks, the reduced camera system is all-reduced, every rank
        #      factorises and updates all poses, back-substitutes its own landmarks == optimize() of the whole window in one process
        for kind in ("mono", "stereo"):
            w2, cams2 = synth_window(5, 10, 2, 240, 6, kind)
            one = LbaWindows([w2], cams2, lib=lib, huber=huber)
            st1 = one.optimize(4)
            llo, lhi = D.shard(len(w2["points"]))
            sh = LbaWindows([D.shard_window_by_landmark(w2, llo, lhi)], cams2, lib=lib, huber=huber)
            st2 = sh.optimize_sharded(4)
            assert st1[0, 0] == st2[0, 0] and st1[0, 3] == st2[0, 3], (st1, st2)                  # iterations, lambda trials
            assert abs(st1[0, 1] - st2[0, 1]) <= 1e-9 * st1[0, 1]                                # final chi2 (of the WHOLE window, on every rank)
            assert np.abs(np.asarray(one.d["poses"])[0] - np.asarray(sh.d["poses"])[0]).max() < 1e-9
            assert np.abs(np.asarray(one.d["points"])[0, llo:lhi] - np.asarray(sh.d["points"])[0, :lhi - llo]).max() < 1e-9
            op, ox, ost = O.lba_optimize(w2, cams2, huber, 4)
            assert ost[0] == st2[0, 0] and np.abs(np.asarray(sh.d["poses"])[0, :len(op)] - op).max() < 1e-6
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_exchange(emu_lib):
    import build_emu
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, build_emu.OUT)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
