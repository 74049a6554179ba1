"""N>1 path on CPU: world_size-2 gloo processes (127.0.0.1).  Checks the frame sharding, the one-collective all-gather of
frame blocks used for cross-frame matching, and the landmark-sharded LBA pose-system reduction against a single-process run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from orbhip import dist as D
    from orbhip.lba import HUBER_MONO, HUBER_STEREO, synth_window
    from orbhip.synth import synth_image
    try:
        # ---- frames shard with no data-path collective; one all-gather brings every rank all descriptor blocks
        nframes, cap = 4, 340
        lo, hi = D.shard(nframes)
        assert (lo, hi) == ((0, 2), (2, 4))[rank]
        o = O.OrbOracle(300)
        kps = torch.zeros((hi - lo, cap, 7)); desc = torch.zeros((hi - lo, cap, 32), dtype=torch.uint8); cnt = torch.zeros((hi - lo, 2), dtype=torch.int32)
        for i, f in enumerate(range(lo, hi)):
            mono, k, d = o.extract(synth_image(50 + f, 320, 240, n_rect=80, n_disc=40), 0, 0)
            kps[i, :len(k)] = torch.from_numpy(k.view(np.float32).reshape(-1, 7).copy()); desc[i, :len(k)] = torch.from_numpy(d)
            cnt[i, 0], cnt[i, 1] = len(k), mono
        ak, ad, ac = D.allgather_frame_blocks(kps, desc, cnt)
        assert ak.shape == (nframes, cap, 7) and ac[:, 0].min() > 100
        for f in range(nframes):   # every rank now holds frame f exactly as its owner extracted it
            mono, k, d = o.extract(synth_image(50 + f, 320, 240, n_rect=80, n_disc=40), 0, 0)
            n = int(ac[f, 0])
            assert n == len(k) and int(ac[f, 1]) == mono
            assert np.array_equal(ak[f, :n].numpy().view(np.uint8).reshape(-1), k.view(np.uint8).reshape(-1)) and np.array_equal(ad[f, :n].numpy(), d)
        # ---- LBA sharded by landmark: local H_pp partials, all-reduce == single-process system; pose all-gather
        w, cams = synth_window(3, 10, 2, 240, 6, "mono")
        full = O.lba_build_system(w, cams, (HUBER_MONO, HUBER_STEREO))
        llo, lhi = D.shard(len(w["points"]))
        e = w["edges"]
        mine = (e["point"] >= llo) & (e["point"] < lhi)
        wl = dict(w, edges=e[mine].copy())
        part = O.lba_build_system(wl, cams, (HUBER_MONO, HUBER_STEREO))
        nf = full["nfree"]
        Hpp = torch.zeros((nf, 36), dtype=torch.float64); bp = torch.zeros((nf, 6), dtype=torch.float64)
        Hpp[:part["nfree"]] = torch.from_numpy(part["Hpp"][:part["nfree"]]); bp[:part["nfree"]] = torch.from_numpy(part["bp"][:part["nfree"]])
        Hs, bs = D.allreduce_pose_system(Hpp, bp)
        assert np.allclose(Hs.numpy(), full["Hpp"][:nf], rtol=1e-12, atol=1e-9) and np.allclose(bs.numpy(), full["bp"][:nf], rtol=1e-12, atol=1e-9)
        assert np.allclose(part["Hll"][llo:lhi], full["Hll"][llo:lhi], rtol=0, atol=0)   # landmark blocks are purely local
        plo, phi = D.shard(len(w["poses"]))
        allp = D.allgather_pose_blocks(torch.from_numpy(w["poses"][plo:phi].copy()))
        assert np.array_equal(allp.numpy(), w["poses"])
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
