"""Run as a script on a GPU box: two processes (gloo rendezvous on 127.0.0.1) that BOTH use GPU 0 exercise the RCCL-free all-gather of
include/orbd.h — IPC export / open of each rank's slabs, one pull per peer and slab (orbd_allgather_frames_peer through orbhip.dist.PeerExchange)
— and check every rank ends up with both ranks' frames, rank-major.  Two GPUs would only change which link the copies travel on."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, CAP = 6, 300


def rank_data(r):
    rng = np.random.default_rng(100 + r)
    return (rng.random((F, CAP, 7)).astype(np.float32), rng.integers(0, 256, (F, CAP, 32), dtype=np.uint8), rng.integers(0, CAP, (F, 2)).astype(np.int32))


def worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from orbhip.dist import PeerExchange
        torch.cuda.set_device(0)
        with PeerExchange(F, CAP, torch.device("cuda", 0)) as px:
            for rep in range(3):
                k, d, c = rank_data(rank + 10 * rep)
                px.kps.copy_(torch.from_numpy(k)); px.desc.copy_(torch.from_numpy(d)); px.counts.copy_(torch.from_numpy(c))
                torch.cuda.synchronize(); dist.barrier()
                ak, ad, ac = px.allgather()
                torch.cuda.synchronize(); dist.barrier()
                for r in range(world):
                    k2, d2, c2 = rank_data(r + 10 * rep)
                    assert np.array_equal(ak[r * F:(r + 1) * F].cpu().numpy(), k2) and np.array_equal(ad[r * F:(r + 1) * F].cpu().numpy(), d2)
                    assert np.array_equal(ac[r * F:(r + 1) * F].cpu().numpy(), c2)
            held = px.kps[1]                       # a view a caller still holds when the exchange is closed (the with-block's exit is the collective close)
            expect = held.cpu().numpy().copy()
        assert px.kps is None and px._closed
        assert np.array_equal(held.cpu().numpy(), expect)      # the slab lives as long as a tensor on it: no dangling device pointer
        del held
        import ctypes as C
        L = C.CDLL(os.path.join(ROOT, "awesome-orb-slam3-3dvisioncraft-version_amd", "liborbd.so"))
        assert L.orbd_peer_shutdown() == 0         # the calling thread's per-peer copy streams, drained and released
        q.put((rank, "ok"))
    except Exception:   # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 2000
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    print(res)
    if res == [(0, "ok"), (1, "ok")]:
        print("orbd peer exchange OK")
        sys.exit(0)
    sys.exit(1)
