"""Run as a script in a fresh process (no torch: liborbd.so binds the system RCCL): ONE process drives every visible GPU, a host thread per GPU
(orbd_comm_init_all_local = ncclCommInitAll) — the exchange entry points of include/orbd.h with world = orb_device_count(): RCCL all-gather of the
frame slabs, all-reduce of a pose-side system, all-gather of pose blocks, and the RCCL-free peer-copy all-gather on raw peer pointers
(orbd_peer_enable_access + orbd_allgather_frames_peer), every rank checking every other rank's block.  On a one-GPU box this is the one-rank case;
on a multi-GPU node it is the first thing to run (RCCL with more than one rank has never been executed by this build)."""
import ctypes as C
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "awesome-orb-slam3-3dvisioncraft-version_amd")
H = C.CDLL(os.path.join(PKG, "liborbhip.so"))
D = C.CDLL(os.path.join(PKG, "liborbd.so"))
for f in (H.orb_dev_alloc, H.orb_dev_free, H.orb_memcpy_h2d, H.orb_memcpy_d2h, H.orb_stream_sync, H.orb_device_count):
    f.restype = C.c_int
H.orb_dev_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
H.orb_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
H.orb_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
H.orb_dev_free.argtypes = [C.c_void_p]
H.orb_stream_sync.argtypes = [C.c_void_p]
D.orbd_comm_init_all_local.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
D.orbd_allgather_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7
D.orbd_allreduce_pose_system.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
D.orbd_allgather_pose_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
D.orbd_comm_destroy.argtypes = [C.c_void_p]
D.orbd_peer_enable_access.argtypes = [C.c_int, C.POINTER(C.c_int)]
D.orbd_allgather_frames_peer.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7

N = int(os.environ.get("ORBD_WORLD", "0")) or H.orb_device_count()
assert N >= 1, "no GPU"
F, CAP, NFREE, PPR = 4, 100, 80, 10


def rank_data(r):
    rng = np.random.default_rng(500 + r)
    return (rng.random((F, CAP, 7)).astype(np.float32), rng.integers(0, 256, (F, CAP, 32), dtype=np.uint8), rng.integers(0, CAP, (F, 2)).astype(np.int32),
            rng.random((NFREE, 36)), rng.random((NFREE, 6)), rng.random((PPR, 7)))


devs = (C.c_int * N)(*range(N))
comms = (C.c_void_p * N)()
assert D.orbd_comm_init_all_local(N, devs, comms) == 0 and all(comms[i] for i in range(N))
slabs = [[None] * 3 for _ in range(N)]          # every rank's own slabs (raw device pointers: one process)
bar = threading.Barrier(N)
errors = []


def worker(r):
    try:
        def dev(a):
            p = C.c_void_p()
            assert H.orb_dev_alloc(r, a.nbytes, C.byref(p)) == 0         # (also makes device r this thread's current device)
            assert H.orb_memcpy_h2d(p, a.ctypes.data, a.nbytes, None) == 0
            return p

        def host(p, like):
            out = np.empty_like(like)
            assert H.orb_memcpy_d2h(out.ctypes.data, p, out.nbytes, None) == 0 and H.orb_stream_sync(None) == 0
            return out
        k, d, c, Hpp, bp, poses = rank_data(r)
        dk, dd, dc = dev(k), dev(d), dev(c)
        allk, alld, allc = np.zeros((N * F, CAP, 7), np.float32), np.zeros((N * F, CAP, 32), np.uint8), np.zeros((N * F, 2), np.int32)
        ok, od, oc = dev(allk), dev(alld), dev(allc)
        assert H.orb_stream_sync(None) == 0
        assert D.orbd_allgather_frames(comms[r], N, F, CAP, dk, dd, dc, ok, od, oc, None) == 0
        gk, gd, gc = host(ok, allk), host(od, alld), host(oc, allc)
        for s in range(N):
            k2, d2, c2 = rank_data(s)[:3]
            assert np.array_equal(gk[s * F:(s + 1) * F], k2) and np.array_equal(gd[s * F:(s + 1) * F], d2) and np.array_equal(gc[s * F:(s + 1) * F], c2), "RCCL all-gather block %d on rank %d" % (s, r)
        dH, db = dev(Hpp), dev(bp)
        assert D.orbd_allreduce_pose_system(comms[r], dH, db, NFREE, None) == 0
        sH, sb = sum(rank_data(s)[3] for s in range(N)), sum(rank_data(s)[4] for s in range(N))
        assert np.allclose(host(dH, Hpp), sH, rtol=1e-13, atol=0) and np.allclose(host(db, bp), sb, rtol=1e-13, atol=0)     # (summation order is RCCL's)
        allp = np.zeros((N * PPR, 7))
        dp, da = dev(poses), dev(allp)
        assert D.orbd_allgather_pose_blocks(comms[r], N, dp, da, PPR, None) == 0
        gp = host(da, allp)
        for s in range(N):
            assert np.array_equal(gp[s * PPR:(s + 1) * PPR], rank_data(s)[5])
        # ---- the same all-gather as peer copies on raw pointers
        slabs[r] = [dk.value, dd.value, dc.value]
        assert D.orbd_peer_enable_access(N, devs) == 0
        pk, pd, pc = dev(allk), dev(alld), dev(allc)
        assert H.orb_stream_sync(None) == 0
        bar.wait()                                   # every rank's slabs exist and are complete
        peer = [(C.c_void_p * N)(*[slabs[s][j] for s in range(N)]) for j in range(3)]
        assert D.orbd_allgather_frames_peer(N, r, F, CAP, peer[0], peer[1], peer[2], pk, pd, pc, None) == 0
        qk, qd, qc = host(pk, allk), host(pd, alld), host(pc, allc)
        assert np.array_equal(qk, gk) and np.array_equal(qd, gd) and np.array_equal(qc, gc), "peer all-gather differs from RCCL's on rank %d" % r
        bar.wait()                                   # nobody frees a slab a peer may still be reading
        assert D.orbd_peer_shutdown() == 0
        for p in (dk, dd, dc, ok, od, oc, dH, db, dp, da, pk, pd, pc):
            H.orb_dev_free(p)
    except BaseException as e:   # noqa: BLE001
        errors.append("rank %d: %r" % (r, e))
        try:
            bar.abort()
        except Exception:   # noqa: BLE001
            pass


ts = [threading.Thread(target=worker, args=(r,)) for r in range(N)]
for t in ts:
    t.start()
for t in ts:
    t.join(300)
for i in range(N):
    assert D.orbd_comm_destroy(comms[i]) == 0
if errors or any(t.is_alive() for t in ts):
    print("FAIL", errors)
    sys.exit(1)
print("orbd all-local exchange OK: world = %d (RCCL all-gather / all-reduce / pose all-gather + peer-copy all-gather, every block checked on every rank)" % N)
sys.exit(0)
