"""Run as a script in a fresh process (no torch: liborbd.so binds the system RCCL): the exchange entry points of include/orbd.h on a one-rank
communicator on GPU 0 — all-gather of frame slabs, all-reduce of a pose-side system, all-gather of pose blocks; with world = 1 every output must
equal its input, which checks the argument plumbing, the group launches and the stream ordering against the liborbhip.so copies."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "awesome-orb-slam3-3dvisioncraft-version_amd")
H = C.CDLL(os.path.join(PKG, "liborbhip.so"))
D = C.CDLL(os.path.join(PKG, "liborbd.so"))
for f in (H.orb_dev_alloc, H.orb_dev_free, H.orb_memcpy_h2d, H.orb_memcpy_d2h, H.orb_stream_sync):
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


def dev(a):
    p = C.c_void_p()
    assert H.orb_dev_alloc(0, a.nbytes, C.byref(p)) == 0
    assert H.orb_memcpy_h2d(p, a.ctypes.data, a.nbytes, None) == 0
    return p


def host(p, like):
    out = np.empty_like(like)
    assert H.orb_memcpy_d2h(out.ctypes.data, p, out.nbytes, None) == 0 and H.orb_stream_sync(None) == 0
    return out


rng = np.random.default_rng(0)
F, cap = 4, 100
kps = rng.random((F, cap, 7)).astype(np.float32)           # orb_keypoint = 28 bytes
desc = rng.integers(0, 256, (F, cap, 32), dtype=np.uint8)
cnt = rng.integers(0, cap, (F, 2)).astype(np.int32)
comm = C.c_void_p()
devs = (C.c_int * 1)(0)
assert D.orbd_comm_init_all_local(1, devs, C.byref(comm)) == 0 and comm.value
dk, dd, dc = dev(kps), dev(desc), dev(cnt)
ok, od, oc = dev(np.zeros_like(kps)), dev(np.zeros_like(desc)), dev(np.zeros_like(cnt))
assert D.orbd_allgather_frames(comm, 1, F, cap, dk, dd, dc, ok, od, oc, None) == 0
assert np.array_equal(host(ok, kps), kps) and np.array_equal(host(od, desc), desc) and np.array_equal(host(oc, cnt), cnt)
Hpp, bp = rng.random((80, 36)), rng.random((80, 6))
dH, db = dev(Hpp), dev(bp)
assert D.orbd_allreduce_pose_system(comm, dH, db, 80, None) == 0
assert np.array_equal(host(dH, Hpp), Hpp) and np.array_equal(host(db, bp), bp)
poses = rng.random((10, 7))
dp, da = dev(poses), dev(np.zeros_like(poses))
assert D.orbd_allgather_pose_blocks(comm, 1, dp, da, 10, None) == 0
assert np.array_equal(host(da, poses), poses)
assert D.orbd_allgather_frames(None, 1, F, cap, dk, dd, dc, ok, od, oc, None) == -3        # ORB_E_INVALID
assert D.orbd_comm_destroy(comm) == 0
for p in (dk, dd, dc, ok, od, oc, dH, db, dp, da):
    H.orb_dev_free(p)
print("orbd single-rank exchange OK")
sys.exit(0)
