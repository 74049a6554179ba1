"""Multi-GPU sharding of the hot path: one process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI on MI355X,
"gloo" in the CPU tests).

Frames, frame pairs and LBA windows are independent units: rank r owns the contiguous block `shard(n_units)` and runs the
stage kernels on it with no data-path collective.  The single exchange step of the path is cross-frame matching against frames
owned by other ranks: `allgather_frame_blocks` moves every rank's per-frame blocks
    [cap x 32 B descriptors | cap x 28 B keypoints | (n, monoIndex)]
in ONE collective per batch (F frames/rank -> F*60 kB per rank at cap=1000), because per-frame collectives would be pure
latency.  On xGMI (point-to-point, 7 links x ~153 GB/s) an all-gather of equal shards is per-link bound;
a 512-frame shard (31 MB) takes ~0.2 ms per peer link."""
import torch
import torch.distributed as dist


def shard(n_units, rank=None, world=None):
    """Contiguous block [lo, hi) of units owned by `rank` (unit i -> rank floor(i*world/n))."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo = -(-n_units * rank // world)
    hi = -(-n_units * (rank + 1) // world)
    return lo, hi


def pack_frame_blocks(kps, desc, counts):
    """kps [F,cap,7] f32, desc [F,cap,32] u8, counts [F,2] i32 -> one u8 tensor [F, cap*60 + 8]."""
    F, cap = kps.shape[0], kps.shape[1]
    return torch.cat([desc.reshape(F, cap * 32), kps.contiguous().view(torch.uint8).reshape(F, cap * 28),
                      counts.contiguous().view(torch.uint8).reshape(F, 8)], dim=1).contiguous()


def unpack_frame_blocks(blocks, cap):
    Ft = blocks.shape[0]
    desc = blocks[:, :cap * 32].reshape(Ft, cap, 32)
    kps = blocks[:, cap * 32:cap * 60].contiguous().view(torch.float32).reshape(Ft, cap, 7)
    counts = blocks[:, cap * 60:cap * 60 + 8].contiguous().view(torch.int32).reshape(Ft, 2)
    return kps, desc, counts


def allgather_frame_blocks(kps, desc, counts, group=None):
    """All ranks end up with the keypoints/descriptors/counts of ALL frames, in global frame order (rank-major).
    Every rank must contribute the same number of frames F (pad the last shard)."""
    world = dist.get_world_size(group)
    cap = kps.shape[1]
    mine = pack_frame_blocks(kps, desc, counts)
    out = torch.empty((world * mine.shape[0], mine.shape[1]), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return unpack_frame_blocks(out, cap)


def _contig(a):
    return a.contiguous() if hasattr(a, "contiguous") else __import__("numpy").ascontiguousarray(a)


def cross_rank_match(kps, desc, counts, all_kps, all_desc, all_counts, matcher, vocab=None, rank=None, world=None, levelsup=4):
    """The consumer of the descriptor all-gather (north_star: "RCCL all-gather of descriptors ... for cross-frame matching"): frame i of THIS rank
    against frame i of rank (rank + 1) % world, read out of the gathered slabs —
      * cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2) as Frame::ComputeStereoFishEyeMatches uses it (reference src/Frame.cc:1300; orbm_knn2), and
      * with a vocabulary, ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (src/ORBmatcher.cc:984-1124; LoopClosing / relocalisation match a
        key frame against candidates other ranks extracted).  Both sides are ComputeBoW'd here (Frame.cc:865-872): the vocabulary is replicated on every
        rank as in any ORB-SLAM3 process, only descriptors travel.
    Inputs: this rank's extractor outputs [F, cap, ...] and the gathered ones [world * F, cap, ...] (torch tensors, or numpy with the emulated build).
    -> dict(knn_idx [F,cap,2], knn_dist [F,cap,2][, bow_m12 [F,cap], bow_nmatches [F]])"""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    F = kps.shape[0]
    peer = (rank + 1) % world
    t_kps, t_desc, t_cnt = (_contig(a[peer * F:(peer + 1) * F]) for a in (all_kps, all_desc, all_counts))
    nq, nt = _contig(counts[:, 0]), _contig(t_cnt[:, 0])
    idx, dst = matcher.knnMatch2(_contig(desc), nq, t_desc, nt)
    res = dict(knn_idx=idx, knn_dist=dst)
    if vocab is not None:
        ra, rb = vocab.transform(_contig(desc), nq, levelsup), vocab.transform(t_desc, nt, levelsup)
        side = lambda r, d_, k_: dict(desc=_contig(d_), angle=_contig(k_[..., 3]), node_id=r["fv_node_id"], node_start=r["fv_node_start"],
                                      feat_idx=r["fv_feat_idx"], n_nodes=r["fv_n_nodes"])
        ones = lambda d_: (d_[..., 0] * 0 + 1) if hasattr(d_, "contiguous") else __import__("numpy").ones(d_.shape[:2], "u1")
        m12, nm = matcher.SearchByBoWKF(side(ra, desc, kps), _contig(ones(desc)), side(rb, t_desc, t_kps), _contig(ones(t_desc)))
        res.update(bow_m12=m12, bow_nmatches=nm)
    return res


def allgather_pose_blocks(poses, group=None):
    """LBA sharded by landmark: after a solve the updated pose blocks (n_p x 7 f64 = 5.6 kB at 100 KFs) of the
    poses each rank updated are all-gathered — the collective BASELINE.json configs[4] names.  poses: [n_local, 7] f64."""
    world = dist.get_world_size(group)
    out = torch.empty((world * poses.shape[0], poses.shape[1]), dtype=poses.dtype, device=poses.device)
    dist.all_gather_into_tensor(out, poses.contiguous(), group=group)
    return out


def shard_window_by_landmark(w, llo, lhi):
    """The part of a flattened LocalBundleAdjustment window (orbhip.lba: dict(poses, pose_hidx, points, edges)) a rank holds when the window is
    sharded by landmark: ALL poses, the points [llo, lhi) re-indexed from 0 and their edges (landmark-major order is kept).  For
    LbaWindows.optimize_sharded / the landmark-sharded linearisation."""
    e = w["edges"]
    el = e[(e["point"] >= llo) & (e["point"] < lhi)].copy()
    el["point"] -= llo
    return dict(w, points=w["points"][llo:lhi].copy(), edges=el)


def allreduce_pose_system(Hpp, bp, group=None):
    """LBA sharded by landmark: every rank holds the H_pp / b_p partial sums of ITS landmarks' edges; the pose-side system is
    their sum (one all-reduce of n_p*(36+6) doubles = 27 kB at 80 free KFs — latency-bound, reported honestly)."""
    buf = torch.cat([Hpp.reshape(-1), bp.reshape(-1)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    n = Hpp.numel()
    return buf[:n].reshape(Hpp.shape), buf[n:].reshape(bp.shape)


class _DevSlab:
    """A hipMalloc'ed buffer of its own (an IPC handle names a whole allocation: a tensor carved out of torch's caching allocator would export
    its neighbours too), visible to torch through __cuda_array_interface__."""

    def __init__(self, hip, device, shape, typestr, itemsize):
        import ctypes as C
        self._hip, self.shape = hip, tuple(shape)
        n = itemsize
        for s in shape:
            n *= s
        p = C.c_void_p()
        if hip.orb_dev_alloc(device, n, C.byref(p)) != 0:
            raise MemoryError("orb_dev_alloc(%d bytes)" % n)
        self.ptr, self.nbytes = p.value, n
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": typestr, "data": (self.ptr, False), "version": 2}

    def free(self):
        if self.ptr:
            import ctypes as C
            self._hip.orb_dev_free(C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        # torch keeps the object a __cuda_array_interface__ tensor was built from alive for as long as the tensor's storage lives (views included):
        # the memory goes when the LAST tensor on it goes, never under one
        try:
            self.free()
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass


class PeerExchange:
    """allgather_frame_blocks without RCCL (include/orbd.h orbd_allgather_frames_peer): every rank pulls each peer's three slabs straight out of
    the peer's memory, one device-to-device copy per peer and slab on a stream per peer — on xGMI one slab per point-to-point link, all seven at
    once.  One process per GPU: the slabs are allocated here (`kps`, `desc`, `counts`: hand them to ORBextractor.extract_batch(out=...)), their
    IPC handles are exchanged once through the process group, and `allgather()` fills `all_kps / all_desc / all_counts` (rank-major).
    The caller orders the ranks around `allgather()` (peers' slabs complete before, not rewritten until every rank is done): a
    torch.cuda.synchronize() + dist.barrier() on both sides, as for any one-sided read.

    Lifetime: use it as a context manager (or call close(), a COLLECTIVE: every rank closes its mappings of the peers' slabs, then a barrier, then
    this rank lets go of its own slabs).  A slab's memory is released when the last tensor on it — `kps` / `desc` / `counts` or any view a caller
    still holds — is gone, never under a live tensor."""

    def __init__(self, frames_per_rank, cap, device, group=None):
        import ctypes as C
        import os
        from . import _lib
        self.C = C
        self.group, self.world, self.rank = group, dist.get_world_size(group), dist.get_rank(group)
        self.F, self.cap, self.device = int(frames_per_rank), int(cap), device
        hip = _lib.load()
        hip.orb_dev_alloc.restype = C.c_int; hip.orb_dev_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
        hip.orb_dev_free.restype = C.c_int; hip.orb_dev_free.argtypes = [C.c_void_p]
        self.L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liborbd.so"))
        self.L.orbd_ipc_export.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orbd_ipc_open.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        self.L.orbd_ipc_close.argtypes = [C.c_void_p]
        self.L.orbd_allgather_frames_peer.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7
        di = device.index if hasattr(device, "index") else int(device)
        F, W = self.F, self.world
        self._slabs, self._opened, self._closed = [], [], False
        try:
            for shape, ts, isz in (((F, cap, 7), "<f4", 4), ((F, cap, 32), "|u1", 1), ((F, 2), "<i4", 4)):
                self._slabs.append(_DevSlab(hip, di, shape, ts, isz))
            dev = torch.device("cuda", di)
            self.kps, self.desc, self.counts = (torch.as_tensor(s_, device=dev) for s_ in self._slabs)
            self.all_kps = torch.empty((W * F, cap, 7), dtype=torch.float32, device=dev)
            self.all_desc = torch.empty((W * F, cap, 32), dtype=torch.uint8, device=dev)
            self.all_counts = torch.empty((W * F, 2), dtype=torch.int32, device=dev)
            mine, err = [], None
            for s_ in self._slabs:
                h = (C.c_uint8 * 64)()
                if self.L.orbd_ipc_export(C.c_void_p(s_.ptr), h) != 0:
                    err = "orbd_ipc_export failed (hipIpcGetMemHandle): is HSA_ENABLE_IPC_MODE_LEGACY=0 set?"
                mine.append(bytes(h))
            allh = [None] * W
            dist.all_gather_object(allh, (mine, err), group=group)        # every rank reaches the collective, also the one whose export failed
            bad = [(r, e) for r, (_, e) in enumerate(allh) if e]
            if bad:
                raise RuntimeError("rank %d: %s" % bad[0])
            self._peer = [(C.c_void_p * W)() for _ in range(3)]
            open_err = None
            for r in range(W):
                for k in range(3):
                    if r == self.rank:
                        self._peer[k][r] = self._slabs[k].ptr
                    elif open_err is None:
                        p = C.c_void_p()
                        hb = (C.c_uint8 * 64).from_buffer_copy(allh[r][0][k])
                        if self.L.orbd_ipc_open(hb, C.byref(p)) != 0:
                            open_err = "orbd_ipc_open failed for the slabs of rank %d" % r
                        else:
                            self._opened.append(p.value)
                            self._peer[k][r] = p.value
            # the opens are agreed on collectively as well: a rank whose open failed would otherwise drop its own slabs while the peers that did map
            # them carry on and read freed memory.  Every rank raises (and releases) together, after all of them have closed what they had opened.
            oks = [None] * W
            dist.all_gather_object(oks, open_err, group=group)
            bad = [(r, e) for r, e in enumerate(oks) if e]
            if bad:
                for p_ in self._opened:
                    self.L.orbd_ipc_close(C.c_void_p(p_))
                self._opened = []
                dist.barrier(group=group)                 # nobody still maps a peer's slab when the owners let go below
                raise RuntimeError("rank %d: %s" % bad[0])
        except Exception:
            self._release()           # what this rank had allocated / opened so far (no barrier: the peers are not known to be in step)
            raise

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close(barrier=exc[0] is None)
        return False

    def __del__(self):
        try:
            self._release()
        except Exception:   # noqa: BLE001
            pass

    def _release(self):
        for p in self._opened:
            self.L.orbd_ipc_close(self.C.c_void_p(p))
        self._opened = []
        self.kps = self.desc = self.counts = None
        self._slabs = []              # each slab frees itself when the last tensor on it is gone (_DevSlab.__del__)
        self._peer = None             # (closed mappings and freed slabs: nothing may hand these to the native call again)
        self._closed = True

    def allgather(self, stream=None):
        if self._closed:
            raise RuntimeError("PeerExchange is closed")
        C = self.C
        st = stream if stream is not None else torch.cuda.current_stream(self.kps.device).cuda_stream
        rc = self.L.orbd_allgather_frames_peer(self.world, self.rank, self.F, self.cap, self._peer[0], self._peer[1], self._peer[2],
                                               C.c_void_p(self.all_kps.data_ptr()), C.c_void_p(self.all_desc.data_ptr()),
                                               C.c_void_p(self.all_counts.data_ptr()), C.c_void_p(st))
        if rc != 0:
            raise RuntimeError("orbd_allgather_frames_peer: %d" % rc)
        return self.all_kps, self.all_desc, self.all_counts

    def close(self, barrier=True):
        """Collective: every rank unmaps its peers' slabs, all ranks meet, then this rank gives up its own (a peer must not still have a mapping of
        memory its owner releases).  barrier=False only on an error path where the peers are not known to be in step."""
        if self._closed:
            return
        torch.cuda.synchronize(self.all_kps.device)
        for p in self._opened:
            self.L.orbd_ipc_close(self.C.c_void_p(p))
        self._opened = []
        if barrier and dist.is_initialized():
            dist.barrier(group=self.group)
        self._release()
