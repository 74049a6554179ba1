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


def allgather_pose_blocks(poses, group=None):
    """LBA sharded by landmark: after a solve the updated pose blocks (n_p x 7 f64 = 5.6 kB at 100 KFs) of the
    poses each rank updated are all-gathered — the collective BASELINE.json configs[4] names.  poses: [n_local, 7] f64."""
    world = dist.get_world_size(group)
    out = torch.empty((world * poses.shape[0], poses.shape[1]), dtype=poses.dtype, device=poses.device)
    dist.all_gather_into_tensor(out, poses.contiguous(), group=group)
    return out


def allreduce_pose_system(Hpp, bp, group=None):
    """LBA sharded by landmark: every rank holds the H_pp / b_p partial sums of ITS landmarks' edges; the pose-side system is
    their sum (one all-reduce of n_p*(36+6) doubles = 27 kB at 80 free KFs — latency-bound, reported honestly)."""
    buf = torch.cat([Hpp.reshape(-1), bp.reshape(-1)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    n = Hpp.numel()
    return buf[:n].reshape(Hpp.shape), buf[n:].reshape(bp.shape)
