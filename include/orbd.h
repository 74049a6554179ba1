/* orbd.h — the two exchange steps of the hot path for a C / C++ host that runs one process (or thread) per GPU,
 * as thin C entry points over RCCL (liborbd.so; liborbhip.so itself needs no peer and does not link RCCL).
 *
 * Frames, frame pairs and LBA windows are independent units: a rank runs orbx_ / orbm_ / lba_ on its own units with no collective.
 * Data crosses ranks in two places only:
 *   - matching against frames another rank extracted: the fixed-capacity per-frame slabs orbx_extract_batch_dev wrote
 *     ([cap] keypoint records, [cap] 32-byte descriptors, (n, monoIndex) per frame) are all-gathered — three all-gathers in ONE RCCL group
 *     launch, no packing copy (the Python mirror orbhip.dist.allgather_frame_blocks packs into one tensor because torch.distributed has no group call);
 *   - ONE LocalBundleAdjustment window sharded by landmark over several GPUs (BASELINE.json configs[4]): all-reduce of the pose-side
 *     partial sums H_pp / b_p that lba_build_system produced from each rank's landmarks, all-gather of the pose blocks each rank updated.
 * `comm` is an ncclComm_t the host created (ncclCommInitRank over its launcher's rendezvous, or orbd_comm_init_all_local for a single
 * process driving several GPUs).  All calls are asynchronous on `stream`; return ORB_OK / ORB_E_INVALID / ORB_E_HIP (an RCCL error). */
#ifndef ORBD_H
#define ORBD_H
#include <stddef.h>
#include <stdint.h>

#include "orbhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void* orbd_comm;   /* ncclComm_t */

/* Every rank contributes frames_per_rank frames (pad the last shard); afterwards every rank holds all world * frames_per_rank frames in rank-major
 * order.  d_kps [frames][cap] orb_keypoint, d_desc [frames][cap][32] u8, d_counts [frames][2] i32 (n, monoIndex) — the layouts of
 * orbx_extract_batch_dev; the d_all_* slabs are world times as large. */
int orbd_allgather_frames(orbd_comm comm, int world, int frames_per_rank, int cap, const orb_keypoint* d_kps, const uint8_t* d_desc,
                          const int32_t* d_counts, orb_keypoint* d_all_kps, uint8_t* d_all_desc, int32_t* d_all_counts, void* stream);

/* In-place sum over the ranks of the pose-side blocks of one landmark-sharded window: d_Hpp [n_free][36], d_bp [n_free][6] doubles
 * (lba_system.Hpp / .bp of the window, rows of fixed poses are zero on every rank); one RCCL group launch. */
int orbd_allreduce_pose_system(orbd_comm comm, double* d_Hpp, double* d_bp, int n_free, void* stream);

/* d_local [poses_per_rank][7] doubles (the pose blocks this rank updated) -> d_all [world * poses_per_rank][7], rank-major. */
int orbd_allgather_pose_blocks(orbd_comm comm, int world, const double* d_local, double* d_all, int poses_per_rank, void* stream);

/* The same all-gather WITHOUT RCCL (xGMI is point-to-point, 7 links per GPU — an all-gather of equal shards is one
 * slab per link): every rank PULLS each peer's three slabs straight out of the peer's memory with one device-to-device copy per peer, each on
 * its own stream, so the 7 incoming slabs travel on the 7 links at once; no ring, no staging buffer, no reduction kernel.
 *   orbd_ipc_export / orbd_ipc_open / orbd_ipc_close: one process per GPU — a rank exports an IPC handle (64 bytes) of each slab once, the handles
 *     are exchanged out of band (the launcher's rendezvous; orbhip.dist.PeerExchange uses torch.distributed.all_gather_object), every rank opens
 *     its peers' handles once and keeps the pointers.  (A single process driving several GPUs passes the peers' device pointers directly.)
 *   orbd_allgather_frames_peer: peer_* [world] = the slabs' device pointers as seen from THIS rank (entry [rank] = its own slabs).  Copies peer
 *     s's slabs into block s of the d_all_* slabs, peers taken in the order rank+1, rank+2, ... so that no two ranks start on the same source.
 *     The copies of different peers run on internal per-peer streams that start after `stream`'s work and are joined back into `stream`.
 *     The CALLER orders the ranks: the peers' slabs must be complete (their producer streams synchronised) before any rank calls this, and must
 *     not be rewritten until every rank's copies are done — a barrier of the launcher on both sides, as for any one-sided read. */
int orbd_ipc_export(const void* d_ptr, uint8_t handle_out[64]);
int orbd_ipc_open(const uint8_t handle[64], void** d_ptr_out);
int orbd_ipc_close(void* d_ptr);
int orbd_allgather_frames_peer(int world, int rank, int frames_per_rank, int cap, const void* const* peer_kps, const void* const* peer_desc,
                               const void* const* peer_counts, orb_keypoint* d_all_kps, uint8_t* d_all_desc, int32_t* d_all_counts, void* stream);
/* orbd_peer_enable_access: the same-process form (one host thread per GPU, raw device pointers instead of IPC mappings) needs peer access from the
 *     calling thread's current device to every other device of `devices`: enabled where missing, ORB_E_HIP if a pair cannot be connected.
 * orbd_peer_shutdown: drains and releases the calling thread's per-peer copy streams and events (they are per thread, made on first use, and NOT
 *     released by any destructor: at process exit the HIP runtime may already be gone).  Optional; after it the next all-gather makes them again.
 * If orbd_allgather_frames_peer fails part-way, the copies it had queued are still joined into `stream`: synchronise `stream` before reuse. */
int orbd_peer_enable_access(int n_devices, const int* devices);
int orbd_peer_shutdown(void);

/* Convenience for a single process that drives n_devices GPUs (and for the 1-GPU test): ncclCommInitAll.  comms[i] belongs to devices[i]. */
int orbd_comm_init_all_local(int n_devices, const int* devices, orbd_comm* comms);
int orbd_comm_destroy(orbd_comm comm);

#ifdef __cplusplus
}
#endif
#endif
