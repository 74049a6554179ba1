// orbd_exchange.cpp — liborbd.so: the exchange steps of the path over RCCL (include/orbd.h).  Host code only; the collectives run on the
// caller's stream next to the orbx_ / orbm_ / lba_ kernels of liborbhip.so.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "orbd.h"

static int rc_of(ncclResult_t r) { return r == ncclSuccess ? ORB_OK : ORB_E_HIP; }

extern "C" int orbd_allgather_frames(orbd_comm comm, int world, int frames_per_rank, int cap, const orb_keypoint* d_kps, const uint8_t* d_desc,
                                     const int32_t* d_counts, orb_keypoint* d_all_kps, uint8_t* d_all_desc, int32_t* d_all_counts, void* stream) {
    if (!comm || world < 1 || frames_per_rank < 1 || cap < 1 || !d_kps || !d_desc || !d_counts || !d_all_kps || !d_all_desc || !d_all_counts) return ORB_E_INVALID;
    ncclComm_t c = (ncclComm_t)comm;
    hipStream_t st = (hipStream_t)stream;
    const size_t F = (size_t)frames_per_rank;
    // one group launch: the three slabs travel together, nothing is packed
    ncclResult_t r = ncclGroupStart();
    if (r == ncclSuccess) r = ncclAllGather(d_kps, d_all_kps, F * cap * sizeof(orb_keypoint), ncclUint8, c, st);
    if (r == ncclSuccess) r = ncclAllGather(d_desc, d_all_desc, F * cap * 32, ncclUint8, c, st);
    if (r == ncclSuccess) r = ncclAllGather(d_counts, d_all_counts, F * 2, ncclInt32, c, st);
    const ncclResult_t e = ncclGroupEnd();
    return rc_of(r == ncclSuccess ? e : r);
}

extern "C" int orbd_allreduce_pose_system(orbd_comm comm, double* d_Hpp, double* d_bp, int n_free, void* stream) {
    if (!comm || !d_Hpp || !d_bp || n_free < 1) return ORB_E_INVALID;
    ncclComm_t c = (ncclComm_t)comm;
    hipStream_t st = (hipStream_t)stream;
    ncclResult_t r = ncclGroupStart();
    if (r == ncclSuccess) r = ncclAllReduce(d_Hpp, d_Hpp, (size_t)n_free * 36, ncclDouble, ncclSum, c, st);
    if (r == ncclSuccess) r = ncclAllReduce(d_bp, d_bp, (size_t)n_free * 6, ncclDouble, ncclSum, c, st);
    const ncclResult_t e = ncclGroupEnd();
    return rc_of(r == ncclSuccess ? e : r);
}

extern "C" int orbd_allgather_pose_blocks(orbd_comm comm, int world, const double* d_local, double* d_all, int poses_per_rank, void* stream) {
    if (!comm || world < 1 || !d_local || !d_all || poses_per_rank < 1) return ORB_E_INVALID;
    return rc_of(ncclAllGather(d_local, d_all, (size_t)poses_per_rank * 7, ncclDouble, (ncclComm_t)comm, (hipStream_t)stream));
}

extern "C" int orbd_comm_init_all_local(int n_devices, const int* devices, orbd_comm* comms) {
    if (n_devices < 1 || !devices || !comms) return ORB_E_INVALID;
    return rc_of(ncclCommInitAll((ncclComm_t*)comms, n_devices, devices));
}

extern "C" int orbd_comm_destroy(orbd_comm comm) {
    if (!comm) return ORB_E_INVALID;
    return rc_of(ncclCommDestroy((ncclComm_t)comm));
}
