// orbd_exchange.cpp — liborbd.so: the exchange steps of the path over RCCL (include/orbd.h).  Host code only; the collectives run on the
// caller's stream next to the orbx_ / orbm_ / lba_ kernels of liborbhip.so.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstring>
#include <vector>

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

// ---- the all-gather as explicit peer copies (no RCCL): one pull per peer and slab, the peers on streams of their own -------------------------
extern "C" int orbd_ipc_export(const void* d_ptr, uint8_t handle_out[64]) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the exchanged handle is 64 bytes");
    if (!d_ptr || !handle_out) return ORB_E_INVALID;
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, const_cast<void*>(d_ptr)) != hipSuccess) { (void)hipGetLastError(); return ORB_E_HIP; }
    std::memcpy(handle_out, &h, 64);
    return ORB_OK;
}
extern "C" int orbd_ipc_open(const uint8_t handle[64], void** d_ptr_out) {
    if (!handle || !d_ptr_out) return ORB_E_INVALID;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, 64);
    if (hipIpcOpenMemHandle(d_ptr_out, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return ORB_E_HIP; }
    return ORB_OK;
}
extern "C" int orbd_ipc_close(void* d_ptr) {
    if (!d_ptr) return ORB_E_INVALID;
    return hipIpcCloseMemHandle(d_ptr) == hipSuccess ? ORB_OK : ORB_E_HIP;
}

namespace {
// per calling thread (= per rank in a one-process-per-GPU host, per device thread otherwise): the copy streams and their events, made once.
// Nothing here is destroyed by a destructor: a thread_local's destructor runs at thread / process exit, possibly after the HIP runtime has been
// torn down (hipStreamDestroy on a dead runtime is undefined).  A host that wants the streams back calls orbd_peer_shutdown() on the thread
// that used them; a process that simply exits leaves them to the runtime's own teardown.
struct PeerStreams {
    int device = -1;
    std::vector<hipStream_t> st;
    std::vector<hipEvent_t> done;
    hipEvent_t start = nullptr;
    // copies may still be in flight on the peer streams (a failed call, or a device switch between calls): drain before destroying
    void drop() {
        if (device >= 0) {
            int cur = -1;
            const bool sw = hipGetDevice(&cur) == hipSuccess && cur != device && hipSetDevice(device) == hipSuccess;
            for (auto s : st) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
            for (auto e : done) (void)hipEventDestroy(e);
            if (start) (void)hipEventDestroy(start);
            if (sw) (void)hipSetDevice(cur);
        }
        st.clear(); done.clear(); start = nullptr; device = -1;
    }
    bool ensure(int n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (device != dev) drop();
        device = dev;
        if (!start && hipEventCreateWithFlags(&start, hipEventDisableTiming) != hipSuccess) return false;
        while ((int)st.size() < n) {
            hipStream_t s; hipEvent_t e;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(s); return false; }
            st.push_back(s); done.push_back(e);
        }
        return true;
    }
};
thread_local PeerStreams g_peer;
}  // namespace

// Releases the calling thread's copy streams and events (after draining them).  Optional: see PeerStreams.
extern "C" int orbd_peer_shutdown(void) {
    g_peer.drop();
    return ORB_OK;
}

// Same-process form (one host thread per GPU, raw device pointers of the other GPUs instead of IPC mappings): the copies need peer access from the
// calling thread's device to every source device.  Enables it where it is missing; ORB_E_HIP if a pair cannot be connected.
extern "C" int orbd_peer_enable_access(int n_devices, const int* devices) {
    if (n_devices < 1 || !devices) return ORB_E_INVALID;
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return ORB_E_HIP;
    for (int i = 0; i < n_devices; i++) {
        if (devices[i] == cur) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, cur, devices[i]) != hipSuccess || !can) { (void)hipGetLastError(); return ORB_E_HIP; }
        const hipError_t e = hipDeviceEnablePeerAccess(devices[i], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return ORB_E_HIP; }
        (void)hipGetLastError();
    }
    return ORB_OK;
}

extern "C" int orbd_allgather_frames_peer(int world, int rank, int frames_per_rank, int cap, const void* const* peer_kps, const void* const* peer_desc,
                                          const void* const* peer_counts, orb_keypoint* d_all_kps, uint8_t* d_all_desc, int32_t* d_all_counts,
                                          void* stream) {
    if (world < 1 || rank < 0 || rank >= world || frames_per_rank < 1 || cap < 1 || !peer_kps || !peer_desc || !peer_counts || !d_all_kps ||
        !d_all_desc || !d_all_counts)
        return ORB_E_INVALID;
    for (int s = 0; s < world; s++)
        if (!peer_kps[s] || !peer_desc[s] || !peer_counts[s]) return ORB_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (!g_peer.ensure(world)) return ORB_E_HIP;
    const size_t F = (size_t)frames_per_rank;
    const size_t bk = F * cap * sizeof(orb_keypoint), bd = F * cap * 32, bc = F * 2 * sizeof(int32_t);
    if (hipEventRecord(g_peer.start, st) != hipSuccess) return ORB_E_HIP;
    bool ok = true;
    for (int i = 0; i < world && ok; i++) {
        const int s = (rank + 1 + i) % world;           // own block last; no two ranks start on the same source
        hipStream_t cs = s == rank ? st : g_peer.st[s];
        if (s != rank) ok = hipStreamWaitEvent(cs, g_peer.start, 0) == hipSuccess;
        ok = ok && hipMemcpyAsync((uint8_t*)d_all_kps + (size_t)s * bk, peer_kps[s], bk, hipMemcpyDeviceToDevice, cs) == hipSuccess;
        ok = ok && hipMemcpyAsync(d_all_desc + (size_t)s * bd, peer_desc[s], bd, hipMemcpyDeviceToDevice, cs) == hipSuccess;
        ok = ok && hipMemcpyAsync((uint8_t*)d_all_counts + (size_t)s * bc, peer_counts[s], bc, hipMemcpyDeviceToDevice, cs) == hipSuccess;
        // joined back into `stream` whether or not every copy of this peer was queued: what WAS queued must not outlive the call unobserved
        if (s != rank) {
            const bool joined = hipEventRecord(g_peer.done[s], cs) == hipSuccess && hipStreamWaitEvent(st, g_peer.done[s], 0) == hipSuccess;
            ok = ok && joined;
        }
    }
    if (!ok) (void)hipGetLastError();
    return ok ? ORB_OK : ORB_E_HIP;
}

extern "C" int orbd_comm_init_all_local(int n_devices, const int* devices, orbd_comm* comms) {
    if (n_devices < 1 || !devices || !comms) return ORB_E_INVALID;
    return rc_of(ncclCommInitAll((ncclComm_t*)comms, n_devices, devices));
}

extern "C" int orbd_comm_destroy(orbd_comm comm) {
    if (!comm) return ORB_E_INVALID;
    return rc_of(ncclCommDestroy((ncclComm_t)comm));
}
