// orb_runtime.hip — thin device-memory helpers of the C ABI (include/orbhip.h), so that the C++ adapters in
// include/orbslam3_hip/ compile without HIP headers.
#include <hip/hip_runtime.h>

#include "../../include/orbhip.h"

static inline int rc(hipError_t e) { return e == hipSuccess ? ORB_OK : (e == hipErrorOutOfMemory ? ORB_E_NOMEM : ORB_E_HIP); }

extern "C" int orb_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
extern "C" int orb_dev_alloc(int device, size_t bytes, void** d_ptr) {
    if (!d_ptr) return ORB_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return ORB_E_HIP;
    return rc(hipMalloc(d_ptr, bytes ? bytes : 1));
}
extern "C" int orb_dev_free(void* d_ptr) { return d_ptr ? rc(hipFree(d_ptr)) : ORB_OK; }
extern "C" int orb_host_alloc(size_t bytes, void** h_ptr) { return h_ptr ? rc(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault)) : ORB_E_INVALID; }
extern "C" int orb_host_free(void* h_ptr) { return h_ptr ? rc(hipHostFree(h_ptr)) : ORB_OK; }
extern "C" int orb_memcpy_h2d(void* d, const void* h, size_t n, void* st) { return rc(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)st)); }
extern "C" int orb_memcpy_d2h(void* h, const void* d, size_t n, void* st) { return rc(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)st)); }
extern "C" int orb_memset(void* d, int v, size_t n, void* st) { return rc(hipMemsetAsync(d, v, n, (hipStream_t)st)); }
extern "C" int orb_stream_sync(void* st) { return rc(hipStreamSynchronize((hipStream_t)st)); }
