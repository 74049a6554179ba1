// Micro-benchmark (experiment, not product): throughput of fp64 atomic adds on MI355X, the scatter step of a landmark-major Schur complement.
//   mode 0: global_atomic_add_f64, each wave-instruction adds 36 contiguous doubles (one 6x6 block) of a block chosen pseudo-randomly
//           inside a per-XCD region of `regionKB`;   mode 1: ds_add_f64 into a 128 KB LDS array, lanes pick random blocks (36 adds per lane);
//   mode 2: global, lane-per-block (each lane adds to its own random block, 36 instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ unsigned rng(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

__global__ __launch_bounds__(256) void k_glob36(double* S, int nblk, int iters, int wgPerRegion) {
    const int lane = threadIdx.x & 63;
    const int region = (blockIdx.x % 8) + 8 * ((blockIdx.x / 8) / wgPerRegion);   // WGs of one XCD share a region for wgPerRegion WGs in a row
    double* R = S + (size_t)region * nblk * 36;
    unsigned s = blockIdx.x * 977u + (threadIdx.x >> 6) * 131u + 7u;
    for (int i = 0; i < iters; i++) {
        const unsigned b = rng(s) % (unsigned)nblk;
        if (lane < 36) atomicAdd(R + (size_t)b * 36 + lane, 1.0);
    }
}
__global__ __launch_bounds__(256) void k_glob_lane(double* S, int nblk, int iters, int wgPerRegion) {
    const int region = (blockIdx.x % 8) + 8 * ((blockIdx.x / 8) / wgPerRegion);
    double* R = S + (size_t)region * nblk * 36;
    unsigned s = blockIdx.x * 977u + threadIdx.x * 131u + 7u;
    for (int i = 0; i < iters; i++) {
        const unsigned b = rng(s) % (unsigned)nblk;
#pragma unroll
        for (int k = 0; k < 36; k++) atomicAdd(R + (size_t)b * 36 + k, 1.0);
    }
}
__global__ __launch_bounds__(256) void k_lds(double* out, int nblk, int iters, int activeOf = 1) {
    extern __shared__ double L[];
    for (int i = threadIdx.x; i < nblk * 37; i += 256) L[i] = 0;
    __syncthreads();
    unsigned s = blockIdx.x * 977u + threadIdx.x * 131u + 7u;
    for (int i = 0; i < iters; i++) {
        const unsigned rr = rng(s);
        const unsigned b = rr % (unsigned)nblk;
        if ((rr >> 12) % (unsigned)activeOf == 0) {      // 1 of `activeOf` lanes takes part (divergent, like a filtered partner loop)
#pragma unroll
            for (int k = 0; k < 36; k++) atomicAdd(&L[b * 37 + k], 1.0);
        }
    }
    __syncthreads();
    double t = 0;
    for (int i = threadIdx.x; i < nblk * 37; i += 256) t += L[i];
    if (t == -1.0) out[0] = t;
}
int main(int argc, char** argv) {
    const int nblk = 3240;                      // one 80-pose window's lower triangle: 933 KB
    double* S; hipMalloc(&S, (size_t)4096 * nblk * 36 * 8);
    hipMemset(S, 0, (size_t)4096 * nblk * 36 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    for (int wgPerRegion : {16, 128, 1 << 20}) {
        const int grid = 8192, iters = 2000;
        hipLaunchKernelGGL(k_glob36, dim3(grid), dim3(256), 0, 0, S, nblk, 10, wgPerRegion);
        hipEventRecord(a); hipLaunchKernelGGL(k_glob36, dim3(grid), dim3(256), 0, 0, S, nblk, iters, wgPerRegion); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("global 36-lane block adds, %7d WGs/region: %.3f ms -> %.2f G block-adds/s (%.1f G lane-atomics/s)\n", wgPerRegion, ms, grid * 4.0 * iters / ms / 1e6, grid * 4.0 * iters * 36 / ms / 1e6);
    }
    for (int wgPerRegion : {16, 128}) {
        const int grid = 8192, iters = 100;
        hipEventRecord(a); hipLaunchKernelGGL(k_glob_lane, dim3(grid), dim3(256), 0, 0, S, nblk, iters, wgPerRegion); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("global lane-per-block adds, %7d WGs/region: %.3f ms -> %.2f G block-adds/s\n", wgPerRegion, ms, grid * 256.0 * iters / ms / 1e6);
    }
    for (int act : {2, 4, 8}) {
        const int grid = 2048, iters = 500, nb = 41;
        hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 400 * 37 * 8);
        hipEventRecord(a); hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 400 * 37 * 8, 0, S, nb, iters, act); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("LDS lane-per-block adds into %d blocks, 1 of %d lanes active: %.3f ms -> %.2f G wave-instructions/s, %.1f G lane-atomics/s\n", nb, act, ms, grid * 4.0 * iters * 36 / ms / 1e6, grid * 256.0 * iters * 36 / act / ms / 1e6);
    }
    for (int nb : {400, 80}) {
        const int grid = 2048, iters = 500;
        hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 400 * 37 * 8);
        hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 400 * 37 * 8, 0, S, nb, 5, 1);
        hipEventRecord(a); hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 400 * 37 * 8, 0, S, nb, iters, 1); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("LDS lane-per-block adds into %d blocks: %.3f ms -> %.2f G block-adds/s (%.1f G lane-atomics/s)\n", nb, ms, grid * 256.0 * iters / ms / 1e6, grid * 256.0 * iters * 36 / ms / 1e6);
    }
    return 0;
}
