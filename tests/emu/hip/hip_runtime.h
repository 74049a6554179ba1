// TEST INFRASTRUCTURE ONLY — a fiber-based emulator of the small slice of the HIP
// programming model the product kernels use, so that the *unmodified* kernel sources under
// awesome-orb-slam3-3dvisioncraft-version_amd/csrc/ can be compiled with g++ and their LOGIC checked
// against the oracle in the CPU-only (`-m "not gpu"`) test tier.  This file shadows <hip/hip_runtime.h>
// only when tests/emu is put on the include path by tests/emu/build_emu.py.  The product never loads
// the emulated library; the shipped path is the hipcc-built liborbhip.so and nothing else.
//
// Model: one workgroup at a time PER HOST THREAD (all scheduler state, the built-in index variables and the LDS image are thread_local: host
// threads may launch concurrently, as Tracking / LocalMapping / LoopClosing do — tests/cpp/threads_test.cpp); every work-item is a ucontext fiber; __syncthreads() and the wave
// intrinsics (__ballot, __shfl*) are barriers among the fibers of the block / of one 64-lane wave.
// Fibers are run in ascending or (EMU_REVERSE=1) descending lane order between barriers, which makes
// most missing-barrier races show up as wrong answers in one of the two orders.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <tuple>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ thread_local
#define __constant__
#define HIP_EMULATED 1

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct short2 { short x, y; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct ushort2 { unsigned short x, y; };
struct uchar4 { unsigned char x, y, z, w; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
static inline double2 make_double2(double a, double b) { return double2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline int2 make_int2(int a, int b) { return int2{a, b}; }
static inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }
static inline short2 make_short2(short a, short b) { return short2{a, b}; }

typedef int hipError_t;
typedef struct emu_stream_t* hipStream_t;
typedef struct emu_event_t { double t; }* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotSupported = 801 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };

#if defined(__SANITIZE_THREAD__)
// ThreadSanitizer cannot follow swapcontext on its own: every fiber is announced through its fiber API (a switch with flags = 0 also
// synchronises the two fibers, so the lanes of one launch — one host thread's sequential schedule — never race with each other; races BETWEEN
// host threads are what the TSAN build of tests/cpp/threads_test.cpp is for)
extern "C" {
void* __tsan_get_current_fiber(void);
void* __tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void* fiber);
void __tsan_switch_to_fiber(void* fiber, unsigned flags);
}
#define EMU_TSAN 1
#endif

namespace emu {
enum { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
// one work-item slot of a host thread: created on first use, then reused by every block that thread runs (its trampoline loops over launches, so
// its stack — and, under TSAN, its shadow call stack — is back at the same depth whenever a block starts)
struct Fiber {
    ucontext_t ctx;            // never moved once made (glibc's ucontext_t points into itself)
    char* stack = nullptr;
    void* tsan = nullptr;
};
struct State {
    ucontext_t sched;
    std::vector<Fiber*> fib;
    std::vector<int> st;
    std::function<void()> body;
    int cur = 0, nthreads = 0;
    void* tsan_main = nullptr;
    dim3 block;
    uint64_t wavebuf[16][64];
    ~State() {
        for (Fiber* f : fib) {
#ifdef EMU_TSAN
            if (f->tsan) __tsan_destroy_fiber(f->tsan);
#endif
            free(f->stack);
            delete f;
        }
    }
};
inline State& S() { static thread_local State s; return s; }
}  // namespace emu

inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
alignas(256) inline thread_local unsigned char orb_smem[160 * 1024];
static const int warpSize = 64;

namespace emu {
inline void set_tid(int t) {
    State& s = S();
    threadIdx.x = t % s.block.x;
    threadIdx.y = (t / s.block.x) % s.block.y;
    threadIdx.z = t / (s.block.x * s.block.y);
}
inline void to_sched(State& s) {
#ifdef EMU_TSAN
    __tsan_switch_to_fiber(s.tsan_main, 0);
#endif
    swapcontext(&s.fib[s.cur]->ctx, &s.sched);
}
inline void to_fiber(State& s, int i) {
#ifdef EMU_TSAN
    __tsan_switch_to_fiber(s.fib[i]->tsan, 0);
#endif
    swapcontext(&s.sched, &s.fib[i]->ctx);
}
inline void yield(int why) {
    State& s = S();
    s.st[s.cur] = why;
    to_sched(s);
}
inline void trampoline() {
    State& s = S();     // (a fiber only ever runs on the host thread that made it)
    for (;;) {          // one turn per block this slot takes part in
        s.body();
        s.st[s.cur] = DONE;
        to_sched(s);
    }
}
inline void run_block(dim3 block, const std::function<void()>& body) {
    State& s = S();
    const int n = (int)(block.x * block.y * block.z);
    const size_t STK = 128 * 1024;
    s.block = block;
    s.nthreads = n;
    s.body = body;
    s.st.assign(n, RUNNABLE);
#ifdef EMU_TSAN
    s.tsan_main = __tsan_get_current_fiber();
#endif
    while ((int)s.fib.size() < n) {
        Fiber* f = new Fiber;
        f->stack = (char*)malloc(STK);
        getcontext(&f->ctx);
        f->ctx.uc_stack.ss_sp = f->stack;
        f->ctx.uc_stack.ss_size = STK;
        f->ctx.uc_link = &s.sched;
        makecontext(&f->ctx, (void (*)())trampoline, 0);
#ifdef EMU_TSAN
        f->tsan = __tsan_create_fiber(0);
#endif
        s.fib.push_back(f);
    }
    static const bool reverse = getenv("EMU_REVERSE") && atoi(getenv("EMU_REVERSE"));
    const int nw = (n + 63) / 64;
    for (;;) {
        bool ran = false;
        for (int k = 0; k < n; k++) {
            int i = reverse ? n - 1 - k : k;
            if (s.st[i] != RUNNABLE) continue;
            s.cur = i;
            set_tid(i);
            to_fiber(s, i);
            ran = true;
        }
        bool released = false, alldone = true, allblock = true;
        for (int w = 0; w < nw; w++) {
            bool allw = true, any = false;
            for (int l = w * 64; l < std::min(n, w * 64 + 64); l++) {
                if (s.st[l] == DONE) continue;
                any = true;
                if (s.st[l] != WAIT_WAVE) allw = false;
            }
            if (any && allw) {
                for (int l = w * 64; l < std::min(n, w * 64 + 64); l++)
                    if (s.st[l] == WAIT_WAVE) s.st[l] = RUNNABLE;
                released = true;
            }
        }
        for (int i = 0; i < n; i++) {
            if (s.st[i] != DONE) alldone = false;
            if (s.st[i] != DONE && s.st[i] != WAIT_BLOCK) allblock = false;
        }
        if (alldone) break;
        if (!released && allblock) {
            for (int i = 0; i < n; i++)
                if (s.st[i] == WAIT_BLOCK) s.st[i] = RUNNABLE;
            released = true;
        }
        if (!released && !ran) {
            fprintf(stderr, "hip_emu: DEADLOCK (divergent barrier) in block (%u,%u,%u)\n", blockIdx.x, blockIdx.y, blockIdx.z);
            abort();
        }
    }
}
inline void launch(dim3 grid, dim3 block, size_t smemBytes, const std::function<void()>& body) {
    if (smemBytes > sizeof(orb_smem)) { fprintf(stderr, "hip_emu: smem %zu too large\n", smemBytes); abort(); }
    gridDim = grid;
    blockDim = block;
    for (unsigned z = 0; z < grid.z; z++)
        for (unsigned y = 0; y < grid.y; y++)
            for (unsigned x = 0; x < grid.x; x++) {
                blockIdx = dim3(x, y, z);
                memset(orb_smem, 0xCD, smemBytes);  // poison: LDS is uninitialised on hardware
                run_block(block, body);
            }
}
inline int lane() { return S().cur & 63; }
inline int wave() { return S().cur >> 6; }
inline uint64_t exchange(uint64_t v, int src) {
    State& s = S();
    const int w = wave();
    s.wavebuf[w][lane()] = v;
    yield(WAIT_WAVE);
    uint64_t r = s.wavebuf[w][src & 63];
    yield(WAIT_WAVE);
    return r;
}
}  // namespace emu

template <typename... KArgs, typename... Args>
inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t smemBytes, hipStream_t, Args... args) {
    std::tuple<KArgs...> targs(args...);
    emu::launch(grid, block, smemBytes, [&] { std::apply(kernel, targs); });
}

// ---- device intrinsics -----------------------------------------------------------------------------------
inline void __syncthreads() { emu::yield(emu::WAIT_BLOCK); }
inline void __builtin_amdgcn_wave_barrier() { emu::yield(emu::WAIT_WAVE); }  // lanes are not lock-step here
inline void __builtin_amdgcn_fence(int, const char*) {}
inline void __threadfence() {}
inline void __threadfence_block() {}
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; *p = std::max(o, v); return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; *p = std::min(o, v); return o; }
template <class T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <class T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __float2int_rn(float v) { return (int)lrintf(v); }
inline int __float2int_rz(float v) { return (int)v; }
inline float __int2float_rn(int v) { return (float)v; }
inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
inline unsigned long long __ballot(int pred) {
    emu::State& s = emu::S();
    const int w = emu::wave();
    s.wavebuf[w][emu::lane()] = pred ? 1 : 0;
    emu::yield(emu::WAIT_WAVE);
    unsigned long long m = 0;
    const int n = std::min(64, s.nthreads - w * 64);
    for (int l = 0; l < n; l++)
        if (s.st[w * 64 + l] != emu::DONE && s.wavebuf[w][l]) m |= 1ull << l;
    emu::yield(emu::WAIT_WAVE);
    return m;
}
inline int __any(int p) { return __ballot(p) != 0; }
inline int __all(int p) { return __ballot(!p) == 0; }
template <class T> inline T __shfl(T v, int src, int width = 64) {
    static_assert(sizeof(T) <= 8, "shfl");
    uint64_t b = 0;
    memcpy(&b, &v, sizeof(T));
    const int l = emu::lane();
    const int base = l & ~(width - 1);
    b = emu::exchange(b, base + (src & (width - 1)));
    T r;
    memcpy(&r, &b, sizeof(T));
    return r;
}
template <class T> inline T __shfl_xor(T v, int m, int width = 64) { return __shfl(v, (emu::lane() ^ m), 64); }
template <class T> inline T __shfl_down(T v, int d, int width = 64) {
    int l = emu::lane();
    return __shfl(v, (l + d < 64 ? l + d : l), 64);
}
template <class T> inline T __shfl_up(T v, int d, int width = 64) {
    int l = emu::lane();
    return __shfl(v, (l - d >= 0 ? l - d : l), 64);
}
// v_mfma_f64_16x16x4_f64: D = A B + C on one wave.  Lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15]; result register i = D[(l >> 4) + 4 i][l & 15].
typedef double emu_v4f64 __attribute__((vector_size(32)));
inline emu_v4f64 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, emu_v4f64 c, int, int, int) {
    const int l = emu::lane(), n = l & 15, rg = l >> 4;
    uint64_t ua, ub;
    memcpy(&ua, &a, 8); memcpy(&ub, &b, 8);
    double bk[4];
    for (int k = 0; k < 4; k++) { const uint64_t u = emu::exchange(ub, n + 16 * k); memcpy(&bk[k], &u, 8); }
    for (int i = 0; i < 4; i++) {
        const int row = rg + 4 * i;
        double acc = c[i];
        for (int k = 0; k < 4; k++) { const uint64_t u = emu::exchange(ua, row + 16 * k); double ak; memcpy(&ak, &u, 8); acc += ak * bk[k]; }
        c[i] = acc;
    }
    return c;
}
inline int __builtin_amdgcn_readlane(int v, int l) { return __shfl(v, l, 64); }   // lane index must be wave-uniform
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // the kernels use it on wave-uniform values only (to tell the compiler so)
// v_mov_b32_dpp as __builtin_amdgcn_update_dpp: the controls the kernels use — quad_perm (0x00-0xFF), row_shl:n (0x100 + n), row_shr:n (0x110 + n),
// row_bcast:15 (0x142), row_bcast:31 (0x143).
// A lane whose row / bank is masked out, or whose source lane lies outside its row, keeps `old` (bound_ctrl = false) or gets 0 (true).
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int l = emu::lane(), row = l >> 4, pos = l & 15;
    int from = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; if (pos - n >= 0) from = l - n; }
    else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl - 0x100; if (pos + n <= 15) from = l + n; }   // row_shl:n
    else if (ctrl >= 0 && ctrl <= 0xFF) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);                          // quad_perm
    else if (ctrl == 0x142) { if (row >= 1) from = row * 16 - 1; }
    else if (ctrl == 0x143) { if (row >= 2) from = 31; }
    else { fprintf(stderr, "hip_emu: unsupported dpp_ctrl 0x%x\n", ctrl); abort(); }
    const int got = (int)(unsigned)emu::exchange((uint64_t)(unsigned)src, from < 0 ? l : from);
    if (!((row_mask >> row) & 1) || !((bank_mask >> (pos >> 2)) & 1)) return old;
    if (from < 0) return bound_ctrl ? 0 : old;
    return got;
}
inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) {
    return (unsigned)(((((uint64_t)hi) << 32) | lo) >> (sh & 31));
}
inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) {
    return (unsigned)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3)));
}
inline unsigned __builtin_amdgcn_lerp(unsigned a, unsigned b, unsigned c) {   // v_lerp_u8: per byte (a + b + (c & 1)) >> 1
    unsigned r = 0;
    for (int i = 0; i < 4; i++) r |= ((((a >> (8 * i)) & 255u) + ((b >> (8 * i)) & 255u) + ((c >> (8 * i)) & 1u)) >> 1) << (8 * i);
    return r;
}
inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel) {   // v_perm_b32 (selectors 0-7 and 0x0c only)
    const uint64_t src = ((uint64_t)s0 << 32) | s1;
    unsigned r = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned c = (sel >> (8 * i)) & 255u;
        const unsigned b = c < 8 ? (unsigned)((src >> (8 * c)) & 255u) : (c == 0x0c ? 0u : 0xFFu);
        r |= b << (8 * i);
    }
    return r;
}
inline int __builtin_amdgcn_mad_i32_i24_emu(int a, int b, int c) { return a * b + c; }
typedef unsigned short emu_u16x2 __attribute__((vector_size(4)));
inline unsigned __builtin_amdgcn_udot2(emu_u16x2 a, emu_u16x2 b, unsigned c, bool) { return c + (unsigned)a[0] * b[0] + (unsigned)a[1] * b[1]; }
inline unsigned __builtin_amdgcn_udot4(unsigned a, unsigned b, unsigned c, bool) {
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 255u) * ((b >> (8 * i)) & 255u);
    return c;
}
using std::max;
using std::min;

// ---- host runtime -------------------------------------------------------------------------------------------
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }   // the host-side work split is sized as on an MI355X
inline hipError_t hipMalloc(void** p, size_t n) {
    n = (n + 255) & ~(size_t)255;
    *p = aligned_alloc(256, n ? n : 256);
    if (*p) memset(*p, 0xAB, n ? n : 256);  // poison: hipMalloc memory is uninitialised
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipMalloc((void**)p, n); }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = nullptr) {
    for (size_t y = 0; y < h; y++) memcpy((char*)d + y * dp, (const char*)s + y * sp, w);
    return hipSuccess;
}
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
typedef void* hipDeviceptr_t;
inline hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t count, hipStream_t = nullptr) { for (size_t i = 0; i < count; i++) ((int*)d)[i] = v; return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
// HIP graphs are not emulated: capture reports "not supported" and the product falls back to direct launches
typedef struct emu_graph_t* hipGraph_t;
typedef struct emu_graph_exec_t* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hip_emu"; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event_t{0}; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event_t{0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "hip_emu");
    strcpy(p->gcnArchName, "emu");
    p->multiProcessorCount = 1;
    return hipSuccess;
}
#define HIP_SYMBOL(x) x
template <class T> inline hipError_t hipMemcpyToSymbol(T& sym, const void* src, size_t n) { memcpy((void*)&sym, src, n); return hipSuccess; }
template <class T> inline hipError_t hipFuncSetAttribute(T, int, int) { return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
