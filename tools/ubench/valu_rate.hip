// Micro-benchmark (measurement, not product): issue rate of the gfx950 instructions the extractor / matcher kernels are made of.
//
// DESIGN.md's "the kernels sit on the VALU-issue roofline" needs the roofline: SIMD cycles per wave64 instruction, per instruction form,
// measured — MI355X_MICROARCH.md says CDNA4 SIMDs issue a wave64 `v_fma_f32` over 2 cycles, rounds 1-3 of this tree assumed 4.
//
// Method.  Every wave runs `iters` x 64 inline-asm instances of ONE instruction between two `s_memtime` reads (shader-clock ticks), either as
// 8 independent chains (8 accumulators: throughput) or as 1 dependent chain (the same accumulator: latency when a wave is alone, throughput
// again once enough waves share the SIMD).  A wave also stores HW_REG_HW_ID / HW_REG_XCC_ID, so the host groups waves by the SIMD they
// actually ran on: for a SIMD with w resident waves,
//     cycles per instruction (SIMD) = (max end tick - min start tick over its waves) / (w x instructions per wave)
// The launch is sized so that every SIMD of the chip holds `w` waves (256 CUs x 4 SIMDs x w, workgroups of 256 threads = one wave per SIMD);
// SIMDs that ended up with another wave count are reported separately and left out of the figure.  Output: CSV on stdout.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate tools/ubench/valu_rate.hip && ./valu_rate > profiles/valu_rates.csv
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Rec { uint64_t t0, t1; uint32_t hwid, xcc; };

__device__ __forceinline__ uint32_t hw_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v; }

// ---- instruction bodies: op(acc, b, c) issues ONE instruction that reads and writes `acc` ---------------------------------------------
#define OP32(NAME, ASM)                                                                                    \
    struct NAME { using T = uint32_t; static constexpr const char* name = #NAME;                           \
        static __device__ __forceinline__ void op(uint32_t& a, uint32_t b, uint32_t c) {                   \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); } };
#define OP32C(NAME, ASM)   /* bodies that write vcc / s20 / s21: the clobber list makes the compiler put an s_nop behind every statement — w = 1 figures of these rows include it */ \
    struct NAME { using T = uint32_t; static constexpr const char* name = #NAME;                           \
        static __device__ __forceinline__ void op(uint32_t& a, uint32_t b, uint32_t c) {                   \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c) : "vcc", "s20", "s21"); } };
#define OP64(NAME, ASM)                                                                                    \
    struct NAME { using T = double; static constexpr const char* name = #NAME;                             \
        static __device__ __forceinline__ void op(double& a, double b, double c) {                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); } };

OP32(v_add_u32, "v_add_u32 %0, %0, %1")
OP32(v_and_b32, "v_and_b32 %0, %0, %1")
OP32(v_xor_b32, "v_xor_b32 %0, %0, %1")
OP32(v_lshlrev_b32, "v_lshlrev_b32 %0, 1, %0")
OP32(v_lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %1")
OP32(v_add3_u32, "v_add3_u32 %0, %0, %1, %2")
OP32(v_and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
OP32(v_bfe_u32, "v_bfe_u32 %0, %0, 3, 8")
OP32(v_min_u32, "v_min_u32 %0, %0, %1")
OP32(v_max3_u32, "v_max3_u32 %0, %0, %1, %2")
OP32(v_alignbyte_b32, "v_alignbyte_b32 %0, %0, %1, 1")
OP32(v_lerp_u8, "v_lerp_u8 %0, %0, %1, %2")
OP32(v_sad_u8, "v_sad_u8 %0, %1, %2, %0")
OP32(v_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
OP32(v_pk_max_u16, "v_pk_max_u16 %0, %0, %1")
OP32(v_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
OP32(v_pk_sub_u16, "v_pk_sub_u16 %0, %0, %1")
OP32(v_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
OP32(v_dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
OP32(v_dot2_u32_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
OP32(v_perm_b32, "v_perm_b32 %0, %0, %1, %2")
OP32(v_bcnt_u32_b32, "v_bcnt_u32_b32 %0, %1, %0")
OP32(v_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
OP32(v_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
OP32(v_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
OP32(v_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
OP32(v_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
OP32C(v_cmp_cndmask, "v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")   // TWO instructions per op (reported per pair)
OP32(v_fma_f32, "v_fma_f32 %0, %0, %1, %2")
OP32(v_mul_f32, "v_mul_f32 %0, %0, %1")
OP32(v_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
OP32(v_rcp_f32, "v_rcp_f32 %0, %0")
OP32(v_add_u32_dpp_row_shr, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
OP32(v_mov_b32_dpp_quad, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OP32(v_add_u32_dpp_row_bcast, "v_add_u32_dpp %0, %1, %0 row_bcast:15 row_mask:0xa bank_mask:0xf")
OP32C(v_readfirstlane_mov, "v_readfirstlane_b32 s20, %0\n\tv_mov_b32 %0, s20")       // TWO instructions per op; s20 clobbered
OP32(v_or_b32, "v_or_b32 %0, %0, %1")
OP32(v_sub_u32, "v_sub_u32 %0, %0, %1")
OP32(v_mov_b32, "v_mov_b32 %0, %1")
OP32(v_not_b32, "v_not_b32 %0, %0")
OP32(v_add_u32_e64, "v_add_u32_e64 %0, %0, %1")
OP32C(v_add_u32_sgpr, "v_add_u32 %0, s20, %0")
OP32(v_add_u32_lit, "v_add_u32 %0, 0x12345, %0")
OP32(v_and_b32_inl, "v_and_b32 %0, 15, %0")
OP32C(v_add_co_u32, "v_add_co_u32 %0, vcc, %0, %1")
OP32C(v_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
OP32C(v_cmp_gt_u32_vcc, "v_cmp_gt_u32 vcc, %0, %1")
OP32C(v_cmp_gt_u32_sgpr, "v_cmp_gt_u32 s[20:21], %0, %1")
OP32(v_lshrrev_b32, "v_lshrrev_b32 %0, 1, %0")
OP32(v_ashrrev_i32, "v_ashrrev_i32 %0, 1, %0")
OP32(v_max_u32, "v_max_u32 %0, %0, %1")
OP32(v_max_i32, "v_max_i32 %0, %0, %1")
OP32(v_mul_i32_i24, "v_mul_i32_i24 %0, %0, %1")
OP32(v_or3_b32, "v_or3_b32 %0, %0, %1, %2")
OP32(v_xad_u32, "v_xad_u32 %0, %0, %1, %2")
OP32(v_bfi_b32, "v_bfi_b32 %0, %0, %1, %2")
OP32(v_xnor_b32, "v_xnor_b32 %0, %0, %1")
OP32(v_add_f32, "v_add_f32 %0, %0, %1")
OP32(v_sub_f32, "v_sub_f32 %0, %0, %1")
OP32(v_max_f32, "v_max_f32 %0, %0, %1")
OP32(v_fmac_f32, "v_fmac_f32 %0, %1, %2")
OP32(v_cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
OP32(v_rndne_f32, "v_rndne_f32 %0, %0")
OP32(v_add_u16, "v_add_u16 %0, %0, %1")
OP32(v_add_u32_sdwa_b0, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")
OP32(v_mbcnt_lo, "v_mbcnt_lo_u32_b32 %0, %1, %0")
OP32(v_sqrt_f32, "v_sqrt_f32 %0, %0")
OP32(v_sin_f32, "v_sin_f32 %0, %0")
OP64(v_fma_f64, "v_fma_f64 %0, %0, %1, %2")
OP64(v_add_f64, "v_add_f64 %0, %0, %1")
OP64(v_mul_f64, "v_mul_f64 %0, %0, %1")
OP64(v_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
OP64(v_lshlrev_b64, "v_lshlrev_b64 %0, 1, %0")
// 64-bit integer forms the compiler makes of size_t index arithmetic (the 32-bit factors are the low halves of b and c)
#define OP64I(NAME, ASM)                                                                                   \
    struct NAME { using T = double; static constexpr const char* name = #NAME;                             \
        static __device__ __forceinline__ void op(double& a, double b, double c) {                         \
            const uint32_t bl = (uint32_t)__double2loint(b), cl = (uint32_t)__double2loint(c);             \
            asm volatile(ASM : "+v"(a) : "v"(bl), "v"(cl), "v"(b) : "s20", "s21"); } };
OP64I(v_mad_u64_u32, "v_mad_u64_u32 %0, s[20:21], %1, %2, %0")
OP64I(v_mad_i64_i32, "v_mad_i64_i32 %0, s[20:21], %1, %2, %0")
OP64I(v_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %3")
OP64I(v_mov_b64, "v_mov_b64 %0, %3")

template <class OP, int CHAINS>
__global__ __launch_bounds__(256) void k_valu(Rec* out, int iters, uint32_t seed) {
    using T = typename OP::T;
    T a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (T)(seed + threadIdx.x * 7 + i);
    T b = (T)(seed * 3 + 1), c = (T)(seed + 5);
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) OP::op(a[CHAINS == 8 ? i : 0], b, c);
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    T s = a[0];
#pragma unroll
    for (int i = 1; i < 8; i++) s = s + a[i];
    if ((threadIdx.x & 63) == 0) {
        Rec r; r.t0 = t0; r.t1 = t1; r.hwid = hw_id(); r.xcc = xcc_id();
        if (s == (T)12345.678) r.xcc |= 0x80000000u;   // keeps the chains alive
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r;
    }
}

// ---- LDS: ds_read_b32 / b64 / b128 (conflict-free, lane-linear), ds_bpermute; dependent form = pointer chase -----------------------------
template <int WIDTH, int CHAINS>   // WIDTH in dwords (1, 2, 4); 0 = ds_bpermute_b32
__global__ __launch_bounds__(256) void k_lds(Rec* out, int iters, uint32_t seed) {
    __shared__ uint32_t L[4096 + 64];
    for (int i = threadIdx.x; i < 4096 + 64; i += 256) L[i] = (WIDTH == 1 && CHAINS == 1) ? (uint32_t)(i * 4) : (uint32_t)i;   // chase: a word holds its own byte offset
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    uint32_t addr = (uint32_t)(size_t)(&L[0]) + ((threadIdx.x >> 6) * 1024 + lane * (WIDTH ? WIDTH : 1)) * 4u % 4096u;
    uint32_t acc = (WIDTH == 1 && CHAINS == 1) ? lane * 4u : seed;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (CHAINS == 8) {
                if (WIDTH == 1) {
                    uint32_t v0, v1, v2, v3, v4, v5, v6, v7;
                    asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:256\n\tds_read_b32 %2, %8 offset:512\n\tds_read_b32 %3, %8 offset:768\n\t"
                                 "ds_read_b32 %4, %8 offset:1024\n\tds_read_b32 %5, %8 offset:1280\n\tds_read_b32 %6, %8 offset:1536\n\tds_read_b32 %7, %8 offset:1792\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(addr) : "memory");
                    acc ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
                } else if (WIDTH == 2) {
                    uint64_t v0, v1, v2, v3, v4, v5, v6, v7;
                    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\tds_read_b64 %3, %8 offset:1536\n\t"
                                 "ds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\tds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(addr & ~7u) : "memory");
                    acc ^= (uint32_t)(v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7);
                } else if (WIDTH == 4) {
                    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                    u4 v0, v1, v2, v3, v4, v5, v6, v7;
                    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                                 "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"((uint32_t)(size_t)(&L[0]) + lane * 16u) : "memory");
                    acc ^= v0.x ^ v1.y ^ v2.z ^ v3.w ^ v4.x ^ v5.y ^ v6.z ^ v7.w;
                } else {
                    uint32_t v0 = acc, v1 = acc + 1, v2 = acc + 2, v3 = acc + 3, v4 = acc + 4, v5 = acc + 5, v6 = acc + 6, v7 = acc + 7;
                    const uint32_t sel = ((lane + 1) & 63) * 4;
                    asm volatile("ds_bpermute_b32 %0, %8, %0\n\tds_bpermute_b32 %1, %8, %1\n\tds_bpermute_b32 %2, %8, %2\n\tds_bpermute_b32 %3, %8, %3\n\t"
                                 "ds_bpermute_b32 %4, %8, %4\n\tds_bpermute_b32 %5, %8, %5\n\tds_bpermute_b32 %6, %8, %6\n\tds_bpermute_b32 %7, %8, %7\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(sel) : "memory");
                    acc ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (WIDTH == 1) {   // pointer chase: the loaded word is the next byte offset
                        uint32_t nx;
                        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nx) : "v"((uint32_t)(size_t)(&L[0]) + acc % 4096u) : "memory");
                        acc = nx;
                    } else {
                        const uint32_t sel = ((lane + 1) & 63) * 4;
                        asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(acc) : "v"(sel) : "memory");
                    }
                }
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        Rec r; r.t0 = t0; r.t1 = t1; r.hwid = hw_id(); r.xcc = xcc_id();
        if (acc == 0x12345679u) r.xcc |= 0x80000000u;
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r;
    }
}

// ---- mixed: a VALU stream (v_add_u32, 8 chains) next to LDS reads in the SAME wave — do the two pipes overlap? ---------------------------
template <int VALU_PER_LDS>
__global__ __launch_bounds__(256) void k_mix(Rec* out, int iters, uint32_t seed) {
    __shared__ uint32_t L[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) L[i] = i;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t addr = (uint32_t)(size_t)(&L[0]) + lane * 4u;
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + i;
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:256\n\tds_read_b32 %2, %8 offset:512\n\tds_read_b32 %3, %8 offset:768\n\t"
                         "ds_read_b32 %4, %8 offset:1024\n\tds_read_b32 %5, %8 offset:1280\n\tds_read_b32 %6, %8 offset:1536\n\tds_read_b32 %7, %8 offset:1792"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(addr) : "memory");
#pragma unroll
            for (int k = 0; k < VALU_PER_LDS; k++) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)::"memory");
            acc ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = acc;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    if ((threadIdx.x & 63) == 0) {
        Rec r; r.t0 = t0; r.t1 = t1; r.hwid = hw_id(); r.xcc = xcc_id();
        if (s == 0x12345679u) r.xcc |= 0x80000000u;
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r;
    }
}

struct Result { double simd_cpi, wave_cpi, wall_cpi; int simds_used, simds_other; };

// group the waves by (xcc, se, sh, cu, simd) and reduce: only SIMDs that held exactly `w` waves count
static Result reduce(const std::vector<Rec>& recs, int w, double insts_per_wave, float ms, double clock_hz, int total_waves) {
    std::map<uint64_t, std::vector<const Rec*>> g;
    for (const Rec& r : recs) {
        const uint64_t key = ((uint64_t)(r.xcc & 0xF) << 32) | (r.hwid & 0x0000FF30u) | ((r.hwid >> 4) & 3u);   // se/sh/cu bits 15:8, simd bits 5:4
        g[key].push_back(&r);
    }
    Result R{0, 0, 0, 0, 0};
    double sum_simd = 0, sum_wave = 0; long nw = 0;
    for (auto& kv : g) {
        if ((int)kv.second.size() != w) { R.simds_other++; continue; }
        uint64_t lo = ~0ull, hi = 0;
        for (const Rec* r : kv.second) { lo = std::min(lo, r->t0); hi = std::max(hi, r->t1); sum_wave += double(r->t1 - r->t0) / insts_per_wave; nw++; }
        sum_simd += double(hi - lo) / (insts_per_wave * w);
        R.simds_used++;
    }
    R.simd_cpi = R.simds_used ? sum_simd / R.simds_used : 0;
    R.wave_cpi = nw ? sum_wave / nw : 0;
    R.wall_cpi = ms * 1e-3 * clock_hz / (insts_per_wave * total_waves / 1024.0);   // launch wall time x nominal clock / instructions per SIMD (1024 SIMDs)
    return R;
}

template <class K>
static void run(const char* name, const char* mode, K kernel, int w, int iters, double insts_per_iter, Rec* d_out, double clock_hz) {
    const int wgs = 256 * w;               // 256 CUs x w workgroups of 4 waves -> w waves per SIMD when the dispatcher spreads them evenly
    const int waves = wgs * 4;
    std::vector<Rec> h(waves);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), 0, 0, d_out, 4, 1u);   // warm
    CK(hipDeviceSynchronize());
    Result best{1e30, 0, 0, 0, 0};
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), 0, 0, d_out, iters, 7u + rep);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), d_out, waves * sizeof(Rec), hipMemcpyDeviceToHost));
        Result R = reduce(h, w, insts_per_iter * iters, ms, clock_hz, waves);
        if (R.simds_used > 0 && R.simd_cpi < best.simd_cpi) best = R;
    }
    printf("%s,%s,%d,%.3f,%.3f,%.3f,%d,%d\n", name, mode, w, best.simd_cpi, best.wave_cpi, best.wall_cpi, best.simds_used, best.simds_other);
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <class OP>
static void run_op(Rec* d_out, double clock_hz, double insts_per_op = 1.0) {
    for (int w : {1, 2, 4, 8}) run(OP::name, "indep8", k_valu<OP, 8>, w, 64, 64.0 * insts_per_op, d_out, clock_hz);
    for (int w : {1, 2, 4, 8}) run(OP::name, "dep1", k_valu<OP, 1>, w, 64, 64.0 * insts_per_op, d_out, clock_hz);
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const double clock_hz = p.clockRate * 1e3;
    fprintf(stderr, "%s, %d CUs, clockRate %.0f MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1e3);
    Rec* d_out; CK(hipMalloc(&d_out, sizeof(Rec) * 256 * 8 * 4));
    printf("# %s, %d CUs, nominal %.0f MHz; cycles = s_memtime ticks; simd_cpi = SIMD cycles per wave64 instruction with w waves resident on the SIMD\n", p.gcnArchName,
           p.multiProcessorCount, p.clockRate / 1e3);
    printf("instruction,mode,waves_per_simd,simd_cycles_per_inst,wave_cycles_per_inst,wall_cycles_per_inst_nominal_clock,simds_with_w_waves,simds_other\n");
    run_op<v_add_u32>(d_out, clock_hz);
    run_op<v_and_b32>(d_out, clock_hz);
    run_op<v_xor_b32>(d_out, clock_hz);
    run_op<v_lshlrev_b32>(d_out, clock_hz);
    run_op<v_lshl_add_u32>(d_out, clock_hz);
    run_op<v_add3_u32>(d_out, clock_hz);
    run_op<v_and_or_b32>(d_out, clock_hz);
    run_op<v_bfe_u32>(d_out, clock_hz);
    run_op<v_min_u32>(d_out, clock_hz);
    run_op<v_max3_u32>(d_out, clock_hz);
    run_op<v_alignbyte_b32>(d_out, clock_hz);
    run_op<v_lerp_u8>(d_out, clock_hz);
    run_op<v_sad_u8>(d_out, clock_hz);
    run_op<v_pk_min_u16>(d_out, clock_hz);
    run_op<v_pk_max_u16>(d_out, clock_hz);
    run_op<v_pk_add_u16>(d_out, clock_hz);
    run_op<v_pk_sub_u16>(d_out, clock_hz);
    run_op<v_pk_mad_u16>(d_out, clock_hz);
    run_op<v_dot4_u32_u8>(d_out, clock_hz);
    run_op<v_dot2_u32_u16>(d_out, clock_hz);
    run_op<v_perm_b32>(d_out, clock_hz);
    run_op<v_bcnt_u32_b32>(d_out, clock_hz);
    run_op<v_mad_u32_u24>(d_out, clock_hz);
    run_op<v_mul_u32_u24>(d_out, clock_hz);
    run_op<v_mul_hi_u32_u24>(d_out, clock_hz);
    run_op<v_mul_lo_u32>(d_out, clock_hz);
    run_op<v_mul_hi_u32>(d_out, clock_hz);
    run_op<v_cmp_cndmask>(d_out, clock_hz, 2.0);
    run_op<v_fma_f32>(d_out, clock_hz);
    run_op<v_mul_f32>(d_out, clock_hz);
    run_op<v_cvt_f32_u32>(d_out, clock_hz);
    run_op<v_rcp_f32>(d_out, clock_hz);
    run_op<v_add_u32_dpp_row_shr>(d_out, clock_hz);
    run_op<v_mov_b32_dpp_quad>(d_out, clock_hz);
    run_op<v_add_u32_dpp_row_bcast>(d_out, clock_hz);
    run_op<v_readfirstlane_mov>(d_out, clock_hz, 2.0);
    run_op<v_or_b32>(d_out, clock_hz);
    run_op<v_sub_u32>(d_out, clock_hz);
    run_op<v_mov_b32>(d_out, clock_hz);
    run_op<v_not_b32>(d_out, clock_hz);
    run_op<v_add_u32_e64>(d_out, clock_hz);
    run_op<v_add_u32_sgpr>(d_out, clock_hz);
    run_op<v_add_u32_lit>(d_out, clock_hz);
    run_op<v_and_b32_inl>(d_out, clock_hz);
    run_op<v_add_co_u32>(d_out, clock_hz);
    run_op<v_cndmask_b32>(d_out, clock_hz);
    run_op<v_cmp_gt_u32_vcc>(d_out, clock_hz);
    run_op<v_cmp_gt_u32_sgpr>(d_out, clock_hz);
    run_op<v_lshrrev_b32>(d_out, clock_hz);
    run_op<v_ashrrev_i32>(d_out, clock_hz);
    run_op<v_max_u32>(d_out, clock_hz);
    run_op<v_max_i32>(d_out, clock_hz);
    run_op<v_mul_i32_i24>(d_out, clock_hz);
    run_op<v_or3_b32>(d_out, clock_hz);
    run_op<v_xad_u32>(d_out, clock_hz);
    run_op<v_bfi_b32>(d_out, clock_hz);
    run_op<v_xnor_b32>(d_out, clock_hz);
    run_op<v_add_f32>(d_out, clock_hz);
    run_op<v_sub_f32>(d_out, clock_hz);
    run_op<v_max_f32>(d_out, clock_hz);
    run_op<v_fmac_f32>(d_out, clock_hz);
    run_op<v_cvt_u32_f32>(d_out, clock_hz);
    run_op<v_rndne_f32>(d_out, clock_hz);
    run_op<v_add_u16>(d_out, clock_hz);
    run_op<v_add_u32_sdwa_b0>(d_out, clock_hz);
    run_op<v_mbcnt_lo>(d_out, clock_hz);
    run_op<v_sqrt_f32>(d_out, clock_hz);
    run_op<v_sin_f32>(d_out, clock_hz);
    run_op<v_fma_f64>(d_out, clock_hz);
    run_op<v_add_f64>(d_out, clock_hz);
    run_op<v_mul_f64>(d_out, clock_hz);
    run_op<v_pk_fma_f32>(d_out, clock_hz);
    run_op<v_lshlrev_b64>(d_out, clock_hz);
    run_op<v_mad_u64_u32>(d_out, clock_hz);
    run_op<v_mad_i64_i32>(d_out, clock_hz);
    run_op<v_lshl_add_u64>(d_out, clock_hz);
    run_op<v_mov_b64>(d_out, clock_hz);
    for (int w : {1, 2, 4, 8}) run("ds_read_b32", "indep8", k_lds<1, 8>, w, 64, 64.0, d_out, clock_hz);
    for (int w : {1, 2, 4, 8}) run("ds_read_b32", "dep1", k_lds<1, 1>, w, 16, 64.0, d_out, clock_hz);
    for (int w : {1, 2, 4, 8}) run("ds_read_b64", "indep8", k_lds<2, 8>, w, 64, 64.0, d_out, clock_hz);
    for (int w : {1, 2, 4, 8}) run("ds_read_b128", "indep8", k_lds<4, 8>, w, 64, 64.0, d_out, clock_hz);
    for (int w : {1, 2, 4, 8}) run("ds_bpermute_b32", "indep8", k_lds<0, 8>, w, 64, 64.0, d_out, clock_hz);
    for (int w : {1, 2, 4, 8}) run("ds_bpermute_b32", "dep1", k_lds<0, 1>, w, 16, 64.0, d_out, clock_hz);
    // per 8 ds_read_b32: 8 / 16 / 32 v_add_u32 in the same wave; figure = SIMD cycles per (8 LDS reads + the VALU block) / 8
    for (int w : {1, 2, 4, 8}) run("mix_8lds_8valu", "per_lds_read", k_mix<1>, w, 64, 64.0, d_out, clock_hz);
    for (int w : {1, 2, 4, 8}) run("mix_8lds_16valu", "per_lds_read", k_mix<2>, w, 64, 64.0, d_out, clock_hz);
    for (int w : {1, 2, 4, 8}) run("mix_8lds_32valu", "per_lds_read", k_mix<4>, w, 64, 64.0, d_out, clock_hz);
    CK(hipFree(d_out));
    return 0;
}
