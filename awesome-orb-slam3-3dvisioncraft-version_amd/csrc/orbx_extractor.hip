// orbx_extractor.hip — stage 1 of the hot path on gfx950: ORBextractor (reference src/ORBextractor.cc).
//
// One handle = one (image size, config).  A call processes a batch of independent frames with
//   k_resize2  x (nlevels-1)   E1  ComputePyramid            ORBextractor.cc:1158-1183 (cv::resize INTER_LINEAR, fixed point; k_resize for scale factors > 1.3)
//   k_fast<0>, k_fast<1>       E2  per-cell FAST-9/16 + NMS   ORBextractor.cc:763-855   (cv::FAST semantics per 30-px cell ROI): batches detect at iniThFAST,
//                              then again the cells that came back empty at minThFAST (second launch); k_fast<2> = one pass (single frames, sparse scenes)
//   k_octree                   E3  DistributeOctTree          ORBextractor.cc:537-761   (+ E4/E8 ordering ranks)
//   k_describe2                E5-E8 IC_Angle, 7x7 blur (at the sampled points), rBRIEF, output assembly  :75-145, 1093-1155
//                              (two key points per wave; k_describe is the one-key-point-per-wave form, -DDESC_KPW=1)
// All integer/fixed-point work is bit-exact w.r.t. the oracle; float expressions are written so that they
// round exactly as the reference's (compile with -ffp-contract=off and correctly rounded fp32 division).
//
// LDS: every kernel carves the dynamic region `orb_smem` only (16-byte aligned base, no static LDS).
#include <hip/hip_runtime.h>

#include <climits>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/orbhip.h"
#include "lds_optin.inc"
#include <mutex>

#define ORBX_MAX_LEVELS 16
// Phase timers for kernel experiments (-DORBX_PROF builds under exp_so/ only; never in the product build): lane 0 of every wave adds the
// s_memtime delta of each phase to g_prof[kernel][phase]; orbx_debug_prof() reads and clears them.
#ifdef ORBX_PROF
__device__ unsigned long long g_prof[2][8][256];   // [kernel][phase][shard]: sharded so that the end-of-wave atomics do not serialise
#define PROF_DECL unsigned long long prof_t = clock64(), prof_d[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_MARK(k, slot) do { const unsigned long long t_ = clock64(); prof_d[slot] += t_ - prof_t; prof_t = t_; } while (0)
#define PROF_FLUSH(k) do { if ((threadIdx.x & 63) < 8) atomicAdd(&g_prof[k][threadIdx.x & 7][(blockIdx.x * 4 + (threadIdx.x >> 6)) & 255], prof_d[threadIdx.x & 7]); } while (0)
#else
#define PROF_DECL do {} while (0)
#define PROF_MARK(k, slot) do {} while (0)
#define PROF_FLUSH(k) do {} while (0)
#endif
#define ORBX_EVENTS_ON(h) ((h)->timing && !(h)->capturing)   // the five stage events of a batch call (orbx_enable_timing / orbx_last_timing)
#define ORBX_EDGE 19          // EDGE_THRESHOLD, ORBextractor.cc:72
#define ORBX_MINB 16          // EDGE_THRESHOLD-3, ORBextractor.cc:769
#define FAST_QCAP 2048        // sizes k_fast's queues: a wave's pre-test queue holds FAST_QCAP / 4 + 64 entries, the staged emit list FAST_QCAP / 2
#define FAST_MAXCELLS 32

static const int8_t h_pattern[1024] = {
#include "orb_pattern.inc"
};
static __constant__ float4 c_patternf[256];   // the same pairs as floats, stored (x0, x1, y0, y1): the packed rotation takes (x0, x1) and (y0, y1) as register pairs straight from the load
static __constant__ int c_umax[16];
// n / g == (n * c_div16[g]) >> 16 for n * g < 65536: k_fast's index maps divide thread ids (< 256) by workgroup-uniform divisors (< 16) — a
// 24-bit multiply and a shift instead of the ~20-instruction integer division sequence
static __constant__ uint32_t c_div16[16] = {0, 65536, 32768, 21846, 16384, 13108, 10923, 9363, 8192, 7282, 6554, 5958, 5462, 5042, 4682, 4370};
static __constant__ uint32_t c_icmask[16][12];   // [|v|][dword k of the patch row]: 0xFF where |col - 21| <= umax[|v|] (IC_Angle's circular patch)

// ============================================================================================================
// E1  pyramid level:  dst(level l) = cv::resize(src(level l-1), INTER_LINEAR)  — 11-bit fixed point
// ============================================================================================================
// XCD-aware work mapping.  Workgroups are dealt to the 8 XCDs round-robin by linear id, and each XCD has its own 4 MB L2.  With a
// (unit, frame) grid the units of ONE frame are spread over all eight XCDs, so every XCD pulls the same frame's pyramid (1.1 MB) through
// its own L2: up to 8x the fill traffic, and these kernels' load phases run at several TB/s.  Instead the grid is 1-D and XCD x takes the
// frames f with f % 8 == x, walking a frame's units in order: a frame's pyramid is fetched once and stays L2-resident while it is worked on.
// grid = units_per_frame * 8 * ceil(batch / 8); returns false for the padding workgroups of a batch that is not a multiple of 8.
static __device__ __forceinline__ bool xcd_frame_unit(const int unitsPerFrame, const int batch, int* frame, int* unit, const int lead = 0) {   // lead: workgroups in front of the mapped ones, a multiple of 8
    const int xcd = blockIdx.x & 7, slot = (int)(blockIdx.x - lead) >> 3;
    const int fr = (slot / unitsPerFrame) * 8 + xcd;
    *frame = fr;
    *unit = slot - (slot / unitsPerFrame) * unitsPerFrame;
    return fr < batch;
}
// The same with the division as a multiplication by magic = floor(2^32 / unitsPerFrame) + 1 (host: xcd_units_magic): exact while slot * unitsPerFrame < 2^32
// (the host passes 0 for larger grids).
// The compiler's own 32-bit division is ~30 instructions around a v_rcp_f32 and a v_readfirstlane — at the head of every workgroup, in front of its first load.
static __device__ __forceinline__ bool xcd_frame_unit_m(const int unitsPerFrame, const uint32_t magic, const int batch, int* frame, int* unit, const int lead = 0, const bool reverse = false) {
    const int xcd = blockIdx.x & 7;
    const uint32_t slot = (uint32_t)(blockIdx.x - lead) >> 3;
    const uint32_t q0 = magic ? (uint32_t)(((unsigned long long)slot * magic) >> 32) : slot / (uint32_t)unitsPerFrame;   // (0: a grid too large for the magic to be exact)
    const uint32_t q = reverse ? (uint32_t)((batch + 7) >> 3) - 1u - q0 : q0;   // the XCD's frames last to first: what the launch before this one wrote last is still in this XCD's L2
    const int fr = (int)q * 8 + xcd;
    *frame = fr;
    *unit = (int)(slot - q0 * (uint32_t)unitsPerFrame);
    return fr < batch;
}
static inline uint32_t xcd_units_magic(int unitsPerFrame, int batch) {
    const unsigned long long u = (unsigned long long)std::max(unitsPerFrame, 1), slots = u * (unsigned long long)((batch + 7) / 8);   // slot < slots
    if (u < 2 || slots * u >= 0x100000000ull) return 0u;   // (u == 1: the magic itself would not fit 32 bits)
    return (uint32_t)(0x100000000ull / u + 1ull);
}

struct ResizeParams {
    const uint8_t* src; size_t sFrame; int sStride, sw, sh;
    uint8_t* dst; size_t dFrame; int dStride, dw, dh;
    double scale_x, scale_y;   // 1 / ((double)dw / sw), 1 / ((double)dh / sh)  — cv::resize's scale_x / scale_y
    const int* coef;           // k_resize2: per-level tables xs[dw] | xw[dw] | ys[dh] | yw[dh] (resize_coef of every column / row)
    const int* tileTab;        // k_resize2: per-level staging footprints {xal, ndw} x tilesX | {ylo, nrows} x tilesY (resize2_footprint, built at orbx_create)
    int tilesX, tilesY, batch; // k_resize2 with R2_XCD: the frame-per-XCD 1-D grid
    uint32_t unitsMagic;       // xcd_units_magic(tilesX * ceil(tilesY / R2_PAIR))
    uint32_t tilesXMagic;      // floor(2^32 / tilesX) + 1 (tilesX >= 2), or 0 + the plain path for one tile column
    int reverse;               // 1: the frames of an XCD in descending order (alternate levels: see xcd_frame_unit_m)
    // the call's bookkeeping, carried by ONE extra workgroup at the front of the level-1 launch (frame_order_body; null: none)
    int* ordCand; int* ordOut; uint32_t* ordHostMax; uint32_t* ordRetry; int ordLevels; uint32_t ordTiles;
};

// cv::resize coefficient of one destination coordinate, computed in-kernel with the same IEEE
// double/float operations as OpenCV's table setup (no table loads on the critical path): source index + the two 11-bit weights.
static __host__ __device__ __forceinline__ void resize_coef(int d, double scale, int slen, bool clampIndex, int& s0, int& w0, int& w1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int si = (int)floorf(f);
    f -= (float)si;
    if (clampIndex) {   // x: OpenCV clamps index and weight (resize.cpp alpha table); y keeps the weight and clips rows later
        if (si < 0) { f = 0.f; si = 0; }
        if (si >= slen - 1) { f = 0.f; si = slen - 1; }
    }
    s0 = si;
    w0 = (int)rintf((1.f - f) * 2048.f);   // cvRound: round half to even (default rounding mode)
    w1 = (int)rintf(f * 2048.f);
    w0 = w0 < -32768 ? -32768 : (w0 > 32767 ? 32767 : w0);
    w1 = w1 < -32768 ? -32768 : (w1 > 32767 ? 32767 : w1);
}

// One workgroup = one 64 x 16 destination tile: the <= 21 x 84 source footprint is staged in LDS with coalesced aligned dword
// row loads (2 KB), every thread then produces 4 horizontally adjacent pixels (one dword store) from LDS byte reads.
#define RS_TW 64
#define RS_TH 16
#define RS_PITCH 92      // bytes per staged source row (23 dwords, odd): 64*scale + 2 taps + 3 alignment slack for scale <= 1.3
#define RS_ROWS 24       // staged rows: 16*scale + 2 for scale <= 1.3 (larger scale factors take the direct path)
static __global__ __launch_bounds__(256) void k_resize(ResizeParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    int* cxs = (int*)orb_smem;                 // [64] source column of each destination column of the tile
    int* cxw = cxs + RS_TW;                    // [64] a0 | a1 << 16
    int* cys = cxw + RS_TW;                    // [16] source row (unclipped)
    int* cyw = cys + RS_TH;                    // [16] b0 | b1 << 16
    uint8_t* tile = (uint8_t*)(cyw + RS_TH);   // [RS_ROWS][RS_PITCH]
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int bx0 = blockIdx.x * RS_TW, by0 = blockIdx.y * RS_TH;
    const int dx0 = bx0 + tx * 4, dy = by0 + ty;
    const uint8_t* S = P.src + (size_t)blockIdx.z * P.sFrame;
    // the tile's 64 + 16 coefficient sets, one per thread (FP64/FP32 setup math is not repeated per pixel)
    if (threadIdx.x < RS_TW) {
        int s0, w0, w1;
        resize_coef(min(bx0 + (int)threadIdx.x, P.dw - 1), P.scale_x, P.sw, true, s0, w0, w1);
        cxs[threadIdx.x] = s0; cxw[threadIdx.x] = (w0 & 0xFFFF) | (w1 << 16);
    } else if (threadIdx.x < RS_TW + RS_TH) {
        const int t = threadIdx.x - RS_TW;
        int s0, w0, w1;
        resize_coef(min(by0 + t, P.dh - 1), P.scale_y, P.sh, false, s0, w0, w1);
        cys[t] = s0; cyw[t] = (w0 & 0xFFFF) | (w1 << 16);
    }
    __syncthreads();
    // source footprint of the tile
    const int xal = cxs[0] & ~3;
    const int xend = min(cxs[RS_TW - 1] + 1, P.sw - 1);  // last source column read
    const int ylo = max(cys[0], 0), yhi = min(cys[RS_TH - 1] + 1, P.sh - 1);
    const int ndw = ((xend - xal) >> 2) + 1, nrows = yhi - ylo + 1;
    const bool staged = ndw * 4 <= RS_PITCH && nrows <= RS_ROWS;   // block-uniform
    if (staged) {
        int r = threadIdx.x / ndw, c = threadIdx.x - r * ndw;
        const int dr = 256 / ndw, dc = 256 - dr * ndw;
        for (int i = threadIdx.x; i < nrows * ndw; i += 256) {
            *(uint32_t*)(tile + r * RS_PITCH + 4 * c) = *(const uint32_t*)(S + (size_t)(ylo + r) * P.sStride + xal + 4 * c);
            r += dr; c += dc;
            if (c >= ndw) { c -= ndw; r++; }
        }
    }
    __syncthreads();
    if (dy >= P.dh || dx0 >= P.dw) return;
    int sy0 = cys[ty];
    const int bw = cyw[ty], b0 = (int)(short)(bw & 0xFFFF), b1 = bw >> 16;
    int sy1 = sy0 + 1;
    sy0 = sy0 < 0 ? 0 : (sy0 < P.sh ? sy0 : P.sh - 1);
    sy1 = sy1 < 0 ? 0 : (sy1 < P.sh ? sy1 : P.sh - 1);
    // NB: LDS and global operands are kept in separate code paths and LDS offsets are formed as in-range indices — a mixed
    // (flat) pointer that is moved below the LDS aperture base faults on gfx950.
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int sx = cxs[tx * 4 + k], aw = cxw[tx * 4 + k];
        const int a0 = (int)(short)(aw & 0xFFFF), a1 = aw >> 16;
        const int x1 = sx + 1 < P.sw ? sx + 1 : P.sw - 1;   // weight is 0 there (fx forced to 0)
        int p00, p01, p10, p11;
        if (staged) {
            const int r0 = (sy0 - ylo) * RS_PITCH, r1 = (sy1 - ylo) * RS_PITCH;
            p00 = tile[r0 + sx - xal]; p01 = tile[r0 + x1 - xal]; p10 = tile[r1 + sx - xal]; p11 = tile[r1 + x1 - xal];
        } else {
            const uint8_t* G0 = S + (size_t)sy0 * P.sStride;
            const uint8_t* G1 = S + (size_t)sy1 * P.sStride;
            p00 = G0[sx]; p01 = G0[x1]; p10 = G1[sx]; p11 = G1[x1];
        }
        const int t0 = p00 * a0 + p01 * a1;
        const int t1 = p10 * a0 + p11 * a1;
        const int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
        out |= (uint32_t)(v & 255) << (8 * k);
    }
    uint8_t* D = P.dst + (size_t)blockIdx.z * P.dFrame + (size_t)dy * P.dStride + dx0;
    if (dx0 + 3 < P.dw) {
        *(uint32_t*)D = out;  // dStride and dx0 are multiples of 4
    } else {
        for (int k = 0; k < 4 && dx0 + k < P.dw; k++) D[k] = (uint8_t)(out >> (8 * k));
    }
}

// Separable form of the same arithmetic for scale factors <= 1.3 (every ORB-SLAM3 configuration: 1.2): one workgroup = one 64 x 32
// destination tile.  The per-column / per-row coefficients come from per-level tables built once at orbx_create with the same
// resize_coef() (host side, same IEEE operations).  H pass: every staged source row is filtered once per destination column
// (t = p0*a0 + p1*a1 as one v_dot2_u32_u16 on a byte pair picked by v_perm from an 8-byte window; cv::resize's ">> 4" is the one v_and that
// clears the low four bits — the V pass is one v_mul_hi_u32_u24 per tap with the weight pre-shifted: ((b << 12) * (16 * (t >> 4))) >> 32 == (b * (t >> 4)) >> 16).
// V pass: 4 adjacent pixels per thread from two 16-byte LDS reads.
#ifndef R2_XCD
#define R2_XCD 1
#endif
#ifndef R2_TH
#define R2_TH 32
#endif
#define R2_ROWS (R2_TH * 13 / 10 + 5)   // R2_TH * 1.3 + 2 taps + 2 slack (estimated footprint): 46 rows for 32
#define R2_HP 68          // H-buffer pitch in u32 (64 + 4: rows skewed across banks, 16-byte aligned)
#define R2_SMEM ((RS_TW * 2 + R2_TH * 4) * 4 + R2_ROWS * R2_HP * 4 + R2_ROWS * RS_PITCH)
// Staging footprint of a k_resize2 tile along one axis: first staged source coordinate (dword-aligned for x) and the staged extent, from a FLOAT
// estimate of the first / last coefficient widened by one on each side (the exact indices come from the double-precision tables and can differ
// by one).  Workgroup-uniform float arithmetic — 30 VALU instructions that every lane of every tile repeated (there is no scalar float unit) —
// so it is evaluated once per level at orbx_create and read back through the scalar cache.
// ---- the whole pyramid of ONE frame in ONE launch (the single-frame entry point) -----------------------------------------------------------------
// Seven dependent k_resize2 launches cost a single frame 7 x 4.8 us of launch-to-launch latency for 0.7 ... 3 us of work each.  Here a workgroup takes a
// tile of the TOP level and computes, level by level from the image up, every pixel its tile depends on — region C_l of level l, two LDS buffers
// swapped per level — plus its own share of every level (the levels are partitioned among the tiles: a pixel no higher level reads must still
// exist, FAST and rBRIEF read it).  Regions of neighbouring workgroups overlap by their footprint halos; the overlap is computed twice from the
// same inputs with the same integer arithmetic and stored twice with the same bytes.  Arithmetic per pixel = k_resize's (cv::resize INTER_LINEAR
// 8UC1 fixed point), coefficients from the same per-level tables.  Regions (host, orbx_create):
// C_top = own tile; C_l = bounding box of (own share of level l) and (footprint of C_{l+1} in level l); region 0 = footprint of C_1 in the image.
#define PYC_T 1024
struct PyrChainParams {
    const uint8_t* img; size_t imgFrame; int imgStride;
    uint8_t* pyr; size_t pyrFrame;
    size_t planeOff[ORBX_MAX_LEVELS]; int stride[ORBX_MAX_LEVELS]; int w[ORBX_MAX_LEVELS]; int h[ORBX_MAX_LEVELS];
    const int* coef; int coefOff[ORBX_MAX_LEVELS];   // xs[dw] | xw[dw] | ys[dh] | yw[dh] of level l >= 1 at coef + coefOff[l]
    const int4* regions;                             // [tiles][nlevels] {x0, y0, w, h}
    int nlevels, bufBytes;                           // bufBytes: one LDS pixel buffer (largest region of any tile and level, 16-byte multiple)
};
static __global__ __launch_bounds__(PYC_T) void k_pyramid_chain(PyrChainParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int tid = threadIdx.x, tile = blockIdx.x, frame = blockIdx.y;
    uint8_t* buf[2] = {orb_smem, orb_smem + P.bufBytes};
    int* ctab = (int*)(orb_smem + 2 * P.bufBytes);   // per level l >= 1: xs | xw of the region's columns, ys | yw of its rows
    const int4* R = P.regions + (size_t)tile * P.nlevels;
    {   // region 0 of the image -> buf[0]; every level's coefficient slices -> LDS (one memory round trip for all of it)
        const int4 r0 = R[0];
        const uint8_t* src = P.img + (size_t)frame * P.imgFrame + (size_t)r0.y * P.imgStride + r0.x;
        for (int y = tid >> 6; y < r0.w; y += PYC_T / 64)
            for (int x = tid & 63; x < r0.z; x += 64) buf[0][y * r0.z + x] = src[(size_t)y * P.imgStride + x];
        int off = 0;
        for (int l = 1; l < P.nlevels; l++) {
            const int4 r = R[l];
            const int* xs = P.coef + P.coefOff[l]; const int* xw = xs + P.w[l]; const int* ys = xw + P.w[l]; const int* yw = ys + P.h[l];
            for (int i = tid; i < r.z; i += PYC_T) { ctab[off + i] = xs[r.x + i]; ctab[off + r.z + i] = xw[r.x + i]; }
            for (int i = tid; i < r.w; i += PYC_T) { ctab[off + 2 * r.z + i] = ys[r.y + i]; ctab[off + 2 * r.z + r.w + i] = yw[r.y + i]; }
            off += 2 * (r.z + r.w);
        }
    }
    __syncthreads();
    int off = 0;
    for (int l = 1; l < P.nlevels; l++) {
        const int4 rs = R[l - 1], rd = R[l];
        const uint8_t* S = buf[(l - 1) & 1];
        uint8_t* Dl = buf[l & 1];
        uint8_t* G = P.pyr + P.planeOff[l] + (size_t)frame * P.pyrFrame + (size_t)rd.y * P.stride[l] + rd.x;
        const int* cxs = ctab + off; const int* cxw = cxs + rd.z; const int* cys = cxw + rd.z; const int* cyw = cys + rd.w;
        const int sw = P.w[l - 1], sh = P.h[l - 1];
        for (int y = tid >> 6; y < rd.w; y += PYC_T / 64) {
            int sy0 = cys[y];
            const int bw = cyw[y], b0 = (int)(short)(bw & 0xFFFF), b1 = bw >> 16;
            int sy1 = sy0 + 1;
            sy0 = sy0 < 0 ? 0 : (sy0 < sh ? sy0 : sh - 1);       // rows are clipped after the weights are fixed (resize.cpp)
            sy1 = sy1 < 0 ? 0 : (sy1 < sh ? sy1 : sh - 1);
            const int o0 = (sy0 - rs.y) * rs.z - rs.x, o1 = (sy1 - rs.y) * rs.z - rs.x;   // (indices, not pointers: an LDS pointer must not leave its buffer)
            for (int x = tid & 63; x < rd.z; x += 64) {
                const int sx = cxs[x], aw = cxw[x];
                const int a0 = (int)(short)(aw & 0xFFFF), a1 = aw >> 16;
                const int x1 = sx + 1 < sw ? sx + 1 : sw - 1;    // weight is 0 there (fx forced to 0)
                const int t0 = S[o0 + sx] * a0 + S[o0 + x1] * a1;
                const int t1 = S[o1 + sx] * a0 + S[o1 + x1] * a1;
                const int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
                Dl[y * rd.z + x] = (uint8_t)v;
                G[(size_t)y * P.stride[l] + x] = (uint8_t)v;
            }
        }
        off += 2 * (rd.z + rd.w);
        __syncthreads();
    }
}

static inline void resize2_footprint(int t0, int tlen, int dlen, int slen, double scale, bool alignX, int& lo, int& ext) {
    const float fs = (float)scale;
    const int a = std::max((int)floorf(((float)t0 + 0.5f) * fs - 0.5f) - 1, 0);
    const int e = std::min((int)floorf(((float)(std::min(t0 + tlen, dlen) - 1) + 0.5f) * fs - 0.5f) + 2, slen - 1);
    if (alignX) { lo = a & ~3; ext = ((e - lo) >> 2) + 1; }   // dwords
    else { lo = a; ext = e - a + 1; }                          // rows
}
static __device__ __forceinline__ void frame_order_body(unsigned char* smem, int* candCount, const int nlevels, const int batch, int* order, uint32_t* hostMax,
                                                        uint32_t* retry, const uint32_t retryTiles);   // (below, with k_frame_order)
#ifndef R2_REVERSE
#define R2_REVERSE 1   // even levels walk an XCD's frames backwards (experiment switch)
#endif
#ifndef R2_PAIR
#define R2_PAIR 2         // vertically adjacent destination tiles per workgroup (2: the second tile's staging loads are in flight during the first tile's H pass)
#endif
template <bool CARRY>   // CARRY: the level-1 launch of a batch call, with the call's bookkeeping workgroup in front (a template parameter, not a field test: the
                        // plain instantiation must not wait for a kernel argument before it asks for its tile)
static __global__ __launch_bounds__(256) void k_resize2(ResizeParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    typedef unsigned short u16x2 __attribute__((vector_size(4)));
    int* cxs = (int*)orb_smem;                 // [64] source column of each destination column of the tile
    int* cxw = cxs + RS_TW;                    // [64] a0 | a1 << 16
    uint4* rowt = (uint4*)(cxw + RS_TW);       // [32] {hbuf byte offset of source row 0, of source row 1, b0 << 7, b1 << 7}
    uint32_t* hbuf = (uint32_t*)(rowt + R2_TH);   // [R2_ROWS][R2_HP]  16 * ((p0*a0 + p1*a1) >> 4)
    uint8_t* tile = (uint8_t*)(hbuf + R2_ROWS * R2_HP);   // [R2_ROWS][RS_PITCH]
    const int tid = threadIdx.x;
    constexpr int NP = R2_PAIR;
    const int unitsY = (P.tilesY + NP - 1) / NP;
#if R2_XCD
    // the launch's FIRST workgroup (it runs ~50 us on 256 threads: dispatched last it was the launch's tail): frame order + counters of the call, no
    // tile of its own; seven idle ones behind it keep every tile on the XCD its frame number names
    if (CARRY && blockIdx.x < 8) {
        if (blockIdx.x == 0) frame_order_body(orb_smem, P.ordCand, P.ordLevels, P.batch, P.ordOut, P.ordHostMax, P.ordRetry, P.ordTiles);
        return;
    }
    int frameZ, tileI;
    if (!xcd_frame_unit_m(P.tilesX * unitsY, P.unitsMagic, P.batch, &frameZ, &tileI, CARRY ? 8 : 0, R2_REVERSE && P.reverse)) return;
    const int uyI = P.tilesXMagic ? (int)(((unsigned long long)(uint32_t)tileI * P.tilesXMagic) >> 32) : tileI, txI = tileI - uyI * P.tilesX;   // tileI / tilesX (magic: exact for tileI * tilesX < 2^32)
#else
    const int frameZ = blockIdx.z;
    const int txI = blockIdx.x, uyI = blockIdx.y;
#endif
    const int bx0 = txI * RS_TW;
    const uint8_t* S = P.src + (size_t)frameZ * P.sFrame;
    const int* xs = P.coef; const int* xw = xs + P.dw; const int* ys = xw + P.dw; const int* yw = ys + P.dh;
    // Source footprint of the tile (resize2_footprint, tabulated per level): two scalar loads from a table the scalar cache holds, instead of
    // float arithmetic on workgroup-uniform values in every lane; the staging loads still do not wait for the coefficient-table (vector) loads.
    const int2 fx = ((const int2*)P.tileTab)[txI];
    const int xal = fx.x, ndw = fx.y;                      // <= RS_PITCH/4 for scale <= 1.3
    constexpr int NV = (R2_ROWS + 7) / 8;
    const int sc = tid & 31, sr0 = tid >> 5;              // staging: lane = dword column, 8 rows per pass (coalesced aligned row segments)
    const bool son = sc < ndw;
    // the footprint rows of destination tile row tyI, and its staging loads into registers
    auto issue = [&](const int tyI, int& ylo, int& nrows, uint32_t* v) {
        const int2 fy = ((const int2*)P.tileTab)[P.tilesX + tyI];
        ylo = fy.x; nrows = fy.y;                          // <= R2_ROWS for scale <= 1.3
        const uint8_t* g = S + (size_t)(ylo + sr0) * P.sStride + xal + 4 * sc;
#pragma unroll
        for (int k = 0; k < NV; k++)
            if (son && sr0 + 8 * k < nrows) v[k] = *(const uint32_t*)(g + (size_t)(8 * k) * P.sStride);
    };
    auto commit = [&](const int nrows, const uint32_t* v) {
        uint8_t* t = tile + sr0 * RS_PITCH + 4 * sc;
#pragma unroll
        for (int k = 0; k < NV; k++)
            if (son && sr0 + 8 * k < nrows) *(uint32_t*)(t + 8 * k * RS_PITCH) = v[k];
    };
    auto row_table = [&](const int by0, const int ylo) {
        if (tid >= RS_TW && tid < RS_TW + R2_TH) {
            const int t = tid - RS_TW, y = min(by0 + t, P.dh - 1);
            int sy0 = ys[y];
            const uint32_t bw = (uint32_t)yw[y];
            int sy1 = sy0 + 1;
            sy0 = sy0 < 0 ? 0 : (sy0 < P.sh ? sy0 : P.sh - 1);   // rows are clipped after the weights are fixed (resize.cpp)
            sy1 = sy1 < 0 ? 0 : (sy1 < P.sh ? sy1 : P.sh - 1);
            rowt[t] = make_uint4((uint32_t)((sy0 - ylo) * R2_HP * 4), (uint32_t)((sy1 - ylo) * R2_HP * 4), (bw & 0xFFFFu) << 12, (bw >> 16) << 12);
        }
    };
    int ylo, nrows;
    uint32_t v[NV];
    issue(uyI * NP, ylo, nrows, v);
    if (tid < RS_TW) {
        const int x = min(bx0 + tid, P.dw - 1);
        cxs[tid] = xs[x]; cxw[tid] = xw[x];
    }
    row_table(uyI * NP * R2_TH, ylo);
    commit(nrows, v);
    __syncthreads();
    // H pass set-up (the same for every tile of the column): thread = 4 adjacent destination columns of one source row, 16 rows per pass
    const int x0 = (tid & 15) * 4;
    const int c0 = cxs[x0], o0 = c0 - xal;
    const uint32_t sh = (uint32_t)(o0 & 3);
    uint32_t sel[4]; u16x2 aw[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t off = (uint32_t)(cxs[x0 + j] - c0);   // 0..4 for scale <= 1.3: both bytes of every pair lie inside the 8-byte window
        sel[j] = off | ((off + 1u) << 16) | 0x0c000c00u;
        aw[j] = __builtin_bit_cast(u16x2, (uint32_t)cxw[x0 + j]);   // weights are in [0, 2048]
    }
    const uint32_t* rowp = (const uint32_t*)(tile + (o0 & ~3));
    const int xg = tid & 15;
    const int dx0 = bx0 + xg * 4;
#pragma unroll
    for (int half = 0; half < NP; half++) {
        const int tyI = uyI * NP + half, by0 = tyI * R2_TH;
        const bool more = half + 1 < NP && tyI + 1 < P.tilesY;
        int ylo2 = 0, nrows2 = 0;
        uint32_t v2[NV];
        if (more) issue(tyI + 1, ylo2, nrows2, v2);       // in flight during this tile's H pass
        for (int r = tid >> 4; r < nrows; r += 16) {
            const uint32_t* d = (const uint32_t*)((const uint8_t*)rowp + r * RS_PITCH);
            const uint32_t d0 = d[0], d1 = d[1], d2 = d[2];
            const uint32_t W0 = __builtin_amdgcn_alignbyte(d1, d0, sh), W1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
            uint4 T;
            uint32_t* Tp = (uint32_t*)&T;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t pp = __builtin_amdgcn_perm(W1, W0, sel[j]);
                const uint32_t t = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pp), aw[j], 0u, false);
                Tp[j] = t & ~15u;   // 16 * (t >> 4)
            }
            *(uint4*)(hbuf + r * R2_HP + x0) = T;
        }
        __syncthreads();
        if (more) commit(nrows2, v2);                     // the staged tile is free: every thread is through its H pass
        if (dx0 < P.dw) {
            uint8_t* D = P.dst + (size_t)frameZ * P.dFrame + (size_t)(by0 + (tid >> 4)) * P.dStride + dx0;
#pragma unroll
            for (int k = 0; k < R2_TH / 16; k++) {
                const int ty = (tid >> 4) + 16 * k;
                if (by0 + ty >= P.dh) break;
                const uint4 rt = rowt[ty];
                // both factors are below 2^24 (weights << 12 <= 2^23, H-pass sums < 2^19); saying so on BOTH lets the compiler pick the full-rate
                // v_mul_hi_u32_u24 — with only one side masked it emitted v_mul_hi_u32 plus the sixteen v_and of the masks
                const uint32_t b0 = rt.z & 0xFFFFFFu, b1 = rt.w & 0xFFFFFFu;
                const uint4 T0 = *(const uint4*)((const uint8_t*)hbuf + rt.x + xg * 16);
                const uint4 T1 = *(const uint4*)((const uint8_t*)hbuf + rt.y + xg * 16);
#define R2_PIX(t0, t1) (((uint32_t)(((uint64_t)b0 * ((t0) & 0xFFFFFFu)) >> 32) + (uint32_t)(((uint64_t)b1 * ((t1) & 0xFFFFFFu)) >> 32) + 2u) >> 2)
                // a pixel is <= 255 without a clamp: the two weights add up to 2048 (+-1), the H sums are <= 255 * 2048, so the two products sum to <= 1020
                const uint32_t out = R2_PIX(T0.x, T1.x) | (R2_PIX(T0.y, T1.y) << 8) | (R2_PIX(T0.z, T1.z) << 16) | (R2_PIX(T0.w, T1.w) << 24);
#undef R2_PIX
                uint8_t* Dk = D + (size_t)(16 * k) * P.dStride;
                if (dx0 + 3 < P.dw) {
                    *(uint32_t*)Dk = out;  // dStride and dx0 are multiples of 4
                } else {
                    for (int j = 0; j < 4 && dx0 + j < P.dw; j++) Dk[j] = (uint8_t)(out >> (8 * j));
                }
            }
        }
        if (!more) break;
        __syncthreads();                                  // the H buffer and the row table are free; the next tile is staged
        ylo = ylo2; nrows = nrows2;
        row_table((tyI + 1) * R2_TH, ylo);                // (read behind the next H pass's barrier)
    }
}

// mvImagePyramid in the reference's own layout (ORBextractor.cc:1164-1179): every level inside a 19-px BORDER_REFLECT_101 frame
// (copyMakeBorder), all levels of ONE frame in one slab — the host-pyramid option of the single-frame entry point brings that slab down in one
// copy.  One workgroup per bordered row of any level (blockIdx.x walks the rows of level 0, then level 1, ...); a thread writes dwords.
struct BorderParams {
    const uint8_t* src[ORBX_MAX_LEVELS]; int sStride[ORBX_MAX_LEVELS]; int w[ORBX_MAX_LEVELS]; int h[ORBX_MAX_LEVELS];
    uint8_t* dst; size_t dOff[ORBX_MAX_LEVELS]; int dPitch[ORBX_MAX_LEVELS];   // dPitch: multiple of 4, >= w + 2 * ORBX_EDGE
    int rowStart[ORBX_MAX_LEVELS + 1];                                         // first bordered row of each level in the grid
    int nlevels;
};
static __device__ __forceinline__ int reflect101(int p, int len) {   // BORDER_REFLECT_101 for |overhang| < len
    if (p < 0) p = -p;
    if (p >= len) p = 2 * (len - 1) - p;
    return p;
}
static __global__ __launch_bounds__(256) void k_border_pyramid(BorderParams P) {
    int l = 0;
    while (l + 1 < P.nlevels && (int)blockIdx.x >= P.rowStart[l + 1]) l++;
    const int y = (int)blockIdx.x - P.rowStart[l], w = P.w[l], ow = w + 2 * ORBX_EDGE;
    const uint8_t* S = P.src[l] + (size_t)reflect101(y - ORBX_EDGE, P.h[l]) * P.sStride[l];
    uint8_t* D = P.dst + P.dOff[l] + (size_t)y * P.dPitch[l];
    for (int x4 = threadIdx.x * 4; x4 < ow; x4 += 1024) {
        uint32_t v = 0;
        for (int j = 0; j < 4; j++) v |= (uint32_t)S[reflect101(min(x4 + j, ow - 1) - ORBX_EDGE, w)] << (8 * j);
        *(uint32_t*)(D + x4) = v;   // (the dwords past ow lie inside the pitch)
    }
}

// Longest work first.  k_fast and k_octree run one workgroup per (frame, tile) / (frame, level) and the hardware hands workgroups out in grid order:
// a frame that costs ten times the others (every pixel a corner; a realistic mix has such frames next to flat ones) and sits at the END of the batch
// leaves its long workgroups running alone after everything else has finished — on the mixed batch of bench.py that tail was a third of k_fast's
// time and two thirds of k_octree's.  The handle's previous call left its FAST candidate counts per (frame, level) in candCount: frames are served
// in descending order of that count (rank sort in LDS, ties by index: a permutation whatever the counts hold — stale or uninitialised counts only
// cost the ordering its benefit).  A camera slot of a rig or a video stream keeps its character from call to call, which is all this relies on; the
// key points never depend on it.  The same workgroup then clears the counters for the call that follows (it replaces that call's memset launch).
#define ORDER_T 1024
#define ORDER_MAX_BATCH 4096
// (any workgroup size; LDS: batch + 4 words.)  retry != null: the FAST second-pass list of the call that follows starts empty — [0] = 0, [1] = its tile count
static __device__ __forceinline__ void frame_order_body(unsigned char* smem, int* candCount, const int nlevels, const int batch, int* order, uint32_t* hostMax,
                                                        uint32_t* retry, const uint32_t retryTiles) {
    // keys[b] = count << 12 | (4095 - b): descending order of the packed word = descending count, ties by ascending frame (batch <= ORDER_MAX_BATCH = 4096;
    // counts above 2^20 - 1 saturate — a 752 x 480 frame holds < 500 000 candidates — which can only cost such frames their mutual order)
    uint32_t* keys = (uint32_t*)smem;   // [batch rounded up to 4 (padded with 0: below every real key)], then one word: the largest count of any (frame, level)
    const int tid = threadIdx.x, NT = blockDim.x, b4 = (batch + 3) & ~3;
    if (tid == 0) { keys[b4] = 0; if (retry) { retry[0] = 0; retry[1] = retryTiles; } }
    for (int i = batch + tid; i < b4; i += NT) keys[i] = 0;
    __syncthreads();
    uint32_t mx = 0;
    for (int b = tid; b < batch; b += NT) {
        uint32_t s = 0;
        for (int l = 0; l < nlevels; l++) { const uint32_t c = (uint32_t)candCount[(size_t)b * nlevels + l]; s += c; mx = max(mx, c); }
        keys[b] = (min(s, 0xFFFFFu) << 12) | (uint32_t)(4095 - b);
    }
    if (mx) atomicMax(&keys[b4], mx);
    __syncthreads();
    if (tid == 0) *hostMax = keys[b4];   // pinned host word: the NEXT call's launch plan reads it (is a 1 024-thread octree pass worth launching?)
    // rank = keys above mine; the workgroup rides a launch whose other workgroups share its compute unit, so the loop is kept short: four keys per LDS read
    for (int b = tid; b < batch; b += NT) {
        const uint32_t k = keys[b];
        int r = 0;
        for (int j = 0; j < b4; j += 4) {
            const uint4 q = *(const uint4*)(keys + j);
            r += (q.x > k ? 1 : 0) + (q.y > k ? 1 : 0) + (q.z > k ? 1 : 0) + (q.w > k ? 1 : 0);
        }
        order[r] = b;
    }
    for (int i = tid; i < batch * nlevels; i += NT) candCount[i] = 0;   // (every key was formed before the barrier above)
}
// A launch of its own only where the pyramid's first level does not take the separable kernel (scale factors above 1.3): otherwise the job rides as one
// extra workgroup of the level-1 k_resize2 launch — a dependent 19-us launch, two memset launches and (k_octree) an 8-byte copy launch less per call
static __global__ __launch_bounds__(ORDER_T) void k_frame_order(int* candCount, int nlevels, int batch, int* order, uint32_t* hostMax) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    frame_order_body(orb_smem, candCount, nlevels, batch, order, hostMax, nullptr, 0u);
}

// ============================================================================================================
// E2  FAST-9/16 + per-cell NMS + per-cell minThFAST retry
// ============================================================================================================
struct FastTile { short level, cellRow, cell0, nCells; };
struct FastLevel {
    const uint8_t* base; size_t frameStride; int rowStride;
    int w, h, nCols, nRows, wCell, hCell;
    int wCellMagic;   // ceil(65536 / wCell): column / wCell == (column * wCellMagic) >> 16 for column < 256
    size_t candOff;   // offset (u32 units) of this level's candidate slab inside one frame's block
    int candCap;
};
struct FastParams {
    FastLevel lv[ORBX_MAX_LEVELS];
    const FastTile* tiles;
    uint32_t* cand; size_t candFrame;   // [frame][candFrame] u32: x | y<<12 | score<<24, coordinates relative to minBorder
    int* candCount; int nlevels;        // [frame][nlevels]
    int iniTh, minTh;
    int imgBytes;                       // LDS bytes reserved for the image tile (== score-map bytes)
    int nTiles, batch;                  // tiles per frame, frames: the XCD-aware 1-D grid
    uint32_t* retry;                    // [0] tiles listed, then {tile | frame << 16, mask of its cells to detect again: bit 16 * cell row + cell}
    const int* order;                   // frame of grid row y (k_frame_order: heaviest frames of the handle's previous call first), or nullptr = y
};

#define RING16(F)                                                                                         \
    F(0, 0, 3) F(1, 1, 3) F(2, 2, 2) F(3, 3, 1) F(4, 3, 0) F(5, 3, -1) F(6, 2, -2) F(7, 1, -3)             \
    F(8, 0, -3) F(9, -1, -3) F(10, -2, -2) F(11, -3, -1) F(12, -3, 0) F(13, -3, 1) F(14, -2, 2) F(15, -1, 3)

// S = max over the sixteen 9-arcs of min(v - x) and of min(x - v): the largest threshold t for which the pixel
// is still a FAST-9 corner is S-1 (== cv cornerScore<16>), and it is a corner at threshold t iff S > t.
static __device__ __forceinline__ int fast_S(const uint8_t* c, int pitch) {
    const int v = c[0];
    int d[16];
#define LD(k, dx, dy) d[k] = v - (int)c[(dy) * pitch + (dx)];
    RING16(LD)
#undef LD
    int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { lo2[i] = min(d[i], d[(i + 1) & 15]); hi2[i] = max(d[i], d[(i + 1) & 15]); }
#pragma unroll
    for (int i = 0; i < 16; i++) { lo4[i] = min(lo2[i], lo2[(i + 2) & 15]); hi4[i] = max(hi2[i], hi2[(i + 2) & 15]); }
    int A = -256, B = 256;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int lo9 = min(min(lo4[i], lo4[(i + 4) & 15]), d[(i + 8) & 15]);
        const int hi9 = max(max(hi4[i], hi4[(i + 4) & 15]), d[(i + 8) & 15]);
        A = max(A, lo9);
        B = min(B, hi9);
    }
    return max(A, -B);
}

// Two 16-bit lanes per register.  Device builds: clang vector types (v_pk_min_u16 / v_pk_max_u16, the half swap is the op_sel every packed
// instruction has for free); the emulator: the same operations written out, so that the CPU tier runs fast_S_pk's logic as it stands.
#ifdef HIP_EMULATED
struct pk16 { uint16_t x, y; };
static inline pk16 pk_from(uint32_t u) { return pk16{(uint16_t)u, (uint16_t)(u >> 16)}; }
static inline pk16 pk_min(pk16 a, pk16 b) { return pk16{a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y}; }
static inline pk16 pk_max(pk16 a, pk16 b) { return pk16{a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y}; }
static inline pk16 pk_swp(pk16 a) { return pk16{a.y, a.x}; }
#else
typedef unsigned short pk16 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ pk16 pk_from(uint32_t u) { return __builtin_bit_cast(pk16, u); }
static __device__ __forceinline__ pk16 pk_min(pk16 a, pk16 b) { return __builtin_elementwise_min(a, b); }
static __device__ __forceinline__ pk16 pk_max(pk16 a, pk16 b) { return __builtin_elementwise_max(a, b); }
static __device__ __forceinline__ pk16 pk_swp(pk16 a) { return __builtin_shufflevector(a, a, 1, 0); }
#endif

// fast_S for the pixels that are candidates: S exact where S > 0, some value <= 0 where fast_S is <= 0 (nobody looks at those: thresholds are >= 0).
// Ring positions (i, i + 8) share a register.  ONE polarity is evaluated, the only one that can hold a 9-arc: G = 255 + x - v per 16-bit lane
// (one 32-bit add on the pair: no carry leaves a lane) has (x > v) as its second byte, a v_dot4 per register counts them, and a 9-arc of brighter
// ring pixels needs nine of them — with fewer, only a darker arc can exist (both polarities would need 18 ring pixels), and 511 - G = G ^ 0x1FF =
// 256 + v - x is that polarity on the same min network: max over the sixteen arcs of the minimum of nine.  ~70 instructions (both polarities on the
// doubling network of rounds 2-3: ~100).
static __device__ __forceinline__ int fast_S_pk(const uint8_t* c, const int pitch) {
    uint32_t x[16];
#define LD(k, dx, dy) x[k] = c[(dy) * pitch + (dx)];
    RING16(LD)
#undef LD
    const uint32_t C = (255u - (uint32_t)c[0]) * 0x00010001u;
    uint32_t G[8], nB = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        G[i] = (x[i] | (x[i + 8] << 16)) + C;
        nB = __builtin_amdgcn_udot4(G[i], 0x01000100u, nB, false);
    }
    const bool bright = nB >= 9;
    const uint32_t flip = bright ? 0u : 0x01FF01FFu;
    // The sixteen arc minima by running minima (van Herk / Gil-Werman): the low lanes are ring positions 0-7, the high lanes 8-15, and the arc that
    // starts at position j is [j..7] of its own block (a suffix minimum) followed by [0..j] of the other block (a prefix minimum, half-swapped):
    // 7 + 7 + 8 packed minima and 7 maxima (the doubling network L2 / L4 / L8 + one took 32 + 8).
    pk16 pre[8], suf[8];
    pre[0] = pk_from(G[0] ^ flip); suf[7] = pk_from(G[7] ^ flip);
#pragma unroll
    for (int i = 1; i < 8; i++) pre[i] = pk_min(pre[i - 1], pk_from(G[i] ^ flip));
#pragma unroll
    for (int i = 6; i >= 0; i--) suf[i] = pk_min(suf[i + 1], pk_from(G[i] ^ flip));
    pk16 A = pk_min(suf[0], pk_swp(pre[0]));
#pragma unroll
    for (int i = 1; i < 8; i++) A = pk_max(A, pk_min(suf[i], pk_swp(pre[i])));
    return (int)max((uint32_t)A.x, (uint32_t)A.y) - (bright ? 255 : 256);
}

// inclusive prefix sum over the 64 lanes of a wave, register-only: four row_shr steps inside each row of 16 lanes, then the two DPP row
// broadcasts gfx9 has for exactly this purpose (row 1/3 += lane 15 of the row before, rows 2-3 += lane 31).  No LDS round trips (a
// __shfl_up scan is seven dependent ds_bpermute).  All 64 lanes must be active.
static __device__ __forceinline__ int wave_scan_incl(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return x;
}
// wave-wide sum / minimum through the same DPP steps (the result forms in lane 63 and is broadcast with one v_readlane): seven register-only
// instructions where a __shfl_xor butterfly is six dependent ds_bpermute round trips.  All 64 lanes must be active.
static __device__ __forceinline__ int wave_sum(int x) { return __builtin_amdgcn_readlane(wave_scan_incl(x), 63); }
// the same on unsigned words (two packed 16-bit sums per register: the adds must wrap, not overflow a signed int)
static __device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
static __device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
#define WMIN_STEP(ctrl, rows) x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, ctrl, rows, 0xF, false))
    WMIN_STEP(0x111, 0xF); WMIN_STEP(0x112, 0xF); WMIN_STEP(0x114, 0xF); WMIN_STEP(0x118, 0xF);   // row_shr:1, 2, 4, 8
    WMIN_STEP(0x142, 0xA); WMIN_STEP(0x143, 0xC);                                                 // row_bcast:15 -> rows 1, 3; row_bcast:31 -> rows 2, 3
#undef WMIN_STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// LDS layout of k_fast (dynamic region), sizes fixed per handle:
//   img[imgBytes] | q1[4 * FAST_Q1W] u16 | q2[FAST_Q2CAP] u16 | colTab[FAST_TW] u8 | sh[8 + FAST_MAXCELLS] int
// q1: pixels that passed the pre-test; q2: every corner (S > min(ini,min)) of the tile.
// Queue entries are (row << 8 | col) inside the tile's detection region (<= 64 rows x <= 256 cols); q2 bit 15 = local maximum.
// The score map of the NMS is the img region itself: the image tile is dead once the last corner is scored, so a wave parks its scores in a byte
// list (its own q1 slice, idle by then) until every wave is through, and they are scattered into the cleared region — with one blank map row
// between the tile's cell rows, so that NMS never compares across a cell boundary.  (A second tile-sized array next to img would cost the
// two-cell-row tiles their eighth workgroup per CU.)  The emit list reuses q1 (FAST_QCAP/2 u32 entries).
#ifndef FAST_Q2CAP
#define FAST_Q2CAP 2176   // corners per tile kept in LDS (544 per wave: 23 % of the pixels of a two-cell-row tile, 46 % of a one-row tile; the benchmark's
                          // densest tile has 6 %); more -> whole-tile fallback (tests build with a tiny value to cover it).  2176: at 752x480 the kernel's
                          // LDS is then 19 904 B = 8 workgroups per CU (with 3584 entries and 7 workgroups 0.81 instead of 0.79 ms)
#endif
#ifndef FAST_XCD
#define FAST_XCD 0   // 1: frame-per-XCD mapping (xcd_frame_unit) for k_fast too.  Measured on MI355X: 1.162 ms vs 1.137 ms with the plain (tile, frame)
                     // grid at batch 512, equal at batch 64 — k_fast's staging is not L2-fill bound (tiles overlap by 6 px only), while k_describe's
                     // 43x48-byte patches overlap heavily and gain 2-3 % from it.  Kept switchable.
#endif
#define FAST_TW 128               // detection columns per tile (threads 0..127 / 128..255 take alternate rows)
#ifndef FAST_PITCH
#define FAST_PITCH 148            // LDS row pitch of every tile: 128 detection columns + 6 (ROI overlap) + 4 (dword alignment of the detection
#endif                            // region) = 138 -> 144 staged bytes, plus one dword: 37 dwords per row.  A compile-time constant, so the ring
                                  // rows of pre-test, classification and score are instruction offsets; and ODD in dwords, so that the
                                  // survivors of a vertical edge (one column, consecutive rows) spread over all 32 banks in the byte gathers of
                                  // stage 2 / 3 (36 dwords put them on 8 banks: 46 % of the kernel's LDS cycles were bank conflicts, 26 % now;
                                  // LDS busy 78 % -> 59 %, 0.885 -> 0.857 ms).  Rows are then only 4-byte aligned: staged with dword stores.
#ifndef FAST_CROWS
#define FAST_CROWS 2              // cell rows per tile where two of them fit the 64 detection rows of the pre-test's row mask (levels 0-3 at 752x480: 81 % of the
#endif                            // pixels): prologue, staging set-up, barriers and the partly filled batches of score / NMS / emit are paid once per tile, the
                                  // 6-row halo is shared — 5.4 % fewer VALU instructions per launch, 0.847 -> 0.792 ms per 512 frames at 8 workgroups per CU
#ifndef FAST_TALL_MIN_BATCH
#define FAST_TALL_MIN_BATCH 8     // frames per call from which the two-cell-row tiles are used
#endif
#ifndef FAST_TWO_PASS
#define FAST_TWO_PASS 1   // 1: batches detect at iniThFAST first and only the tiles with an empty cell again at min(ini, min) (k_fast, second launch)
#endif
#ifndef FAST_TWO_PASS_MIN_BATCH
#define FAST_TWO_PASS_MIN_BATCH 64   // frames per call from which the two-pass form is considered
#endif
#ifndef FAST_TWO_PASS_MAX_LISTED
#define FAST_TWO_PASS_MAX_LISTED 0.22   // listed share of the tiles up to which the two passes are kept (measured: -9 % of k_fast at 0.14, +15 % at 0.42)
#endif
#ifndef FAST_PROBE_EVERY
#define FAST_PROBE_EVERY 16         // calls between two-pass probes while a handle runs one pass
#endif
#ifndef FAST_FITS_MAX
#define FAST_FITS_MAX (FAST_QCAP / 4 + 64)   // survivors a wave queues in one go (tests build with a small value to send ordinary tiles row group by row group)
#endif
#define FAST_Q1W (FAST_QCAP / 4 + 64)               // a wave's q1 slice (576 entries): all of its pre-test survivors, or one row group of them (<= 256)

// Stage-1 thresholds of k_fast on D = (x + 255 - v) >> 1 (one v_lerp_u8 per ring position: four pixels per instruction).  x - v > t implies
// D >= (t + 256) >> 1 and v - x > t implies D <= (254 - t) >> 1 (>> is monotone), so both compares are necessary conditions — at most one grey
// level weaker than the exact ones, which only the exact stage 2 decides.  A byte compare D >= th is bit 7 of (D + 256 - th) >> 1.
static __device__ __forceinline__ void fast_pretest_consts(const int t0, uint32_t* KB, uint32_t* KG) {
    const int tp = min(t0, 254);
    *KB = 0x01010101u * (uint32_t)(256 - ((tp + 256) >> 1));   // bit 7 <=> D >= (t + 256) >> 1   (may be brighter than v + t)
    *KG = 0x01010101u * (uint32_t)(255 - ((254 - tp) >> 1));    // bit 7 <=> D >  (254 - t) >> 1   (can NOT be darker than v - t)
}
// The pre-test of four adjacent pixels (one dword of an LDS row of pitch FAST_PITCH): bit 7 of byte t of the result is set if pixel t MAY be a FAST-9
// corner at the threshold the constants were made for — never clear for a pixel that is one (tests/cpp/fast_score_test.cpp checks exactly that).
// cw = the dword LEFT of the centre dword, three rows up.
static __device__ __forceinline__ uint32_t fast_pretest4(const uint32_t* cw, const uint32_t KB, const uint32_t KG) {
    constexpr int p4 = FAST_PITCH >> 2;
    const uint32_t U3 = cw[1], A0 = cw[p4], A1 = cw[p4 + 1], A2 = cw[p4 + 2];                // rows -3, -2
    const uint32_t Cp = cw[3 * p4], C = cw[3 * p4 + 1], Cn = cw[3 * p4 + 2];                 // row 0
    const uint32_t B0 = cw[5 * p4], B1 = cw[5 * p4 + 1], B2 = cw[5 * p4 + 2], D3 = cw[6 * p4 + 1];   // rows +2, +3
    const uint32_t nV = ~C;
    uint32_t accB, accG;   // bit 7 of a byte: every pair so far has a member that may be brighter / a pair so far has no member that may be darker
#define PAIR(first, xa, xb) {                                                                                                               \
        const uint32_t Da = __builtin_amdgcn_lerp(xa, nV, 0u), Db = __builtin_amdgcn_lerp(xb, nV, 0u);                              \
        const uint32_t b = __builtin_amdgcn_lerp(Da, KB, 0u) | __builtin_amdgcn_lerp(Db, KB, 0u);                                   \
        const uint32_t g = __builtin_amdgcn_lerp(Da, KG, 0u) & __builtin_amdgcn_lerp(Db, KG, 0u);                                   \
        if (first) { accB = b; accG = g; } else { accB &= b; accG |= g; } }
    PAIR(true, D3, U3)                                                                                      // ring 0 (0, 3) and 8 (0, -3)
    PAIR(false, __builtin_amdgcn_alignbyte(Cn, C, 3), __builtin_amdgcn_alignbyte(C, Cp, 1))                 // 4 (3, 0) and 12 (-3, 0)
    PAIR(false, __builtin_amdgcn_alignbyte(B2, B1, 2), __builtin_amdgcn_alignbyte(A1, A0, 2))               // 2 (2, 2) and 10 (-2, -2)
    PAIR(false, __builtin_amdgcn_alignbyte(A2, A1, 2), __builtin_amdgcn_alignbyte(B1, B0, 2))               // 6 (2, -2) and 14 (-2, 2)
#undef PAIR
    return (accB | ~accG) & 0x80808080u;
}

// a * b for operands below 2^24: ONE full-rate v_mul_u32_u24 (the compiler cannot prove the ranges of row indices, strides and table
// multipliers and emits the quarter-rate v_mul_lo_u32)
static __device__ __forceinline__ uint32_t mul24(const uint32_t a, const uint32_t b) {
#ifdef HIP_EMULATED
    return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);
#else
    return __umul24(a, b);
#endif
}

// signed form (v_mul_i32_i24 / v_mad_i32_i24)
static __device__ __forceinline__ int imul24(const int a, const int b) {   // the low 24 bits of both, sign-extended
#ifdef HIP_EMULATED
    return (int)((uint32_t)(((int)((uint32_t)a << 8)) >> 8) * (uint32_t)(((int)((uint32_t)b << 8)) >> 8));
#else
    return __mul24(a, b);
#endif
}
// cvRound / lrintf for |x| < 2^22 by the float adder: x + 1.5 * 2^23 is rounded to an integer (ties to even, like v_rndne_f32) that sits in the low
// mantissa bits — rint_bits(x) = RINT_BIAS + rint(x) as an integer, one full-rate v_add_f32 where v_rndne_f32 + v_cvt_i32_f32 are two half-rate
// instructions; users fold RINT_BIAS into a constant they add anyway
#define RINT_BIAS 0x4B400000u
static __device__ __forceinline__ uint32_t rint_bits(const float x) { return __float_as_uint(x + 12582912.0f); }

static __device__ __forceinline__ int wave_append(bool pass, int* counter, int lane) {
    // ordered-within-wave append: returns the slot for passing lanes (one LDS atomic per wave)
    const unsigned long long m = __ballot(pass);
    int base = 0;
    if (m) {
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) base = atomicAdd(counter, __popcll(m));
        base = __shfl(base, leader);
    }
    return base + __popcll(m & ((1ull << lane) - 1ull));
}

// One tile.  listId (PASS 1): the tile's entry in the retry list.  Every `return` in here is taken by the whole workgroup (the conditions are
// workgroup-uniform), so k_fast<1> may call it in a loop with a barrier between the tiles.
template <int PASS>   // 0: first pass at iniThFAST, 1: the listed tiles at min(ini, min), 2: one pass at min(ini, min) (see below)
static __device__ __forceinline__ void fast_tile(const FastParams& P, const int listId) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    PROF_DECL;
    // the tile record is workgroup-uniform: read it through the scalar path (constant address space) — as a plain global load it is a
    // vector-memory round trip of its own in front of everything else
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIP_EMULATED)
    static_assert(sizeof(FastTile) == 8, "FastTile is read as one 64-bit scalar");
#define FAST_TILE_AT(i) __builtin_bit_cast(FastTile, *(const unsigned long long __attribute__((address_space(4)))*)(unsigned long long)(P.tiles + (i)))
#else
#define FAST_TILE_AT(i) (P.tiles[i])
#endif
#if FAST_XCD
    int frame, tileIdx;
    if (!xcd_frame_unit(P.nTiles, P.batch, &frame, &tileIdx)) return;
    const FastTile T = FAST_TILE_AT(tileIdx);
    const uint32_t emitMask = 0xFFFFFFFFu;   // (one pass only in this mapping)
#else
    // Two thresholds, two launches (PASS).  The reference runs cv::FAST at iniThFAST on every cell and again at minThFAST only on the cells that came
    // back empty (ORBextractor.cc:812-828).  Detecting at the lower threshold and filtering by score gives the same key points ("FAST as set
    // algebra") but scores every corner between the two thresholds for nothing wherever iniThFAST finds something: on the benchmark's frames 98 % of
    // the cells, while 35 % of the pre-test's survivors and 41 % of the corners at 7 are below 20.  Pass 0 therefore detects at iniThFAST and emits what it
    // finds — final for every cell that holds a local maximum — and a tile with cells that hold none (flat ones; a third of the benchmark's level-0
    // tiles have one) lists itself with their mask; a tile whose corner list overflows lists all of its cells and emits nothing.  Pass 1 (the same grid,
    // launched behind pass 0) takes a listed tile at min(ini, min) exactly as the one-pass form (pass 2) does — as a one-cell tile if one cell is
    // asked for — and reports the listed cells only.
    int tileIdx = blockIdx.x, frame = P.order ? P.order[blockIdx.y] : (int)blockIdx.y;
    FastTile T;
    uint32_t emitMask = 0xFFFFFFFFu;   // cells (bit 16 * cell row + cell) this workgroup reports
    if (PASS == 1) {
        // The retry list is read through the scalar cache, like the tile record: most workgroups of this launch only find out that they have nothing
        // to do, and each holds its slot for as long as that takes.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIP_EMULATED)
        const uint32_t __attribute__((address_space(4)))* rl = (const uint32_t __attribute__((address_space(4)))*)(unsigned long long)P.retry;
#else
        const uint32_t* rl = P.retry;
#endif
        const int id = listId;                           // (< rl[0]: k_fast<1>'s loop)
        const uint32_t e = rl[2 + 2 * id];
        emitMask = rl[3 + 2 * id];
        tileIdx = (int)(e & 0xFFFFu); frame = (int)(e >> 16);
        T = FAST_TILE_AT(tileIdx);
        if (emitMask != 0xFFFFFFFFu) {   // (all ones: the first pass's corner list overflowed, the whole tile again)
            // the tile shrinks to the bounding box of the listed cells — one cell in the usual case —, and inside the box the pre-test below looks at
            // the listed cells' columns only: what the second pass costs follows the area that came back empty, not the tile
            uint32_t r0 = emitMask & 0xFFFFu, r1 = emitMask >> 16;
            int rows = (int)T.level >> 8;
            if (r1 == 0) rows = 1;
            else if (r0 == 0) { T.cellRow = (short)(T.cellRow + 1); rows = 1; r0 = r1; r1 = 0; }
            const uint32_t u = r0 | r1;
            const int cmin = __ffs((int)u) - 1, cmax = 31 - __clz((int)u);
            T.cell0 = (short)(T.cell0 + cmin); T.nCells = (short)(cmax - cmin + 1);
            T.level = (short)((T.level & 0xFF) | (rows << 8));
            emitMask = (r0 >> cmin) | ((r1 >> cmin) << 16);
        }
    } else {
        T = FAST_TILE_AT(tileIdx);
    }
    frame = __builtin_amdgcn_readfirstlane(frame);
    emitMask = (uint32_t)__builtin_amdgcn_readfirstlane((int)emitMask);
#endif
#undef FAST_TILE_AT
    const int level = T.level & 0xFF, nCR = (int)T.level >> 8;   // the tile covers nCR whole cell rows, T.cellRow the first
    const FastLevel& L = P.lv[level];

    const int iniY = ORBX_MINB + T.cellRow * L.hCell;
    const int maxY = min(iniY + nCR * L.hCell + 6, L.h - ORBX_MINB);
    const int iniX = ORBX_MINB + T.cell0 * L.wCell;
    const int maxX = min(ORBX_MINB + (T.cell0 + T.nCells) * L.wCell + 6, L.w - ORBX_MINB);
    // LDS column 0 = image column iniX-1, so that the detection region starts at byte 4 of every LDS row: detection column c
    // lives in dword 1 + c/4, which lets stage 1 treat one dword = 4 pixels per lane (rows are staged with a byte shift).
    const int xal = iniX - 1;
    constexpr int pitch = FAST_PITCH;
    const int wbytes = ((maxX - xal) + 15) & ~15;   // bytes of a row that are actually staged
    const int rows = maxY - iniY;
    // detection region of the tile, local coordinates (cv::FAST skips a 3-px frame of each ROI; ROIs overlap by 6)
    const int dx0 = 4, dxe = maxX - 3 - xal;
    const int dy0 = 3, dye = rows - 3;
    const int detW = dxe - dx0, detH = dye - dy0;
    if (detW <= 0 || detH <= 0) return;

    uint8_t* img = orb_smem;
    uint8_t* smap = orb_smem;     // aliased onto the image tile once the last corner is scored
    uint16_t* q1 = (uint16_t*)(orb_smem + P.imgBytes);
    auto cellRowOf = [&](const int ry) {
        int cr = 0;
#pragma unroll
        for (int j = 1; j < FAST_CROWS; j++) cr += (j < nCR && ry >= j * L.hCell) ? 1 : 0;
        return cr;
    };
    uint32_t* elist = (uint32_t*)q1;                  // reused after the scoring phase
    uint16_t* q2 = q1 + 4 * FAST_Q1W;
    uint8_t* colTab = (uint8_t*)(q2 + FAST_Q2CAP);    // per detection column: cell | leftEdge<<6 | rightEdge<<7
    int* sh = (int*)(colTab + FAST_TW);                   // [1]=emit count [2]=emit base [4]=corner-list overflow [8..]=per-cell counts of local maxima >= iniTh

    {   // stage the tile, 16 bytes per lane and step: five consecutive aligned dwords of the row (the compiler merges them into dwordx4 +
        // dword) -> four funnel shifts -> one 128-bit LDS store.  ALL global loads of the tile are issued before the first one is consumed
        // (a plain `for` over the steps waits for each step's data before issuing the next: two to three memory round trips in series at
        // the head of a workgroup that lives ~14 us), and the score map is cleared while they are in flight.
        const uint32_t sh8 = (uint32_t)(xal & 3);
        const uint8_t* src = L.base + (size_t)frame * L.frameStride + (size_t)iniY * L.rowStride + (xal & ~3);
        const int gq = wbytes >> 4;                      // 16-byte groups staged per LDS row
        const int ng = rows * gq;
        const int safe = L.w - (xal & ~3);               // bytes of an image row that may be read from src
        const uint32_t gqm = c_div16[gq];                // gq <= FAST_PITCH / 16 = 9
        const int dr = (int)((256u * gqm) >> 16), dc = 256 - dr * gq;
        // Only the tile at the right image border can reach past the end of an image row.  The test is workgroup-uniform: everywhere else
        // a step is dwordx4 + dword with no per-lane branch (a per-lane branch makes the compiler drain the outstanding loads at its join —
        // both sides write the same registers); in the border tile a dword that would start past the row's end re-reads the row's last
        // dword instead (those columns are never tested).
        const bool edge = gq * 16 + 4 > safe;
        const int lastOff = (safe - 4) & ~3;
        // byte offsets inside the frame are 32-bit (src is workgroup-uniform): scalar base + vector offset addressing, no 64-bit VALU arithmetic
        auto fetch = [&](const int r, const int c, uint32_t* w) {
            const uint32_t o = mul24((uint32_t)r, (uint32_t)L.rowStride);
            if (!edge) {
                const uint32_t* gw = (const uint32_t*)(src + (o + 16u * (uint32_t)c));
                w[0] = gw[0]; w[1] = gw[1]; w[2] = gw[2]; w[3] = gw[3]; w[4] = gw[4];
            } else {
#pragma unroll
                for (int k = 0; k < 5; k++) w[k] = *(const uint32_t*)(src + (o + (uint32_t)min(16 * c + 4 * k, lastOff)));
            }
        };
        auto put = [&](const int r, const int c, const uint32_t* w) {
#if FAST_PITCH % 16 == 0
            ((uint4*)img)[r * (FAST_PITCH / 16) + c] = make_uint4(__builtin_amdgcn_alignbyte(w[1], w[0], sh8), __builtin_amdgcn_alignbyte(w[2], w[1], sh8),
                                                                 __builtin_amdgcn_alignbyte(w[3], w[2], sh8), __builtin_amdgcn_alignbyte(w[4], w[3], sh8));
#else
            uint32_t* dq = (uint32_t*)(img + mul24((uint32_t)r, FAST_PITCH)) + 4 * c;
            dq[0] = __builtin_amdgcn_alignbyte(w[1], w[0], sh8); dq[1] = __builtin_amdgcn_alignbyte(w[2], w[1], sh8);
            dq[2] = __builtin_amdgcn_alignbyte(w[3], w[2], sh8); dq[3] = __builtin_amdgcn_alignbyte(w[4], w[3], sh8);
#endif
        };
        constexpr int NS = 3;                            // steps held in registers: 768 groups = 85 rows of 144 bytes (a cell row is ~36 rows)
        uint32_t w[NS][5];
        int rr[NS], cc[NS];
        int r = (int)(mul24((uint32_t)tid, gqm) >> 16), c = tid - (int)mul24((uint32_t)r, (uint32_t)gq);
        // interior and right-border tiles take separate code copies of the request loop: merged behind one set of load instructions, the
        // interior case lost its dwordx4 + dword form (five single-dword loads with an address computation each)
        if (!edge) {
#pragma unroll
            for (int k = 0; k < NS; k++) {
                rr[k] = r; cc[k] = c;
                if (tid + 256 * k < ng) {
                    const uint32_t* gw = (const uint32_t*)(src + (mul24((uint32_t)r, (uint32_t)L.rowStride) + 16u * (uint32_t)c));
                    w[k][0] = gw[0]; w[k][1] = gw[1]; w[k][2] = gw[2]; w[k][3] = gw[3]; w[k][4] = gw[4];
                }
                r += dr; c += dc;
                if (c >= gq) { c -= gq; r++; }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NS; k++) {
                rr[k] = r; cc[k] = c;
                if (tid + 256 * k < ng) fetch(r, c, w[k]);
                r += dr; c += dc;
                if (c >= gq) { c -= gq; r++; }
            }
        }
#pragma unroll
        for (int k = 0; k < NS; k++)
            if (tid + 256 * k < ng) put(rr[k], cc[k], w[k]);
        for (int i = tid + 256 * NS; i < ng; i += 256) {   // taller tiles than any configuration in use: the plain loop for the rest
            uint32_t wx[5];
            fetch(r, c, wx);
            put(r, c, wx);
            r += dr; c += dc;
            if (c >= gq) { c -= gq; r++; }
        }
        if (tid < 8 + FAST_MAXCELLS) sh[tid] = 0;
        if (tid < detW) {
            const int cell = (int)(mul24((uint32_t)tid, (uint32_t)L.wCellMagic) >> 16);   // tid / L.wCell
            const int cellx0 = (int)mul24((uint32_t)cell, (uint32_t)L.wCell);
            const int cx = tid - cellx0, cw = min(L.wCell, detW - cellx0);
            colTab[tid] = (uint8_t)(cell | (cx == 0 ? 0x40 : 0) | (cx + 1 >= cw ? 0x80 : 0));
        }
    }
    PROF_MARK(0, 0);   // prologue + staging issue
    __syncthreads();
    PROF_MARK(0, 1);   // staging wait

    const int t0 = PASS == 0 ? P.iniTh : min(P.iniTh, P.minTh);
    // Stages 1-3 are wave-private: wave w owns the detection rows 2w, 2w+1 (mod 8), compacts its own survivors and corners into its own
    // slices of q1 / q2 and scores them itself -> no workgroup barrier and no LDS atomic until NMS.
    const int wave = tid >> 6;
    const int dcol = lane & 31, rsub = lane >> 5;       // stage 1: lane = one LDS dword (4 detection columns) of one row
    uint16_t* q1w = q1 + wave * FAST_Q1W;               // this wave's pre-test survivors
    uint16_t* q2w = q2 + wave * (FAST_Q2CAP / 4);       // this wave's corner list
    uint8_t* sbuf = (uint8_t*)q1w;                      // ... and the corners' scores, parked at the FRONT of the wave's own queue slice: entry i of the
                                                        // queue is read before the i-th score byte can be written (a batch reads its 64 entries, then
                                                        // writes at most 64 bytes below them), so the list eats its way into consumed entries only
    static_assert(FAST_Q2CAP / 4 <= 2 * FAST_Q1W, "a wave's scores are parked in its q1 slice");
    static_assert((FAST_Q2CAP / 4 + 1) / 2 + 256 <= FAST_Q1W && FAST_FITS_MAX <= FAST_Q1W, "a row group (<= 256 survivors) queues behind the parked scores");
    static_assert(4 * FAST_PITCH <= 4 * FAST_Q1W * 2, "the fallback's four-row score ring lives in the q1 region");
    static_assert(FAST_CROWS <= 2 && FAST_MAXCELLS / FAST_CROWS <= 16, "a tile's cells are bit 16 * cell row + cell of a 32-bit mask (retry list)");
    int n2w = 0;                                        // corners of this wave (wave-uniform)
    bool ovf = false;
    // stage 2 of one batch of 64 queue entries: the EXACT score decides (S > t <=> FAST-9 corner at threshold t, S - 1 is cv's score) and is kept —
    // on the benchmark's pyramid 22 % of the pixels pass the pre-test and 10 % are corners (a third of the pixels on the coarsest levels), so a
    // classification by ring bit masks followed by a score pass over the corners computed the ring differences of almost every second survivor
    // twice; the packed score costs little more than the classification did.
    auto classify = [&](const int i, const bool valid) {
        int ent = 0, S = 0;
        if (valid) {
            ent = q1w[i];
            S = fast_S_pk(img + (dy0 + (ent >> 8)) * pitch + dx0 + (ent & 255), pitch);
        }
        const bool corner = valid && S > t0;
        const unsigned long long mc = __ballot(corner);   // (also orders every lane's queue read before the score writes below: the emulator
        const int nc = __popcll(mc);                      // runs the lanes one after the other between such points)
        if (n2w + nc <= FAST_Q2CAP / 4) {
            if (corner) {
                const int slot = n2w + __popcll(mc & ((1ull << lane) - 1ull));
                q2w[slot] = (uint16_t)ent;
                sbuf[slot] = (uint8_t)(S - 1);            // S > t0 >= 0 here
            }
            n2w += nc;
        } else if (mc) ovf = true;
    };
    uint32_t KB, KG;   // (fast_pretest_consts: thresholds on the halved differences)
    fast_pretest_consts(t0, &KB, &KG);
    // ---- stage 1: pre-test on four antipodal pairs of the ring, {0,8} {2,10} {4,12} {6,14}, over ALL rows of the tile.  A 9-arc of the
    //      16-ring contains at least one member of EVERY antipodal pair {i, i+8}, so a corner has a brighter (> v+t) member in each of them,
    //      or a darker one in each (cv::FAST's own "high-speed test", fast.cpp, uses all eight pairs; the two compass pairs alone let 20 % of
    //      the benchmark's pixels through, these four 11 %, all eight 7.5 %, against 3 % corners).  Byte-parallel: one dword = 4 pixels per
    //      lane, every ring position is the same dword window shifted (v_alignbyte), every compare a v_lerp_u8 whose bit 7 per byte is the flag.
    //      Lane (rsub, dcol) of wave w takes the dword dcol of the rows 2w + rsub + 8k, k = 0 .. 7 (a tile has <= 64 detection rows).
    const int K = (detH + 7) >> 3;
    uint32_t mask = 0;   // bit 8*t + k: column 4*dcol + t of row 2*wave + rsub + 8k passed
    uint32_t colMask[FAST_CROWS];   // second pass: bit 7 of byte t = column 4*dcol + t belongs to a listed cell of that cell row
    if (PASS == 1) {
#pragma unroll
        for (int cr = 0; cr < FAST_CROWS; cr++) {
            colMask[cr] = 0;
#pragma unroll
            for (int t = 0; t < 4; t++)
                colMask[cr] |= (emitMask >> (16 * cr + (colTab[min(4 * dcol + t, detW - 1)] & 63)) & 1u) << (8 * t + 7);
        }
    }
    for (int k = 0; k < K; k++) {
        const int ry = 8 * k + 2 * wave + rsub;
        if (ry < detH && 4 * dcol < detW) {
            // base = the dword left of the centre dword, three rows up: every operand is a non-negative instruction offset (ds_read2_b32)
            const uint32_t* cw = (const uint32_t*)(img + ((dy0 - 3 + ry) & 0xFF) * pitch) + dcol;   // (& 0xFF: a 24-bit multiply)
            uint32_t bits = fast_pretest4(cw, KB, KG);
            if (PASS == 1) bits &= colMask[cellRowOf(ry)];
            const int nvalid = detW - 4 * dcol;                             // columns of this dword inside the detection region
            if (nvalid < 4) bits &= (1u << (8 * nvalid)) - 1u;
            mask |= bits >> (7 - k);
        }
    }
    // ---- compaction: ONE wave scan and one queue fill per tile (the survivors of all rows; ~11 % of the pixels), then stage 2 on batches of 64
    //      dense lanes.  If a wave's survivors do not fit its queue slice (more than half of its pixels pass: noise images), it goes row
    //      group by row group instead (a row group is at most 2 x 128 pixels).
    {
        const int total = __builtin_amdgcn_readlane(wave_scan_incl(__popc(mask)), 63);
        const bool fits = total <= FAST_FITS_MAX;
        for (int kp = 0; kp < (fits ? 1 : K); kp++) {
            uint32_t m = fits ? mask : mask & (0x01010101u << kp);
            const int cnt = __popc(m);
            const int incl = wave_scan_incl(cnt);
            const int n1 = __builtin_amdgcn_readlane(incl, 63);
            const int qb = fits ? 0 : (n2w + 1) >> 1;                           // a later row group queues behind the score bytes parked so far
            int slot = qb + incl - cnt;
            const int ent0 = ((2 * wave + rsub) << 8) | (4 * dcol);           // the entry of column t = 0 in the lane's first row (k = 0)
            while (m) {
                const int j = __ffs((int)m) - 1;                               // bit 8t + k
                m &= m - 1;
                q1w[slot++] = (uint16_t)(ent0 + ((j & 7) << 11) + (j >> 3));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int i0 = 0; i0 < n1; i0 += 64) classify(qb + i0 + lane, i0 + lane < n1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // q1w is refilled by the next row group
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    PROF_MARK(0, 2);   // stage 1 + compaction + stage 2
    if (ovf) sh[4] = 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PROF_MARK(0, 3);   // (stage 3 of earlier builds: the score is stage 2's by-product now)
    __syncthreads();
    PROF_MARK(0, 4);   // barrier: every wave is through with the image tile
    const bool overflow = sh[4] != 0;
    int* cellCnt = sh + 8;
    auto retry_later = [&](const uint32_t cells) {   // (thread 0) these cells again in pass 1
        const uint32_t i = atomicAdd(P.retry, 1u);
        P.retry[2 + 2 * i] = (uint32_t)tileIdx | ((uint32_t)frame << 16);
        P.retry[3 + 2 * i] = cells;
    };
    if (overflow && PASS == 0) { if (tid == 0) retry_later(0xFFFFFFFFu); PROF_FLUSH(0); return; }   // more corners at iniThFAST than the lists hold: the second pass has the fallback
    if (!overflow) {
        for (int i = tid; i < (P.imgBytes >> 4); i += 256) ((uint4*)smap)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        for (int i = lane; i < n2w; i += 64) {
            const int slot = i;
            const int ent = q2w[slot];
            const int ry = ent >> 8;
            smap[(dy0 + ry + cellRowOf(ry)) * pitch + dx0 + (ent & 255)] = sbuf[slot];
        }
        __syncthreads();
        // ---- NMS over the corner lists: strict maximum over the 8 neighbours inside the same cell's detection region.
        //      survive(T) = s >= T && localmax (threshold-independent localmax, DESIGN.md "FAST as set algebra").
        for (int i = lane; i < n2w; i += 64) {
            const int ent = q2w[i];
            const int rx = ent & 255, ryn = ent >> 8, crn = cellRowOf(ryn);
            const uint8_t* m = smap + (dy0 + ryn + crn) * pitch + dx0 + rx;
            const int s = m[0], ct = colTab[rx];
            bool ok = s > 0 && s > m[-pitch] && s > m[pitch];
            if (!(ct & 0x40)) ok = ok && s > m[-1] && s > m[-pitch - 1] && s > m[pitch - 1];
            if (!(ct & 0x80)) ok = ok && s > m[1] && s > m[-pitch + 1] && s > m[pitch + 1];
            if (ok) {
                q2w[i] = (uint16_t)(ent | 0x8000);
                if (s >= P.iniTh) atomicAdd(&cellCnt[crn * T.nCells + (ct & 63)], 1);
            }
        }
        __syncthreads();
        if (PASS == 0) {   // first pass: the cells (with pixels) that hold no local maximum — all of them are >= iniThFAST here — go on the list
            const int ncx = min((int)T.nCells, (int)(mul24((uint32_t)(detW + L.wCell - 1), (uint32_t)L.wCellMagic) >> 16));
            uint32_t emptyCells = 0;
            for (int cr = 0; cr < nCR; cr++)
                if (cr * L.hCell < detH)
                    for (int c = 0; c < ncx; c++) emptyCells |= cellCnt[cr * T.nCells + c] > 0 ? 0u : 1u << (16 * cr + c);
            if (emptyCells && tid == 0) retry_later(emptyCells);
        }
        for (int i = lane; i < n2w; i += 64) {
            const int ent = q2w[i];
            if (!(ent & 0x8000)) continue;
            const int rx = ent & 255, ry = (ent >> 8) & 127;
            const int cr2 = cellRowOf(ry);
            const int s = smap[(dy0 + ry + cr2) * pitch + dx0 + rx];
            const int cell2 = colTab[rx] & 63;
            const int Tth = cellCnt[cr2 * T.nCells + cell2] > 0 ? P.iniTh : P.minTh;   // per-cell retry, ORBextractor.cc:825-828
            if (s >= Tth && (emitMask >> (16 * cr2 + cell2) & 1u)) {
                // coordinates relative to minBorder, as vToDistributeKeys holds them (ORBextractor.cc:845-850)
                const uint32_t xr = (uint32_t)(xal + dx0 + rx - ORBX_MINB), yr = (uint32_t)(iniY + dy0 + ry - ORBX_MINB);
                const uint32_t packed = xr | (yr << 12) | ((uint32_t)s << 24);
                const int slot = atomicAdd(&sh[1], 1);
                if (slot < FAST_QCAP / 2) elist[slot] = packed;
                else {
                    const int g = atomicAdd(P.candCount + (size_t)frame * P.nlevels + level, 1);
                    if (g < L.candCap) P.cand[(size_t)frame * P.candFrame + L.candOff + g] = packed;
                }
            }
        }
        __syncthreads();
    } else {
        // ---- fallback (a wave met more corners than its list holds: noise images, thresholds near 0): score EVERY pixel of the tile, in
        //      place.  The tile has one LDS image and the score map takes its place, so the scores of a row wait in a four-row ring (the q1
        //      region) until the rows that still read its pixels (ring radius 3) are scored; then NMS and the per-cell retry over the whole
        //      map with explicit cell-row bounds, and every survivor goes straight to global memory.  Same results, no list bound; slow.
        uint8_t* ring = (uint8_t*)q1;                    // [4][pitch]
        for (int ry = 0; ry < detH + 3; ry++) {
            if (ry < detH && tid < detW) {
                const int S = fast_S(img + (dy0 + ry) * pitch + dx0 + tid, pitch);
                ring[(ry & 3) * pitch + tid] = (uint8_t)(S > t0 ? S - 1 : 0);
            }
            __syncthreads();
            const int rf = ry - 3;                       // rows <= rf + 3 are scored: nothing reads the pixels of row rf any more
            if (rf >= 0 && tid < detW) smap[(dy0 + rf) * pitch + dx0 + tid] = ring[(rf & 3) * pitch + tid];
            __syncthreads();
        }
        const int npix = detW * detH;
        for (int pass = 0; pass < 2; pass++) {
            for (int p = tid; p < npix; p += 256) {
                const int ry = p / detW, rx = p - ry * detW, cr = cellRowOf(ry);
                const uint8_t* m = smap + (dy0 + ry) * pitch + dx0 + rx;
                const int s = m[0], ct = colTab[rx];
                if (s == 0) continue;
                const int ryc = ry - cr * L.hCell;
                const bool top = ryc == 0, bot = ryc == L.hCell - 1 || ry == detH - 1, lft = (ct & 0x40) != 0, rgt = (ct & 0x80) != 0;
                bool ok = (lft || s > m[-1]) && (rgt || s > m[1]);
                if (!top) ok = ok && s > m[-pitch] && (lft || s > m[-pitch - 1]) && (rgt || s > m[-pitch + 1]);
                if (!bot) ok = ok && s > m[pitch] && (lft || s > m[pitch - 1]) && (rgt || s > m[pitch + 1]);
                if (!ok) continue;
                if (pass == 0) {
                    if (s >= P.iniTh) atomicAdd(&cellCnt[cr * T.nCells + (ct & 63)], 1);
                } else {
                    const int Tth = cellCnt[cr * T.nCells + (ct & 63)] > 0 ? P.iniTh : P.minTh;
                    if (s >= Tth && (emitMask >> (16 * cr + (ct & 63)) & 1u)) {
                        const uint32_t xr = (uint32_t)(xal + dx0 + rx - ORBX_MINB), yr = (uint32_t)(iniY + dy0 + ry - ORBX_MINB);
                        const int g = atomicAdd(P.candCount + (size_t)frame * P.nlevels + level, 1);
                        if (g < L.candCap) P.cand[(size_t)frame * P.candFrame + L.candOff + g] = xr | (yr << 12) | ((uint32_t)s << 24);
                    }
                }
            }
            __syncthreads();
        }
        PROF_FLUSH(0);
        return;
    }
    PROF_MARK(0, 5);   // NMS + per-cell retry + list
    const int ne = min(sh[1], FAST_QCAP / 2);
    if (ne == 0) { PROF_FLUSH(0); return; }
    if (tid == 0) sh[2] = atomicAdd(P.candCount + (size_t)frame * P.nlevels + level, ne);
    __syncthreads();
    const int gbase = sh[2];
    uint32_t* out = P.cand + (size_t)frame * P.candFrame + L.candOff;
    for (int i = tid; i < ne; i += 256)
        if (gbase + i < L.candCap) out[gbase + i] = elist[i];
    PROF_MARK(0, 6);   // emit
    PROF_FLUSH(0);
}
// Passes 0 and 2: one tile per workgroup, grid = (tiles, frames).  Pass 1: a 1-D grid sized by the host from the PREVIOUS call's list (a scheduling
// guess only: any size is correct) walks the retry list with a workgroup stride — launched on the full (tiles, frames) grid, seven of eight of its
// workgroups did nothing but read the list's length, each holding a slot with the kernel's 20 KB of LDS while it did.
template <int PASS>
static __global__ __launch_bounds__(256) void k_fast(FastParams P) {
    if (PASS == 1) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIP_EMULATED)
        const uint32_t n = *(const uint32_t __attribute__((address_space(4)))*)(unsigned long long)P.retry;
#else
        const uint32_t n = P.retry[0];
#endif
        for (uint32_t id = blockIdx.x; id < n; id += gridDim.x) {
            fast_tile<1>(P, (int)id);
            __syncthreads();                             // the next tile rewrites the LDS this one's last phase reads
        }
    } else {
        fast_tile<PASS>(P, 0);
    }
}

// ============================================================================================================
// E3  DistributeOctTree — one workgroup per (frame, level); the serial list algorithm restated as rounds of
//     data-parallel steps that reproduce the reference's node order, early break and tie-breaks exactly.
// ============================================================================================================
struct OctLevel {
    int W, H;            // maxBorderX-minBorderX, maxBorderY-minBorderY
    int N;               // mnFeaturesPerLevel[level]
    int nIni; float hX;  // ORBextractor.cc:541-543
    int nCols, wCell, hCell;
    int wCellM19, hCellM19;   // ceil(2^19 / cell side): v / side == (v * M) >> 19 for v < 4096 and sides <= 127 (v * (side * M - 2^19) < 2^19)
    size_t candOff; int candCap;
    int selOff; int selCap;   // per-frame offsets into sel/selAux (u32 units)
    float scale;         // mvScaleFactor[level]
};
struct OctParams {
    OctLevel lv[ORBX_MAX_LEVELS];
    const uint32_t* cand; size_t candFrame;
    const int* candCount; int nlevels;
    uint16_t* keyNode;                 // [frame][candFrame] scratch: current node of every key
    uint32_t* sel; uint32_t* selAux; int selFrame;   // [frame][selFrame]
    int* selCount; int* lapCount;      // [frame][nlevels]
    int nodeCap;                       // LDS node capacity C
    int merge;                         // 1: second child-count buffer present (key move counts the next round's children)
    int keyCap, keyOff;                // LDS key cache: capacity (keys) and byte offset inside the dynamic LDS block
    int lap0, lap1;
    const int* order;                  // as FastParams::order
    int gridLevels;                    // levels this launch's grid covers (0 .. gridLevels - 1; nlevels, or the levels that can hold a heavy problem)
    int heavyMode, heavyMin;           // 0: every problem; 2: problems of fewer than heavyMin candidates (a second launch, mode 1, takes the others with OCT_T_HEAVY threads)
    const uint32_t* retry; uint32_t* retryHost;   // non-null: the FAST second-pass list's {count, tiles} of this call -> the pinned host words (in place of a copy launch)
};

// In-place exclusive scan of a[0..n) (LDS) by the whole OCT_T-thread block; returns the total.  Thread-serial chunks, one
// DPP scan per wave, wave totals through LDS: two workgroup barriers per call (the octree calls this ~5 times per round).
// OCT_T = threads per (frame, level) octree problem, a template parameter of the section: batches run 256 (many problems share the machine), the
// single-frame entry point — eight problems on 256 compute units, every split round a chain of barriers — runs 1 024 (measured in round 4:
// faster for one frame, slower for a batch)
#ifndef OCT_T_BATCH
#define OCT_T_BATCH 256
#endif
#ifndef OCT_HEAVY_MIN
#define OCT_HEAVY_MIN 8192   // candidates of one (frame, level) from which a batch's problem goes to the OCT_T_HEAVY-thread launch (tests build with a small value)
#endif
#ifndef OCT_HEAVY_PASS
#define OCT_HEAVY_PASS 1
#endif
#ifndef OCT_T_SINGLE
#ifdef HIP_EMULATED
#define OCT_T_SINGLE 256    // (the CPU tier's emulator pays per work-item and barrier: one variant test builds the 1 024-thread instantiation)
#else
#define OCT_T_SINGLE 1024
#endif
#endif
#define OCT_T_HEAVY OCT_T_SINGLE   // threads of the heavy-problem launch of a batch: the single-frame shape
template <int OCT_T>
static __device__ int block_scan_excl(int* a, int n, int* scratch) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = (n + OCT_T - 1) / OCT_T;
    const int s0 = tid * chunk, s1 = min(s0 + chunk, n);
    int sum = 0;
    for (int i = s0; i < s1; i++) sum += a[i];
    const int incl = wave_scan_incl(sum);   // (register-only DPP steps; every thread of the block is here)
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < OCT_T / 64; w++) { const int v = scratch[w]; base += w < wave ? v : 0; total += v; }
    int run = base + incl - sum;
    for (int i = s0; i < s1; i++) { const int t = a[i]; a[i] = run; run += t; }
    __syncthreads();
    return total;
}

// The same scan over values that are COMPUTED by the scanning thread: f(i) for the entries of its own chunk (no barrier between producing the
// values and scanning them).  a[i] receives the exclusive prefix; two workgroup barriers.
template <int OCT_T, class F>
static __device__ __forceinline__ int block_scan_excl_fn(int* a, const int n, int* scratch, F f) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = (n + OCT_T - 1) / OCT_T;
    const int s0 = tid * chunk, s1 = min(s0 + chunk, n);
    int sum = 0;
    for (int i = s0; i < s1; i++) { const int t = f(i); a[i] = t; sum += t; }
    const int incl = wave_scan_incl(sum);
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < OCT_T / 64; w++) { const int v = scratch[w]; base += w < wave ? v : 0; total += v; }
    int run = base + incl - sum;
    for (int i = s0; i < s1; i++) { const int t = a[i]; a[i] = run; run += t; }
    __syncthreads();
    return total;
}
// Two such scans behind the same two barriers: a[0..na) over fa, b[0..nb) over fb.
template <int OCT_T, class FA, class FB>
static __device__ __forceinline__ void block_scan_excl_fn2(int* a, const int na, FA fa, int* b, const int nb, FB fb, int* scratch, int* totalA, int* totalB) {
    constexpr int NW = OCT_T / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ca = (na + OCT_T - 1) / OCT_T, cb = (nb + OCT_T - 1) / OCT_T;
    const int a0 = tid * ca, a1 = min(a0 + ca, na), b0 = tid * cb, b1 = min(b0 + cb, nb);
    int sa = 0, sb2 = 0;
    for (int i = a0; i < a1; i++) { const int t = fa(i); a[i] = t; sa += t; }
    for (int i = b0; i < b1; i++) { const int t = fb(i); b[i] = t; sb2 += t; }
    const int ia = wave_scan_incl(sa), ib = wave_scan_incl(sb2);
    if (lane == 63) { scratch[wave] = ia; scratch[NW + wave] = ib; }
    __syncthreads();
    int baseA = 0, totA = 0, baseB = 0, totB = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const int va = scratch[w], vb = scratch[NW + w];
        baseA += w < wave ? va : 0; totA += va;
        baseB += w < wave ? vb : 0; totB += vb;
    }
    int ra = baseA + ia - sa, rb = baseB + ib - sb2;
    for (int i = a0; i < a1; i++) { const int t = a[i]; a[i] = ra; ra += t; }
    for (int i = b0; i < b1; i++) { const int t = b[i]; b[i] = rb; rb += t; }
    __syncthreads();
    *totalA = totA; *totalB = totB;
}

// cnt[t] += 1 for every lane with t >= 0, aggregated by the wave — for the assignment walk, where ALL keys of a level fall on one or two roots and
// their eight quadrants: a plain ds_add_u32 then works through 64 lanes on the same word.  Up to OCT_AGG distinct targets are taken one by one
// (leader's target -> ballot of its lanes -> one add of their number); what is left adds directly.  Every lane of the wave must call it.
// (The later rounds' key moves spread over many nodes: aggregating those made the kernel slower.)
#ifndef OCT_AGG
#define OCT_AGG 8
#endif
static __device__ __forceinline__ void wave_agg_add(int* cnt, const int t) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(t >= 0);
    for (int it = 0; it < OCT_AGG && todo; it++) {
        const int leader = __ffsll((long long)todo) - 1;
        const int tt = __builtin_amdgcn_readlane(t, leader);
        const unsigned long long m = __ballot(t == tt);
        if (lane == leader) atomicAdd(&cnt[tt], __popcll(m));
        todo &= ~m;
    }
    if ((todo >> lane) & 1ull) atomicAdd(&cnt[t], 1);
}

struct ONode { short x0, y0, x1, y1; };
#ifndef OCT_KEYCAP
#define OCT_KEYCAP 4096   // candidates per (frame, level) that the LDS key cache holds (6 B each); levels with more take the global-memory path
#endif
#ifndef OCT_LEVEL_MAJOR
#define OCT_LEVEL_MAJOR 1   // workgroup id -> (level, frame) with the level in the slow position: the long problems (level 0 holds 4x the keys of level 7)
#endif                      // start first and the short ones fill the slots they leave — 0.203 -> 0.157 ms per 512 frames on MI355X against the
                            // frame-major order with a rotated level (0), where the launch ends on whichever level-0 problems happened to start last
#ifndef OCT_U
#define OCT_U 4   // key-walk unroll: loads of OCT_U strides are issued before any is consumed
#endif
#ifndef OCT_CACHE_MAX_BATCH
#define OCT_CACHE_MAX_BATCH 48   // the cache is used for batches up to this size.  It shortens a workgroup's life ~1.8x (every round walks all keys
                                 // twice: latency of the single-frame drop-in call) but costs 24 KB of LDS, i.e. half the workgroups per CU:
                                 // at batch 512 on MI355X 0.277 ms with it vs 0.253 ms without -> large batches run without it
#endif

// The list algorithm proper.  keys / keyNode live either in LDS (the usual case: every round walks all keys twice, and the
// per-workgroup critical path — the big levels — is latency bound) or in global memory (more candidates than the LDS cache holds).
template <int OCT_T>
static __device__ __forceinline__ void octree_run(const OctParams& P, const OctLevel& L, const int level, const int frame, const int nk,
                                                  const uint32_t* keys, uint16_t* keyNode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int tid = threadIdx.x;
    const int C = P.nodeCap;
    // LDS carve-up (all int-aligned)
    int* scratch = (int*)orb_smem;                 // 256
    int* ctl = scratch + 256;                      // 16 control words
    int* lp = ctl + 16;
    ONode* rect[2];
    rect[0] = (ONode*)lp; lp += 2 * C;
    rect[1] = (ONode*)lp; lp += 2 * C;
    int* cnt[2]; cnt[0] = lp; lp += C; cnt[1] = lp; lp += C;
    int* seq[2]; seq[0] = lp; lp += C; seq[1] = lp; lp += C;
    int* ccb[2]; ccb[0] = lp; lp += 4 * C; ccb[1] = lp; if (P.merge) lp += 4 * C;   // [C][4] child key counts, double-buffered: the key move of round r
                                                                       // already counts the children of round r+1 (one key walk per round)
    int* nchild = lp; lp += C;                     // [C] #non-empty children of an expandable node | 0x100 if divided
    int* sb = lp; lp += C;                         // [C] scan buffer (survivor positions)
    int* elist = lp; lp += C;                      // [C] expandable node indices (list order)
    int* porder = lp; lp += C;                     // [C] processing order
    int* pb = lp; lp += C;                         // [C] push base per processed node
    // [C] what the key move needs of a node of the CURRENT list in one 16-byte read: {x0 | undivided << 15 | y0 << 16, x1 | y1 << 16, and the list positions
    // of its children n1..n4 as four u16 (0xFFFF: empty) — of an undivided node its own new position four times}.  A key then costs ONE LDS read where
    // divided flag, rectangle, child position and the child's rectangle were a chain of four (the kernel's LDS pipe is its busiest unit: 59 % conflicts).
    uint4* mv = (uint4*)(((uintptr_t)lp + 15) & ~(uintptr_t)15);

    int* selCountOut = P.selCount + (size_t)frame * P.nlevels + level;
    int* lapCountOut = P.lapCount + (size_t)frame * P.nlevels + level;
    const int N = L.N;
    int cur = 0;
    PROF_DECL;   // (-DORBX_PROF builds: slot 1 of the phase timers is k_octree's — k_describe, whose slot it was, is not launched)
    // ---- roots (ORBextractor.cc:550-561) and key assignment by kp.pt.x/hX (:564-568)
    for (int i = tid; i < L.nIni; i += OCT_T) {
        ONode n;
        n.x0 = (short)(int)(L.hX * (float)i); n.y0 = 0;
        n.x1 = (short)(int)(L.hX * (float)(i + 1)); n.y1 = (short)L.H;
        rect[0][i] = n; cnt[0][i] = 0; seq[0][i] = i;
        if (P.merge) { int* c1 = ccb[1]; c1[4 * i] = 0; c1[4 * i + 1] = 0; c1[4 * i + 2] = 0; c1[4 * i + 3] = 0; }
    }
    __syncthreads();
    // (with the second child-count buffer the assignment walk also counts the keys into the roots' children — the first round's own key walk — and
    // the walk that renumbers the keys runs only if a root came back empty: two of a level's nine key walks)
    for (int kb = 0; kb < nk; kb += OCT_T) {             // (wave-uniform trip count: wave_agg_add)
        const int k = kb + tid;
        int r = -1, cq = -1;
        if (k < nk) {
            const uint32_t key = keys[k];
            const float x = (float)(key & 0xFFF);
            r = (int)(x / L.hX);
            r = min(r, L.nIni - 1);
            keyNode[k] = (uint16_t)r;
            if (P.merge) {
                const ONode n = rect[0][r];
                const int mx = n.x0 + ((n.x1 - n.x0 + 1) >> 1), my = n.y0 + ((n.y1 - n.y0 + 1) >> 1);
                const int kx = key & 0xFFF, ky = (key >> 12) & 0xFFF;
                cq = 4 * r + ((kx < mx) ? (ky < my ? 0 : 2) : (ky < my ? 1 : 3));
            }
        }
        wave_agg_add(cnt[0], r);
        if (P.merge) wave_agg_add(ccb[1], cq);
    }
    __syncthreads();
    // drop empty roots (:572-583), keep order
    int size = block_scan_excl_fn<OCT_T>(sb, L.nIni, scratch, [&](const int i) { return cnt[0][i] > 0 ? 1 : 0; });
    for (int i = tid; i < L.nIni; i += OCT_T)
        if (cnt[0][i] > 0) {
            const int np = sb[i];
            rect[1][np] = rect[0][i]; cnt[1][np] = cnt[0][i]; seq[1][np] = seq[0][i];
            if (P.merge) *(int4*)&ccb[0][4 * np] = *(const int4*)&ccb[1][4 * i];
        }
    __syncthreads();
    if (size != L.nIni) {                                // (workgroup-uniform)
        for (int k = tid; k < nk; k += OCT_T) keyNode[k] = (uint16_t)sb[keyNode[k]];
        __syncthreads();
    }
    cur = 1;
    PROF_MARK(1, 0);   // roots, key assignment, empty roots dropped
    int seqCounter = L.nIni;
    bool sortedMode = false;
    int ccCur = 0;
    bool haveCounts = P.merge != 0;                      // (counted by the assignment walk)

    for (;;) {
        const int prevSize = size;
        int* cc = ccb[ccCur];
        int* ccn = ccb[ccCur ^ 1];
        const ONode* R = rect[cur];
        const int* CN = cnt[cur];
        // Barriers, not instructions, are this kernel's time (a round had 17, a sorted one 26): flags are evaluated inside the scans by the thread
        // that owns the entry, the two scans of step 4 share their barriers, a list-order round needs no copy of the list and no scatter of
        // its "divided" marks, and the children of a node are counted where the node is listed — 7 barriers per list-order round, 21 per sorted one.
        // 1. expandable nodes E (count > 1), list order
        if (tid == 0) ctl[1] = 0;                           // (step 5's counter: last read before the previous round's closing barrier)
        const int nE = block_scan_excl_fn<OCT_T>(sb, size, scratch, [&](const int i) { return CN[i] > 1 ? 1 : 0; });
        if (nE == 0) break;  // nothing can be divided: size stays == prevSize (:667)
        // nchild[i]: #non-empty children of an expandable node; | 0x100 = "divided" — in a list-order round every expandable node is
        const int dflag = sortedMode ? 0 : 0x100;
        for (int i = tid; i < size; i += OCT_T) {
            int nc = 0;
            if (CN[i] > 1) {
                elist[sb[i]] = i;
                if (!haveCounts) { cc[4 * i] = 0; cc[4 * i + 1] = 0; cc[4 * i + 2] = 0; cc[4 * i + 3] = 0; }
                else nc = ((cc[4 * i] > 0) + (cc[4 * i + 1] > 0) + (cc[4 * i + 2] > 0) + (cc[4 * i + 3] > 0)) | dflag;
            }
            nchild[i] = nc;
        }
        __syncthreads();
        PROF_MARK(1, 1);   // expandable-node scan + list
        // 2. child key counts (DivideNode :479-535): the first round (later ones get them from the previous round's key move)
        // (key walks are unrolled by 4 with the loads hoisted: the walk is a chain of global-memory round trips otherwise, and a
        // workgroup's lifetime — not its instruction count — is what bounds this kernel)
        if (!haveCounts) {
            for (int k0 = tid; k0 < nk; k0 += OCT_U * OCT_T) {
                int ndv[OCT_U]; uint32_t kyv[OCT_U];
#pragma unroll
                for (int u = 0; u < OCT_U; u++) { const int k = k0 + u * OCT_T; ndv[u] = k < nk ? (int)keyNode[k] : -1; kyv[u] = k < nk ? keys[k] : 0u; }
#pragma unroll
                for (int u = 0; u < OCT_U; u++) {
                    const int nd = ndv[u];
                    if (nd >= 0 && CN[nd] > 1) {
                        const ONode n = R[nd];
                        const int mx = n.x0 + ((n.x1 - n.x0 + 1) >> 1), my = n.y0 + ((n.y1 - n.y0 + 1) >> 1);
                        const int x = kyv[u] & 0xFFF, y = (kyv[u] >> 12) & 0xFFF;
                        const int q = (x < mx) ? (y < my ? 0 : 2) : (y < my ? 1 : 3);
                        atomicAdd(&cc[4 * nd + q], 1);
                    }
                }
            }
            __syncthreads();
            for (int e = tid; e < nE; e += OCT_T) {
                const int i = elist[e];
                nchild[i] = ((cc[4 * i] > 0) + (cc[4 * i + 1] > 0) + (cc[4 * i + 2] > 0) + (cc[4 * i + 3] > 0)) | dflag;
            }
            __syncthreads();
        }
        PROF_MARK(1, 2);   // first round: child counts by a key walk
        // 3. processing order and cut
        int nProc = nE;
        const int* po = elist;                              // list order: the list itself
        if (sortedMode) {
            po = porder;
            // descending (size, creation seq): rule R1 replaces the reference's pointer tie-break (:679-683)
            // (sb / pb are free here: the (count, seq) sort keys are staged list-order so the rank loop reads two broadcast words per
            // step instead of chasing elist -> CN / seq)
            for (int e = tid; e < nE; e += OCT_T) { const int i = elist[e]; sb[e] = CN[i]; pb[e] = seq[cur][i]; }
            __syncthreads();
            for (int e = tid; e < nE; e += OCT_T) {
                const int ci = sb[e], si = pb[e];
                int rank = 0;
                int f = 0;
                for (; f + 4 <= nE; f += 4) {
                    const int c0 = sb[f], c1 = sb[f + 1], c2 = sb[f + 2], c3 = sb[f + 3];
                    const int s0 = pb[f], s1 = pb[f + 1], s2 = pb[f + 2], s3 = pb[f + 3];
                    rank += ((c0 > ci) || (c0 == ci && s0 > si)) + ((c1 > ci) || (c1 == ci && s1 > si)) +
                            ((c2 > ci) || (c2 == ci && s2 > si)) + ((c3 > ci) || (c3 == ci && s3 > si));
                }
                for (; f < nE; f++) { const int cj = sb[f]; rank += (cj > ci) || (cj == ci && pb[f] > si); }
                porder[rank] = elist[e];
            }
            __syncthreads();
            // early break once lNodes.size() >= N (:728-729): running size after each division
            for (int e = tid; e < nE; e += OCT_T) pb[e] = nchild[porder[e]] - 1;
            __syncthreads();
            block_scan_excl<OCT_T>(pb, nE, scratch);  // pb[e] = growth before processing e
            if (tid == 0) ctl[0] = nE;
            __syncthreads();
            for (int e = tid; e < nE; e += OCT_T) {
                const int after = prevSize + pb[e] + nchild[porder[e]] - 1;
                if (after >= N) atomicMin(&ctl[0], e + 1);
            }
            __syncthreads();
            nProc = ctl[0];
            __syncthreads();
        }
        PROF_MARK(1, 3);   // sorted rounds: rank, cut
        // 4. push bases (children are push_front'ed in processing order, n1..n4) and the survivors' positions: one pair of scans
        if (sortedMode) {                                   // the cut leaves expandable nodes undivided: mark the processed ones
            for (int i = tid; i < size; i += OCT_T) sb[i] = 1;  // 1 = survives
            __syncthreads();
            for (int e = tid; e < nProc; e += OCT_T) { sb[po[e]] = 0; nchild[po[e]] |= 0x100; }
            __syncthreads();
        }
        int totalPushed, nSurv;
        if (sortedMode)
            block_scan_excl_fn2<OCT_T>(pb, nProc, [&](const int e) { return nchild[po[e]] & 0xFF; }, sb, size, [&](const int i) { return sb[i]; }, scratch, &totalPushed, &nSurv);
        else
            block_scan_excl_fn2<OCT_T>(pb, nProc, [&](const int e) { return nchild[po[e]] & 0xFF; }, sb, size, [&](const int i) { return CN[i] > 1 ? 0 : 1; }, scratch, &totalPushed,
                                &nSurv);
        PROF_MARK(1, 4);   // the two scans of step 4
        const int newSize = totalPushed + nSurv;
        // 5. build the next list: [children, most recently pushed first] ++ [survivors in order]
        const int nxt = cur ^ 1;
        for (int e = tid; e < nProc; e += OCT_T) {
            const int i = po[e];
            const ONode n = R[i];
            const int hx = (n.x1 - n.x0 + 1) >> 1, hy = (n.y1 - n.y0 + 1) >> 1;
            int p = pb[e];
            int nexp = 0;
            uint32_t cp[4];
            for (int q = 0; q < 4; q++) {
                const int c = cc[4 * i + q];
                if (c == 0) { cp[q] = 0xFFFFu; continue; }
                ONode ch;
                ch.x0 = (q & 1) ? (short)(n.x0 + hx) : n.x0;
                ch.x1 = (q & 1) ? n.x1 : (short)(n.x0 + hx);
                ch.y0 = (q & 2) ? (short)(n.y0 + hy) : n.y0;
                ch.y1 = (q & 2) ? n.y1 : (short)(n.y0 + hy);
                const int np = totalPushed - 1 - p;
                if (np < C) { rect[nxt][np] = ch; cnt[nxt][np] = c; seq[nxt][np] = seqCounter + p; if (P.merge) *(int4*)&ccn[4 * np] = make_int4(0, 0, 0, 0); }
                cp[q] = (uint32_t)np & 0xFFFFu;
                nexp += c > 1;
                p++;
            }
            mv[i] = make_uint4((uint32_t)(uint16_t)n.x0 | ((uint32_t)(uint16_t)n.y0 << 16), (uint32_t)(uint16_t)n.x1 | ((uint32_t)(uint16_t)n.y1 << 16),
                               cp[0] | (cp[1] << 16), cp[2] | (cp[3] << 16));
            if (nexp) atomicAdd(&ctl[1], nexp);
        }
        for (int i = tid; i < size; i += OCT_T)
            if (!(nchild[i] & 0x100)) {
                const int np = totalPushed + sb[i];
                const ONode n = R[i];
                if (np < C) { rect[nxt][np] = n; cnt[nxt][np] = CN[i]; seq[nxt][np] = seq[cur][i]; if (P.merge) *(int4*)&ccn[4 * np] = make_int4(0, 0, 0, 0); }
                const uint32_t pp = ((uint32_t)np & 0xFFFFu) * 0x00010001u;
                mv[i] = make_uint4((uint32_t)(uint16_t)n.x0 | 0x8000u | ((uint32_t)(uint16_t)n.y0 << 16), (uint32_t)(uint16_t)n.x1 | ((uint32_t)(uint16_t)n.y1 << 16), pp, pp);
            }
        __syncthreads();
        PROF_MARK(1, 5);   // next list built
        // 6. move keys, and count them into the children of their NEW node (next round's step 2)
        {
            for (int k0 = tid; k0 < nk; k0 += OCT_U * OCT_T) {
                int ndv[OCT_U]; uint32_t kyv[OCT_U];
#pragma unroll
                for (int u = 0; u < OCT_U; u++) { const int k = k0 + u * OCT_T; ndv[u] = k < nk ? (int)keyNode[k] : -1; kyv[u] = k < nk ? keys[k] : 0u; }
#pragma unroll
                for (int u = 0; u < OCT_U; u++) {
                    const int nd = ndv[u], k = k0 + u * OCT_T;
                    if (nd < 0) continue;
                    const int x = kyv[u] & 0xFFF, y = (kyv[u] >> 12) & 0xFFF;
                    const uint4 rec = mv[nd];
                    const bool und = (rec.x & 0x8000u) != 0;
                    const int x0 = (int)(rec.x & 0x7FFFu), y0 = (int)(rec.x >> 16), x1 = (int)(rec.y & 0xFFFFu), y1 = (int)(rec.y >> 16);
                    const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1;
                    const int q = (x < x0 + hx ? 0 : 1) | (y < y0 + hy ? 0 : 2);   // == (x < mx) ? (y < my ? 0 : 2) : (y < my ? 1 : 3)
                    const uint32_t cpw = (q & 2) ? rec.w : rec.z;
                    const int nn = (int)((q & 1) ? cpw >> 16 : cpw & 0xFFFFu);      // (an undivided node: its own new position in all four)
                    keyNode[k] = (uint16_t)nn;
                    if (P.merge && nn < C) {
                        // the rectangle of the key's NEW node, as step 5 made it (the quadrant's child, or the node itself), and the key's quadrant in it;
                        // counted for every node (the counts of a node with one key are never read)
                        int cx0 = x0, cx1 = x1, cy0 = y0, cy1 = y1;
                        if (!und) {
                            if (q & 1) cx0 = x0 + hx; else cx1 = x0 + hx;
                            if (q & 2) cy0 = y0 + hy; else cy1 = y0 + hy;
                        }
                        const int cmx = cx0 + ((cx1 - cx0 + 1) >> 1), cmy = cy0 + ((cy1 - cy0 + 1) >> 1);
                        atomicAdd(&ccn[4 * nn + ((x < cmx ? 0 : 1) | (y < cmy ? 0 : 2))], 1);
                    }
                }
            }
        }
        const int nToExpand = ctl[1];
        __syncthreads();
        PROF_MARK(1, 6);   // key move
        cur = nxt;
        if (P.merge) { ccCur ^= 1; haveCounts = true; }
        size = min(newSize, C);
        seqCounter += totalPushed;
        // 7. termination (:667-671, :731-732)
        if (newSize >= N || newSize == prevSize) break;
        if (!sortedMode && newSize + 3 * nToExpand > N) sortedMode = true;
    }

    // ---- best key per node: max response, first in vToDistributeKeys order on ties (:737-758).
    // Original order = cell-row-major, then row-major inside the cell's detection region.
    // One walk: a 64-bit LDS atomicMax on (response << 32 | ~order) picks both; order encodes the key's position, so the winning key is
    // rebuilt from the payload by the node's thread instead of two more walks over the candidates.
    unsigned long long* best = (unsigned long long*)ccb[0];   // [C] (cc holds 4C ints, 16-byte aligned)
    for (int i = tid; i < size; i += OCT_T) best[i] = 0ull;
    __syncthreads();
    for (int k0 = tid; k0 < nk; k0 += OCT_U * OCT_T) {
        int ndv[OCT_U]; uint32_t kyv[OCT_U];
#pragma unroll
        for (int u = 0; u < OCT_U; u++) { const int k = k0 + u * OCT_T; ndv[u] = k < nk ? (int)keyNode[k] : -1; kyv[u] = k < nk ? keys[k] : 0u; }
#pragma unroll
        for (int u = 0; u < OCT_U; u++) {
            if (ndv[u] < 0) continue;
            const uint32_t key = kyv[u];
            const int x = (int)(key & 0xFFF) - 3, y = (int)((key >> 12) & 0xFFF) - 3;
            // (two integer divisions by run-time values per candidate were a fifth of the kernel's instructions: multiply-shift by tabulated reciprocals,
            // exact for the 12-bit coordinates and cell sides below 128)
            const int cj = (int)(mul24((uint32_t)x, (uint32_t)L.wCellM19) >> 19), ci = (int)(mul24((uint32_t)y, (uint32_t)L.hCellM19) >> 19);
            const uint32_t ord = (uint32_t)((((ci * L.nCols + cj) * 128 + (y - ci * L.hCell)) * 128) + (x - cj * L.wCell));  // cell sides < 70
            atomicMax(&best[ndv[u]], ((unsigned long long)((key >> 24) + 1u) << 32) | (0x7FFFFFFFu - ord));
        }
    }
    __syncthreads();
    uint32_t* sel = P.sel + (size_t)frame * P.selFrame + L.selOff;
    uint32_t* selAux = P.selAux + (size_t)frame * P.selFrame + L.selOff;
    for (int i = tid; i < min(size, L.selCap); i += OCT_T) {
        const unsigned long long b = best[i];
        const uint32_t ord = 0x7FFFFFFFu - (uint32_t)(b & 0xFFFFFFFFu);
        const int cell = (int)(ord >> 14), dy = (int)((ord >> 7) & 127), dx = (int)(ord & 127);
        const int ci = cell / L.nCols, cj = cell - ci * L.nCols;
        const uint32_t x = (uint32_t)(cj * L.wCell + dx + 3), y = (uint32_t)(ci * L.hCell + dy + 3);
        sel[i] = (((uint32_t)(b >> 32) - 1u) << 24) | (y << 12) | x;
    }
    __syncthreads();
    // ---- E8 ordering ranks: lapping keypoints are written from the back (ORBextractor.cc:1137-1152)
    const int nsel = min(size, L.selCap);
    for (int i = tid; i < nsel; i += OCT_T) {
        float xs = (float)((int)(sel[i] & 0xFFF) + ORBX_MINB);
        if (level != 0) xs = xs * L.scale;
        sb[i] = (xs >= (float)P.lap0 && xs <= (float)P.lap1) ? 1 : 0;
        nchild[i] = sb[i];
    }
    __syncthreads();
    const int nLap = block_scan_excl<OCT_T>(sb, nsel, scratch);
    for (int i = tid; i < nsel; i += OCT_T) {
        const int lap = nchild[i];
        const int rank = lap ? sb[i] : (i - sb[i]);
        selAux[i] = (uint32_t)rank | ((uint32_t)lap << 31);
    }
    if (tid == 0) { *selCountOut = nsel; *lapCountOut = nLap; }
    PROF_MARK(1, 7);   // best key per node, outputs
    PROF_FLUSH(1);
}

template <int OCT_T>
static __global__ __launch_bounds__(OCT_T) void k_octree(OctParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int tid = threadIdx.x;
    // Workgroups are dealt round-robin to the 8 XCDs by linear id.  Level-major (id = level * frames + frame): an XCD owns the frames = its number
    // (mod 8) with all their levels, and the longest problems are dispatched first.  Frame-major with (level, frame) = (id % nlevels, id / nlevels)
    // and 8 levels would give every XCD ONE level (XCD 0 all the level-0 problems, 4x the work of level 7); the alternative build rotates the
    // level by the frame index for an even mix.
#if OCT_LEVEL_MAJOR
    const int nframes = gridDim.x / P.gridLevels;
    const int level = blockIdx.x / nframes, slot = blockIdx.x - level * nframes;
#else
    const int slot = blockIdx.x / P.gridLevels;
    const int level = (blockIdx.x - slot * P.gridLevels + slot) % P.gridLevels;
#endif
    if (P.retryHost && blockIdx.x == 0 && tid == 0) *(unsigned long long*)P.retryHost = *(const unsigned long long*)P.retry;   // (both words in one store: a pair of one call)
    const int frame = P.order ? P.order[slot] : slot;
    const OctLevel& L = P.lv[level];
    int nk = P.candCount[(size_t)frame * P.nlevels + level];
    nk = min(nk, L.candCap);
    if (P.heavyMode == 1 ? nk < P.heavyMin : (P.heavyMode == 2 && nk >= P.heavyMin)) return;   // the other launch's problem
    const uint32_t* gkeys = P.cand + (size_t)frame * P.candFrame + L.candOff;
    if (nk == 0) {
        if (tid == 0) { P.selCount[(size_t)frame * P.nlevels + level] = 0; P.lapCount[(size_t)frame * P.nlevels + level] = 0; }
        return;
    }
    if (nk <= P.keyCap) {
        uint32_t* lkeys = (uint32_t*)(orb_smem + P.keyOff);
        uint16_t* lnode = (uint16_t*)(lkeys + P.keyCap);
        for (int k = tid; k < nk; k += OCT_T) lkeys[k] = gkeys[k];
        __syncthreads();
        octree_run<OCT_T>(P, L, level, frame, nk, lkeys, lnode);
    } else {
        octree_run<OCT_T>(P, L, level, frame, nk, gkeys, P.keyNode + (size_t)frame * P.candFrame + L.candOff);
    }
}

// ============================================================================================================
// E5-E8  one wave per keypoint: IC_Angle on the level image, 7x7 sigma=2 integer Gaussian of the 43x43
//        neighbourhood (only the 37x37 the rBRIEF pattern can reach), 256 rotated comparisons, output write.
// ============================================================================================================
struct DescLevel {
    const uint8_t* base; size_t frameStride; int rowStride;
    int w, h; int selOff; float scale; float size;   // size = (float)(int)(31*scale)
    int selCap;
};
struct DescParams {
    DescLevel lv[ORBX_MAX_LEVELS];
    const uint32_t* sel; const uint32_t* selAux; int selFrame;
    const int* selCount; const int* lapCount; int nlevels;
    orb_keypoint* kps; uint8_t* desc; int cap; int32_t* counts;
    int groups, batch;   // workgroups (4 keypoints each) per frame, frames: the XCD-aware 1-D grid
    uint32_t groupsMagic; // xcd_units_magic(groups * 2): k_describe2's units per frame
    int unitStart[ORBX_MAX_LEVELS + 1];   // workgroup u of a frame serves level l with unitStart[l] <= u < unitStart[l+1] (ceil(selCap_l / 4) each)
};

// cv::fastAtan2 (degrees), restated; float ops must not be contracted
static __device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float s = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
    const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const float ax = fabsf(x), ay = fabsf(y);
    // the reference's two branches (ax >= ay: c = ay / (ax + eps); else c = ax / (ay + eps) and a = 90 - a) as selects around ONE division and
    // polynomial: the same operations on the same values, and lanes that disagree do not run the ~22 instructions twice
    const bool steep = ax < ay;
    const float c = (steep ? ax : ay) / ((steep ? ay : ax) + (float)DBL_EPSILON);
    const float c2 = c * c;
    float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    if (steep) a = 90.f - a;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// Rule R2: sin/cos as a fixed double polynomial rounded once to float
static __device__ __forceinline__ void det_sincos(float angle, float* s_out, float* c_out) {
    const double x = (double)angle;
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632673412561417e+00;
    const double PIO2_LO = 6.07710050650619224932e-11;
    const double kd = floor(x * TWO_OVER_PI + 0.5);
    const int k = (int)kd;
    const double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    const double z = r * r;
    double ps = -1.0 / 355687428096000.0;
    ps = ps * z + 1.0 / 1307674368000.0;
    ps = ps * z - 1.0 / 6227020800.0;
    ps = ps * z + 1.0 / 39916800.0;
    ps = ps * z - 1.0 / 362880.0;
    ps = ps * z + 1.0 / 5040.0;
    ps = ps * z - 1.0 / 120.0;
    ps = ps * z + 1.0 / 6.0;
    const double sr = r - r * z * ps;
    double pc = 1.0 / 20922789888000.0;
    pc = pc * z - 1.0 / 87178291200.0;
    pc = pc * z + 1.0 / 479001600.0;
    pc = pc * z - 1.0 / 3628800.0;
    pc = pc * z + 1.0 / 40320.0;
    pc = pc * z - 1.0 / 720.0;
    pc = pc * z + 1.0 / 24.0;
    pc = pc * z - 0.5;
    const double cr = 1.0 + z * pc;
    double s, c;
    switch (k & 3) {
        case 0: s = sr; c = cr; break;
        case 1: s = cr; c = -sr; break;
        case 2: s = -sr; c = -cr; break;
        default: s = -cr; c = sr; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

#define DP 43            // source patch edge (radius 21 = 18 pattern reach + 3 blur taps)
#define DPP 52           // patch pitch in bytes (13 dwords: odd -> lane-per-row accesses are bank-conflict free)
#define DB 37            // blurred edge (radius 18)
#define DRP 46           // row-pass buffer: 37 columns x 46 rows of u16 (transposed; 23 dwords per column, odd)
#define DBP 40           // blurred pitch
#ifndef DESC_ALIAS
#define DESC_ALIAS 1
#endif
#ifndef DESC_SPARSE
#define DESC_SPARSE 1   // 1: the Gaussian column pass is evaluated only where rBRIEF samples (512 points per key point, 7 taps each) instead of on all
#endif                  //    37 x 37 pixels of the blurred neighbourhood (9 583 taps) — no blurred tile is materialised at all (round 3: 701 -> 642 VALU
                        //    instructions per key point, 0.665 -> 0.625 ms per 512 frames, 3 408 instead of 4 896 B of LDS per wave)
#if DESC_ALIAS
// LDS per keypoint: [ region A: the 43x52 source patch, overwritten IN PLACE by the 37x46 u16 row-pass buffer (3404 B) | 37x40 blurred tile ].
// A lane reads its whole patch row into registers before any lane of the wave writes a row-pass value (same wave, program order, fenced),
// so the two can share storage: 4896 B per wave instead of 5648 -> 8 workgroups per CU instead of 7.
#define DESC_ROWP_OFF 0
#define DESC_BLUR_OFF 3408
#if DESC_SPARSE
#define DESC_WAVE_STRIDE 3408   // no blurred tile
#else
#define DESC_WAVE_STRIDE 4896
#endif
#else
#define DESC_ROWP_OFF (DP * DPP)
#define DESC_BLUR_OFF 0
#define DESC_WAVE_STRIDE 5648   // 43*52 (patch; reused for the 37x40 blurred tile once the row pass is done) + 37*46*2 (row pass) + pad to 16
#endif

#ifndef DESC_WAVES
#define DESC_WAVES 8   // waves per SIMD the register allocator leaves room for.  The kernel's load phase is latency bound, so residency matters:
                      // 76 VGPRs / 5648 B of LDS per wave gave 6 workgroups per CU (0.850 ms on MI355X), 72 VGPRs 7 (0.792 ms); with the row-pass
                      // buffer aliased onto the patch (DESC_ALIAS) it is 57 VGPRs, 4896 B and 8 workgroups (0.759 ms)
#endif
#ifndef DESC_WPB
#define DESC_WPB 1    // keypoints (= waves) per workgroup: 1 (every wave its own workgroup: no cross-wave barrier coupling, wave-level LDS hand-offs;
                      // the column pass runs over the wave's own 111 tasks) or 4 (block-cooperative column pass, trig shared by 4 keypoints).
                      // MI355X, batch 512: 0.635 ms vs 0.694 ms.
#endif
#if DESC_WPB == 1   // one wave per workgroup: LDS hand-offs need program order inside the wave only
#define DESC_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define DESC_SYNC() __syncthreads()
#endif
static __global__ __launch_bounds__(64 * DESC_WPB, DESC_WAVES) void k_describe(DescParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int frame, grp;
#if DESC_WPB == 4
    if (!xcd_frame_unit(P.groups, P.batch, &frame, &grp)) return;
    const int kwave = wave;                     // keypoint of the group this wave serves
#else
    if (!xcd_frame_unit(P.groups * 4, P.batch, &frame, &grp)) return;
    const int kwave = grp & 3;
    grp >>= 2;
#endif
    PROF_DECL;
    uint8_t* patch = orb_smem + wave * DESC_WAVE_STRIDE;
    uint16_t* rowp = (uint16_t*)(patch + DESC_ROWP_OFF);
    uint8_t* blur = patch + DESC_BLUR_OFF; (void)blur;

    // The workgroup's level and position follow from its index alone (a constant table), so the keypoint record is fetched in the FIRST
    // global round trip, together with the per-level counts that are only needed for `valid` and, at the very end, for the output slot;
    // the patch rows are the second round trip.  (Locating keypoint g through the prefix sums of the counts first cost a third one, and
    // this kernel's load phase is latency bound.)
    // Everything of this prologue is ONE memory round trip: the level follows from a fixed-trip compare chain over the kernel-argument table
    // (unused entries hold INT_MAX; a loop bounded by nlevels compiles to one dependent scalar load per level), and the per-level counts are
    // fetched by lanes 0..nlevels-1 with two coalesced loads next to the keypoint record and reduced in registers (a scalar loop over the
    // levels compiles to one dependent round trip per level: that was a third of the kernel's wave time).
    int level = 0, ustart = 0;
#pragma unroll
    for (int l = 1; l < ORBX_MAX_LEVELS; l++) { const int us = P.unitStart[l]; if (grp >= us) { level = l; ustart = us; } }
    const int pos = (grp - ustart) * 4 + kwave;
    const DescLevel& L = P.lv[level];
    const bool inSlab = pos < L.selCap;
    uint32_t key = 0, aux = 0;
    if (inSlab) {
        key = P.sel[(size_t)frame * P.selFrame + L.selOff + pos];
        aux = P.selAux[(size_t)frame * P.selFrame + L.selOff + pos];
    }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIP_EMULATED) && !defined(DESC_COUNTS_VECTOR)
    // The per-level counts and their prefix sums are wave-uniform (frame and level are): scalar loads through the constant address space —
    // adjacent dwords, merged into wide s_loads, ONE round trip next to the key point record — and scalar adds, instead of two vector loads,
    // two wave scans and four v_readlane per key point (~35 VALU instructions of 640).  (The scalar cache is invalidated at kernel boundaries:
    // k_octree's counts are visible.)
    int nTotal = 0, monoTotal = 0, nLevel = 0, monoBase = 0, lapBase = 0;
    {
        const int __attribute__((address_space(4)))* cN = (const int __attribute__((address_space(4)))*)(unsigned long long)(P.selCount + (size_t)frame * P.nlevels);
        const int __attribute__((address_space(4)))* cL = (const int __attribute__((address_space(4)))*)(unsigned long long)(P.lapCount + (size_t)frame * P.nlevels);
        // eight levels per block, each block's sixteen entries loaded unconditionally (the arrays are padded) — loads behind a uniform
        // branch per level are one round trip per level, and all sixteen levels at once spill scalar registers; the second block only runs for
        // pyramids of more than eight levels
#pragma unroll
        for (int blk = 0; blk < ORBX_MAX_LEVELS; blk += 8) {
            if (blk < P.nlevels) {
                int nn[8], ll[8];
#pragma unroll
                for (int l = 0; l < 8; l++) { nn[l] = cN[blk + l]; ll[l] = cL[blk + l]; }
#pragma unroll
                for (int l = 0; l < 8; l++) {
                    const int n = blk + l < P.nlevels ? nn[l] : 0, lp = blk + l < P.nlevels ? ll[l] : 0;
                    nTotal += n; monoTotal += n - lp;
                    monoBase += blk + l < level ? n - lp : 0;          // monocular / lapping-area keypoints of the levels before this one
                    lapBase += blk + l < level ? lp : 0;
                    nLevel = blk + l == level ? n : nLevel;
                }
            }
        }
    }
#else
    int myN = 0, myL = 0;
    if (lane < P.nlevels) {
        myN = P.selCount[(size_t)frame * P.nlevels + lane];
        myL = P.lapCount[(size_t)frame * P.nlevels + lane];
    }
    const int myM = myN - myL;                                  // monocular keypoints of level `lane`
    const int inclN = wave_scan_incl(myN), inclM = wave_scan_incl(myM);
    const int nTotal = __builtin_amdgcn_readlane(inclN, 63), monoTotal = __builtin_amdgcn_readlane(inclM, 63);
    const int nLevel = __builtin_amdgcn_readlane(myN, level);
    const int monoBase = __builtin_amdgcn_readlane(inclM - myM, level);               // monocular keypoints of the levels before this one
    const int lapBase = __builtin_amdgcn_readlane(inclN - myN, level) - monoBase;     // lapping-area keypoints of the levels before this one
#endif
    if (grp == 0 && kwave == 0 && lane == 0) { P.counts[2 * frame] = nTotal; P.counts[2 * frame + 1] = monoTotal; }
    const bool valid = inSlab && pos < nLevel;
    PROF_MARK(1, 0);   // record + counts (first global round trip)
    const int cx = (int)(key & 0xFFF) + ORBX_MINB, cy = (int)((key >> 12) & 0xFFF) + ORBX_MINB;
    int ox = 0;   // column of the patch's first pixel inside the LDS rows
    if (valid) {
        const uint8_t* img = L.base + (size_t)frame * L.frameStride;
        const int xs = cx - 21, ys = cy - 21;
        if (xs >= 0 && ys >= 0 && cy + 21 < L.h && cx + 27 <= L.w) {
            // interior: 43 rows x 12 aligned dwords (48 B cover the 43 columns at any alignment), coalesced per row
            const int x0 = xs & ~3;
            ox = xs - x0;
            // lanes 0..59 = 5 rows x 12 dwords per pass, 9 passes (rows 43, 44 of the last pass are skipped); all loads are issued
            // before the first LDS store
            const int c = lane % 12, r5 = lane / 12;
            const uint8_t* src = img + (size_t)(ys + r5) * L.rowStride + x0 + 4 * c;
            uint8_t* dstp = patch + r5 * DPP + 4 * c;
            uint32_t v[9];
            if (lane < 60) {
#pragma unroll
                for (int k = 0; k < 9; k++)
                    if (k < 8 || r5 < 3) v[k] = *(const uint32_t*)(src + (size_t)(5 * k) * L.rowStride);
#pragma unroll
                for (int k = 0; k < 9; k++)
                    if (k < 8 || r5 < 3) *(uint32_t*)(dstp + 5 * k * DPP) = v[k];
            }
        } else {
            // the 7x7 blur taps may cross the image border: BORDER_REFLECT_101 at load (GaussianBlur on the un-bordered clone)
            for (int i = lane; i < DP * DP; i += 64) {
                const int r = i / DP, c = i - r * DP;
                const int y = reflect101(ys + r, L.h), x = reflect101(xs + c, L.w);
                patch[r * DPP + c] = img[(size_t)y * L.rowStride + x];
            }
        }
    }
    PROF_MARK(1, 1);   // patch rows (second round trip) -> LDS
    DESC_SYNC();
    PROF_MARK(1, 2);   // barrier 1
#ifndef DESC_KO
#define DESC_KO 0   // experiment builds only: 1 no column pass, 2 no row pass, 3 no rBRIEF, 4 no IC_Angle (results are wrong; timing split)
#endif
    float angle = 0.f;
    int* vflag = (int*)(orb_smem + DESC_WPB * DESC_WAVE_STRIDE);   // [4] this wave holds a keypoint
    int* mom = vflag + 4;                                   // [4][2] m01, m10 of the four keypoints
    float* trig = (float*)(mom + 8);                        // [4][3] angle (degrees), sin, cos — computed once per keypoint by lanes 0..3 of wave 0
    if (lane == 0) vflag[wave] = valid ? 1 : 0;
    if (valid) {
        // lane r owns patch row r: 12 aligned dword LDS reads, realigned by the wave-uniform column offset ox into packed dwords
        // e[k] = patch bytes 4k..4k+3; both IC_Angle (rows 6..36) and the Gaussian row pass (all 43 rows) run on packed bytes
        // with v_dot4_u32_u8
        int m10 = 0, m01 = 0;
        uint32_t e[11];   // e[k] = bytes 4k..4k+3 of this lane's patch row
        if (lane < DP) {
            const uint32_t* rw = (const uint32_t*)(patch + lane * DPP);
            uint32_t d[12];
#pragma unroll
            for (int k = 0; k < 12; k++) d[k] = rw[k];
#pragma unroll
            for (int k = 0; k < 11; k++) e[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], (uint32_t)ox);
            // IC_Angle (ORBextractor.cc:75-102): integer moments over the circular patch of radius 15; row v = lane - 21,
            // column u = byte index - 21.  m10 = sum u*I = sum (u+15)*I - 15*sum I over the row's masked bytes (weights 0..30 fit u8).
            const int v = lane - 21;
            const int av = v < 0 ? -v : v;
            if (av <= 15 && DESC_KO != 4) {
                const uint32_t* mk = c_icmask[av];
                uint32_t s1 = 0, sw = 0;
#pragma unroll
                for (int k = 1; k <= 9; k++) {
                    uint32_t W = 0;
#pragma unroll
                    for (int t = 0; t < 4; t++) { const int j = 4 * k + t; if (j >= 6 && j <= 36) W |= (uint32_t)(j - 6) << (8 * t); }
                    const uint32_t m = e[k] & mk[k];
                    s1 = __builtin_amdgcn_udot4(m, 0x01010101u, s1, false);
                    sw = __builtin_amdgcn_udot4(m, W, sw, false);
                }
                m10 = (int)sw - 15 * (int)s1;
                m01 = v * (int)s1;
            }
        }
        // (DESC_ALIAS: the row-pass buffer overwrites the patch; every lane's patch row is in registers by now — the wave-level fence keeps
        // the LDS reads above the writes)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < DP && DESC_KO != 2) {
            // Gaussian row pass: k = cvRound(256*g) = {18,34,49,55,49,34,18}; out(c) = dot4(bytes c..c+3, k[0..3]) + dot4(bytes c+4..c+7,
            // {k[4..6],0}); sums <= 255*257 fit u16
            const uint32_t K0 = 18u | (34u << 8) | (49u << 16) | (55u << 24), K1 = 49u | (34u << 8) | (18u << 16);
            uint32_t A[41];   // A[c] = bytes c..c+3
#pragma unroll
            for (int c = 0; c < 41; c++) A[c] = (c & 3) == 0 ? e[c >> 2] : __builtin_amdgcn_alignbyte(e[(c >> 2) + 1], e[c >> 2], (uint32_t)(c & 3));
            // stored transposed, rowp[c][r] (u16, pitch DRP): the column pass then reads vertical neighbours as packed pairs
            uint16_t* o = rowp + lane;
#pragma unroll
            for (int c = 0; c < DB; c++)
                o[c * DRP] = (uint16_t)__builtin_amdgcn_udot4(A[c], K0, __builtin_amdgcn_udot4(A[c + 4], K1, 0u, false), false);
        }
        for (int off = 32; off > 0; off >>= 1) {
            m10 += __shfl_xor(m10, off);
            m01 += __shfl_xor(m01, off);
        }
        if (lane == 0) { mom[2 * wave] = m01; mom[2 * wave + 1] = m10; }
    }
    PROF_MARK(1, 3);   // row reads + IC_Angle + row pass
    DESC_SYNC();
    PROF_MARK(1, 4);   // barrier 2
    // fastAtan2 and the double-precision sin/cos are wave-uniform work (~130 VALU instructions that every lane of every wave would repeat):
    // four lanes of wave 0 do them for the four keypoints while the block runs the column pass
    if (threadIdx.x < DESC_WPB && vflag[threadIdx.x]) {
        const float ang = fast_atan2_deg((float)mom[2 * threadIdx.x], (float)mom[2 * threadIdx.x + 1]);
        float sn, cs;
        det_sincos(ang * (float)(3.1415926535897932384626433832795 / 180.f), &sn, &cs);
        trig[3 * threadIdx.x] = ang; trig[3 * threadIdx.x + 1] = sn; trig[3 * threadIdx.x + 2] = cs;
    }
#if !DESC_SPARSE
    // column pass: a task filters DESC_CLEN rows of one column with v_dot2_u32_u16 on vertical pairs (segments start on even rows so that the
    // pair loads stay dword aligned; the row shared by two segments is written twice with the same value); out = (sum + 32768) >> 16, saturated.
    //   4 keypoints per block: 4 x 37 columns x 2 segments of 19 rows = 296 tasks over 256 threads (1.16 rounds of 19 rows);
    //   1 keypoint per block:      37 columns x 3 segments of 13 rows = 111 tasks over  64 lanes   (2 rounds of 13 rows; two segments would be
    //                               74 tasks = 2 rounds of 19 rows with 10 live lanes in the second).
#if DESC_WPB == 1
#define DESC_CSEG 3
#define DESC_CLEN 13
#else
#define DESC_CSEG 2
#define DESC_CLEN 19
#endif
    for (int t = threadIdx.x; t < (DESC_KO == 1 ? 0 : DESC_WPB * DESC_CSEG * DB); t += 64 * DESC_WPB) {
        const int w = t / (DESC_CSEG * DB), rem = t - w * (DESC_CSEG * DB);
        const int seg = rem / DB, col = rem - seg * DB;
        if (!vflag[w]) continue;
        const int r0 = seg * (DESC_CLEN - 1);
        const uint32_t* cp = (const uint32_t*)((const uint16_t*)(orb_smem + w * DESC_WAVE_STRIDE + DESC_ROWP_OFF) + col * DRP + r0);
        uint8_t* bl = orb_smem + w * DESC_WAVE_STRIDE + DESC_BLUR_OFF + r0 * DBP + col;
        typedef unsigned short u16x2 __attribute__((vector_size(4)));
        constexpr int NE = (DESC_CLEN + 7) / 2;
        uint32_t E[NE], O[NE - 1];   // E[k] = rows (r0+2k, r0+2k+1), O[k] = rows (r0+2k+1, r0+2k+2)
#pragma unroll
        for (int k = 0; k < NE; k++) E[k] = cp[k];
#pragma unroll
        for (int k = 0; k < NE - 1; k++) O[k] = __builtin_amdgcn_alignbyte(E[k + 1], E[k], 2u);
        const u16x2 Wa = {18, 34}, Wb = {49, 55}, Wc = {49, 34}, Wd = {18, 0};
#pragma unroll
        for (int j = 0; j < DESC_CLEN; j++) {
            const uint32_t* Q = (j & 1) ? O : E;
            const int m = j >> 1;
            uint32_t acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, Q[m]), Wa, 32768u, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, Q[m + 1]), Wb, acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, Q[m + 2]), Wc, acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, Q[m + 3]), Wd, acc, false);
            bl[j * DBP] = (uint8_t)min(acc >> 16, 255u);
        }
    }
#endif
    PROF_MARK(1, 5);   // trig (wave 0) + column pass
    DESC_SYNC();
    PROF_MARK(1, 6);   // barrier 3
    if (!valid) return;
    // rBRIEF (ORBextractor.cc:106-145): lane i evaluates pairs 4i..4i+3
    angle = trig[3 * wave];
    const float b = trig[3 * wave + 1], a = trig[3 * wave + 2];
    uint32_t nib = 0;
#pragma unroll
    for (int j = 0; j < (DESC_KO == 3 ? 0 : 4); j++) {
        // the two points of a pair are rotated together on packed float pairs (v_pk_mul_f32 / v_pk_add_f32: the same IEEE mul, mul, add per
        // component as the scalar form, no contraction)
        typedef float f32x2 __attribute__((vector_size(8)));
        const float4 pt = c_patternf[lane * 4 + j];
        const f32x2 X = {pt.x, pt.y}, Y = {pt.z, pt.w}, Bv = {b, b}, Av = {a, a};
        const f32x2 R = X * Bv + Y * Av, Q = X * Av - Y * Bv;
        const int r0 = __float2int_rn(R[0]), q0 = __float2int_rn(Q[0]);
        const int r1 = __float2int_rn(R[1]), q1 = __float2int_rn(Q[1]);
#if DESC_SPARSE
        // blurred(r, c) = sat8((sum_k g[k] * rowpass(r + k, c) + 32768) >> 16): seven u16 of one column of the transposed row-pass buffer.  The run
        // starts on either parity: five ALIGNED dwords from the even row below it, realigned by a per-lane v_alignbyte shift of 0 or 2 bytes
        // (a 14-byte load at 2-byte alignment is what the compiler makes of seven u16 reads: ds_read_b96 at odd offsets, 0.78 instead of 0.66 ms)
        typedef unsigned short u16x2 __attribute__((vector_size(4)));
        const u16x2 Wa = {18, 34}, Wb = {49, 55}, Wc = {49, 34}, Wd = {18, 0};
        auto blurred = [&](const int r, const int c) {
            const int rr = 18 + r;
            const uint32_t* cq = (const uint32_t*)(rowp + (18 + c) * DRP + (rr & ~1));   // DRP is even: dword aligned
            const uint32_t sh = (uint32_t)(rr & 1) * 2u;
            const uint32_t d0 = cq[0], d1 = cq[1], d2 = cq[2], d3 = cq[3];   // (the run's seventh value is the low or the high half of d3)
            uint32_t acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_alignbyte(d1, d0, sh)), Wa, 32768u, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_alignbyte(d2, d1, sh)), Wb, acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_alignbyte(d3, d2, sh)), Wc, acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_alignbyte(d3, d3, sh)), Wd, acc, false);   // (the other half has weight 0)
            return (int)min(acc >> 16, 255u);
        };
        const int t0 = blurred(r0, q0), t1 = blurred(r1, q1);
#else
        const int t0 = blur[(18 + r0) * DBP + 18 + q0], t1 = blur[(18 + r1) * DBP + 18 + q1];
#endif
        nib |= (uint32_t)(t0 < t1) << j;
    }
    // pack: byte = nibble(even lane) | nibble(odd lane) << 4 ; dword = 4 consecutive bytes
    uint32_t v = nib | (__shfl_xor(nib, 1) << 4);           // valid on even lanes
    const uint32_t b1 = __shfl_down(v, 2), b2 = __shfl_down(v, 4), b3 = __shfl_down(v, 6);
    const uint32_t dw = (v & 255) | ((b1 & 255) << 8) | ((b2 & 255) << 16) | ((b3 & 255) << 24);
    // output slot (ORBextractor.cc:1141-1152)
    const int rank = (int)(aux & 0x7FFFFFFF);
    const int idx = (aux >> 31) ? (nTotal - 1 - (lapBase + rank)) : (monoBase + rank);
    if (idx < 0 || idx >= P.cap) return;
    uint32_t* dout = (uint32_t*)(P.desc + ((size_t)frame * P.cap + idx) * 32);
    if ((lane & 7) == 0) dout[lane >> 3] = dw;
    if (lane < 7) {
        float fx = (float)cx, fy = (float)cy;
        if (level != 0) { fx = fx * L.scale; fy = fy * L.scale; }
        uint32_t w;
        switch (lane) {
            case 0: w = __float_as_uint(fx); break;
            case 1: w = __float_as_uint(fy); break;
            case 2: w = __float_as_uint(L.size); break;
            case 3: w = __float_as_uint(angle); break;
            case 4: w = __float_as_uint((float)(key >> 24)); break;
            case 5: w = (uint32_t)level; break;
            default: w = 0xFFFFFFFFu; break;
        }
        ((uint32_t*)(P.kps + (size_t)frame * P.cap + idx))[lane] = w;
    }
    PROF_MARK(1, 7);   // rBRIEF + outputs
    PROF_FLUSH(1);
}

// Two key points per wave (round 3).  The ~130 wave-uniform instructions of fastAtan2 + the double-precision sin / cos are a fifth of a key point's
// instructions and occupy ONE lane; here lanes 0 and 1 evaluate them for the wave's two key points at once.  Everything else runs one key point
// after the other on all lanes: both 43 x 48 patches are staged side by side, every lane reads its row of BOTH into registers (IC_Angle moments on
// the way), and only then the row pass of the first key point overwrites the patch region with its transposed row-pass buffer — the second one
// follows from registers after the first has been sampled.  LDS per wave: two patches = 4 472 B (the round-2 attempt at several key points per wave
// carried a blurred tile per key point and lost its instruction saving to occupancy; there is no blurred tile any more), 8 waves per SIMD.
#ifndef DESC_KPW
#define DESC_KPW 2   // key points per wave: 2 (k_describe2) or 1 (k_describe)
#endif
#define DESC2_PATCH (DP * DPP)                       // 2 236 B
#define DESC2_WAVE_BYTES (2 * DESC2_PATCH + 8)       // region A: two patches, later the 3 404-byte row-pass buffer
static __global__ __launch_bounds__(64, DESC_WAVES) void k_describe2(DescParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int lane = threadIdx.x & 63;
    int frame, grp;
    if (!xcd_frame_unit_m(P.groups * 2, P.groupsMagic, P.batch, &frame, &grp)) return;
    const int kpair = grp & 1;
    grp >>= 1;
    uint8_t* patch = orb_smem;
    uint16_t* rowp = (uint16_t*)orb_smem;
    int* mom = (int*)(orb_smem + DESC2_WAVE_BYTES);          // [2][2] m01, m10
    float* trig = (float*)(mom + 4);                         // [2][3] angle (degrees), sin, cos
    int level = 0, ustart = 0;
#pragma unroll
    for (int l = 1; l < ORBX_MAX_LEVELS; l++) { const int us = P.unitStart[l]; if (grp >= us) { level = l; ustart = us; } }
    const int pos0 = (grp - ustart) * 4 + 2 * kpair;         // the wave's key points are pos0 and pos0 + 1 of the level's slab
    const DescLevel& L = P.lv[level];
    uint32_t key[2] = {0, 0}, aux[2] = {0, 0};
#pragma unroll
    for (int s = 0; s < 2; s++)
        if (pos0 + s < L.selCap) {
            key[s] = P.sel[(size_t)frame * P.selFrame + L.selOff + pos0 + s];
            aux[s] = P.selAux[(size_t)frame * P.selFrame + L.selOff + pos0 + s];
        }
    // per-level counts and their prefix sums: wave-uniform (see k_describe)
    int nTotal = 0, monoTotal = 0, nLevel = 0, monoBase = 0, lapBase = 0;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIP_EMULATED)
    {
        const int __attribute__((address_space(4)))* cN = (const int __attribute__((address_space(4)))*)(unsigned long long)(P.selCount + (size_t)frame * P.nlevels);
        const int __attribute__((address_space(4)))* cL = (const int __attribute__((address_space(4)))*)(unsigned long long)(P.lapCount + (size_t)frame * P.nlevels);
#pragma unroll
        for (int blk = 0; blk < ORBX_MAX_LEVELS; blk += 8) {
            if (blk < P.nlevels) {
                int nn[8], ll[8];
#pragma unroll
                for (int l = 0; l < 8; l++) { nn[l] = cN[blk + l]; ll[l] = cL[blk + l]; }
#pragma unroll
                for (int l = 0; l < 8; l++) {
                    const int n = blk + l < P.nlevels ? nn[l] : 0, lp = blk + l < P.nlevels ? ll[l] : 0;
                    nTotal += n; monoTotal += n - lp;
                    monoBase += blk + l < level ? n - lp : 0;
                    lapBase += blk + l < level ? lp : 0;
                    nLevel = blk + l == level ? n : nLevel;
                }
            }
        }
    }
#else
    for (int l = 0; l < P.nlevels; l++) {
        const int n = P.selCount[(size_t)frame * P.nlevels + l], lp = P.lapCount[(size_t)frame * P.nlevels + l];
        nTotal += n; monoTotal += n - lp;
        if (l < level) { monoBase += n - lp; lapBase += lp; }
        if (l == level) nLevel = n;
    }
#endif
    if (grp == 0 && kpair == 0 && lane == 0) { P.counts[2 * frame] = nTotal; P.counts[2 * frame + 1] = monoTotal; }
    const bool valid[2] = {pos0 < L.selCap && pos0 < nLevel, pos0 + 1 < L.selCap && pos0 + 1 < nLevel};
    if (!valid[0]) return;                                   // (positions fill from the front: no second key point without a first)
    // IC_Angle's circular patch as byte masks of the lane's patch row (lane = row, |row - 21| <= 15), read BEFORE the patches are asked for: a
    // table read behind the staging barrier was a memory round trip of its own in the middle of every key point
    const int icv = lane - 21;
    uint32_t icm[9];
    {
        const int av = icv < 0 ? -icv : icv;
        const uint32_t* mk = c_icmask[min(av, 15)];
#pragma unroll
        for (int k = 0; k < 9; k++) icm[k] = av <= 15 ? mk[k + 1] : 0u;
    }
    int cx[2], cy[2], ox[2] = {0, 0};
    const uint8_t* img = L.base + (size_t)frame * L.frameStride;
    // ---- both patches -> LDS (all loads of a key point in flight before its first store; the second key point's follow the first's stores)
#pragma unroll
    for (int s = 0; s < 2; s++) {
        cx[s] = (int)(key[s] & 0xFFF) + ORBX_MINB; cy[s] = (int)((key[s] >> 12) & 0xFFF) + ORBX_MINB;
        if (!valid[s]) continue;
        uint8_t* pt = patch + s * DESC2_PATCH;
        const int xs = cx[s] - 21, ys = cy[s] - 21;
        if (xs >= 0 && ys >= 0 && cy[s] + 21 < L.h && cx[s] + 27 <= L.w) {
            const int x0 = xs & ~3;
            ox[s] = xs - x0;
            const int c = lane % 12, r5 = lane / 12;       // lanes 0..59 = 5 rows x 12 dwords per pass, 9 passes
            // 32-bit byte offsets inside the level (img is wave-uniform: scalar base + vector offset, no 64-bit multiply-adds per load)
            const uint32_t off0 = mul24((uint32_t)(ys + r5), (uint32_t)L.rowStride) + (uint32_t)(x0 + 4 * c);
            const uint32_t rs5 = 5u * (uint32_t)L.rowStride;
            uint8_t* dstp = pt + r5 * DPP + 4 * c;
            uint32_t v[9];
            if (lane < 60) {
#pragma unroll
                for (int k = 0; k < 9; k++)
                    if (k < 8 || r5 < 3) v[k] = *(const uint32_t*)(img + (off0 + (uint32_t)k * rs5));
#pragma unroll
                for (int k = 0; k < 9; k++)
                    if (k < 8 || r5 < 3) *(uint32_t*)(dstp + 5 * k * DPP) = v[k];
            }
        } else {   // the 7x7 blur taps may cross the image border: BORDER_REFLECT_101 at load
            for (int i = lane; i < DP * DP; i += 64) {
                const int r = i / DP, c = i - r * DP;
                pt[r * DPP + c] = img[(size_t)reflect101(ys + r, L.h) * L.rowStride + reflect101(xs + c, L.w)];
            }
        }
    }
    DESC_SYNC();
    // ---- lane r owns patch row r of both key points: packed rows into registers, IC_Angle moments (ORBextractor.cc:75-102) on the way
    uint32_t e[2][11];
    int m10[2] = {0, 0}, m01[2] = {0, 0};
    const int prow = min(lane, DP - 1);   // the lanes beyond the patch repeat its last row: their masks are 0 and the row pass leaves them out — no divergent
                                          // branch around the reads, no zeroed copy of the row for the lanes that would have skipped it
#pragma unroll
    for (int s = 0; s < 2; s++) {
        if (!valid[s]) {                  // wave-uniform
#pragma unroll
            for (int k = 0; k < 11; k++) e[s][k] = 0;
        } else {
            const uint32_t* rw = (const uint32_t*)(patch + s * DESC2_PATCH + prow * DPP);
            uint32_t d[12];
#pragma unroll
            for (int k = 0; k < 12; k++) d[k] = rw[k];
#pragma unroll
            for (int k = 0; k < 11; k++) e[s][k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], (uint32_t)ox[s]);
            {   // (rows outside the circle have an all-zero mask: both sums are 0)
                uint32_t s1 = 0, sw = 0;
#pragma unroll
                for (int k = 1; k <= 9; k++) {
                    uint32_t W = 0;
#pragma unroll
                    for (int t = 0; t < 4; t++) { const int j = 4 * k + t; if (j >= 6 && j <= 36) W |= (uint32_t)(j - 6) << (8 * t); }
                    const uint32_t m = e[s][k] & icm[k - 1];
                    s1 = __builtin_amdgcn_udot4(m, 0x01010101u, s1, false);
                    sw = __builtin_amdgcn_udot4(m, W, sw, false);
                }
                m10[s] = imul24((int)s1, -15) + (int)sw;     // (s1 <= 31 * 255: one v_mad_i32_i24)
                m01[s] = imul24(icv, (int)s1);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 2; s++) { m10[s] = wave_sum(m10[s]); m01[s] = wave_sum(m01[s]); }   // (register-only DPP steps: no LDS round trips)
    // fastAtan2 + sin / cos once for both key points: lane s serves key point s
    if (lane < 2) {
        const int my01 = lane == 0 ? m01[0] : m01[1], my10 = lane == 0 ? m10[0] : m10[1];
        const float ang = fast_atan2_deg((float)my01, (float)my10);
        float sn, cs;
        det_sincos(ang * (float)(3.1415926535897932384626433832795 / 180.f), &sn, &cs);
        trig[3 * lane] = ang; trig[3 * lane + 1] = sn; trig[3 * lane + 2] = cs;
    }
    DESC_SYNC();   // every lane's patch rows are in registers (the row-pass buffer may overwrite the patches), and trig[] is visible
    typedef unsigned short u16x2 __attribute__((vector_size(4)));
#pragma unroll
    for (int s = 0; s < 2; s++) {
        if (!valid[s]) break;
        if (s == 1) DESC_SYNC();   // the first key point's samples are read
        if (lane < DP) {
            // Gaussian row pass: k = {18,34,49,55,49,34,18}; out(c) = dot4(bytes c..c+3, k[0..3]) + dot4(bytes c+4..c+7, {k[4..6],0}); stored
            // transposed, rowp[c][r] (u16, pitch DRP)
            // on the ALIGNED dwords of the row, with the seven weights shifted to the column's byte phase instead of the bytes shifted to the
            // weights: 2 v_dot4 for phases 0 and 1 (the taps span two dwords), 3 for phases 2 and 3 — 92 per row instead of 74 + 30 v_alignbyte
            uint16_t* o = rowp + lane;
#pragma unroll
            for (int c = 0; c < DB; c++) {
                constexpr uint32_t g[7] = {18, 34, 49, 55, 49, 34, 18};
                const int q = c >> 2, ph = c & 3;
                uint32_t acc = 0;
#pragma unroll
                for (int dwi = 0; dwi < 3; dwi++) {
                    uint32_t Wd = 0;   // weights of the bytes of dword q + dwi: byte t is tap 4 * dwi + t - ph
#pragma unroll
                    for (int t = 0; t < 4; t++) { const int tap = 4 * dwi + t - ph; if (tap >= 0 && tap < 7) Wd |= g[tap] << (8 * t); }
                    if (Wd != 0) acc = __builtin_amdgcn_udot4(e[s][q + dwi], Wd, acc, false);
                }
                o[c * DRP] = (uint16_t)acc;
            }
        }
        DESC_SYNC();
        // rBRIEF (ORBextractor.cc:106-145): lane i evaluates pairs 4i..4i+3; the blur's column pass only where it samples (see k_describe)
        const float angle = trig[3 * s], b = trig[3 * s + 1], a = trig[3 * s + 2];
        uint32_t nib = 0;
        const u16x2 Wa = {18, 34}, Wb = {49, 55}, Wc = {49, 34}, Wd = {18, 0};
        // byte address of the run's aligned start inside the row-pass buffer: column 18 + c is (18 + c) * 2 * DRP bytes in, row 18 + r another
        // 2 * (18 + r), rounded down to a dword.  r and c arrive as rint_bits (RINT_BIAS + value): c * 2 * DRP + K is ONE v_mad_i32_i24 on the raw
        // bits (their low 24 are 0x400000 + c), 2 * r one add, and K takes every constant — the biases (2 * RINT_BIAS and 36 = 2 * 18 are multiples
        // of 4: they pass the "& ~3" untouched and leave the two low bits alone) — which the compiler must not see through: folded into the reads'
        // immediate offsets (1 692 bytes: beyond ds_read2's reach) the constant cost a second address register per point
        uint32_t K = (uint32_t)(18 * 2 * DRP + 36) - 0x400000u * (uint32_t)(2 * DRP) - 2u * RINT_BIAS;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIP_EMULATED)
        asm volatile("" : "+s"(K));
#endif
        auto blurred = [&](const uint32_t rb, const uint32_t cb) {
            const uint32_t r2 = rb + rb;
            const uint32_t* cq = (const uint32_t*)((const uint8_t*)rowp + ((uint32_t)imul24((int)cb, 2 * DRP) + K + (r2 & ~3u)));
            const uint32_t sh = r2 & 2u;
            const uint32_t d0 = cq[0], d1 = cq[1], d2 = cq[2], d3 = cq[3];   // (the run's seventh value is the low or the high half of d3)
            uint32_t acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_alignbyte(d1, d0, sh)), Wa, 32768u, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_alignbyte(d2, d1, sh)), Wb, acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_alignbyte(d3, d2, sh)), Wc, acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_alignbyte(d3, d3, sh)), Wd, acc, false);   // (the other half has weight 0)
            return (int)min(acc >> 16, 255u);
        };
#pragma unroll
        for (int j = 0; j < 4; j++) {
            typedef float f32x2 __attribute__((vector_size(8)));
            const float4 pt = c_patternf[lane * 4 + j];
            const f32x2 X = {pt.x, pt.y}, Y = {pt.z, pt.w}, Bv = {b, b}, Av = {a, a};
            const f32x2 R = X * Bv + Y * Av, Q = X * Av - Y * Bv;
            // cvRound of the rotated coordinates (ORBextractor.cc:120-123; |coordinates| <= 19)
            nib |= (uint32_t)(blurred(rint_bits(R[0]), rint_bits(Q[0])) < blurred(rint_bits(R[1]), rint_bits(Q[1]))) << j;
        }
        // eight lanes' nibbles -> one descriptor dword in the first of them, by DPP moves inside the row (quad_perm [1,0,3,2], row_shl:2 / 4 / 6)
        const uint32_t v = nib | ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)nib, 0xB1, 0xF, 0xF, false) << 4);           // valid on even lanes
        const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xF, 0xF, false), b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0xF, false),
                       b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x106, 0xF, 0xF, false);
        const uint32_t dw = (v & 255) | ((b1 & 255) << 8) | ((b2 & 255) << 16) | ((b3 & 255) << 24);
        // output slot (ORBextractor.cc:1141-1152)
        const int rank = (int)(aux[s] & 0x7FFFFFFF);
        const int idx = (aux[s] >> 31) ? (nTotal - 1 - (lapBase + rank)) : (monoBase + rank);
        if (idx >= 0 && idx < P.cap) {
            uint32_t* dout = (uint32_t*)(P.desc + ((size_t)frame * P.cap + idx) * 32);
            if ((lane & 7) == 0) dout[lane >> 3] = dw;
            if (lane < 7) {
                float fx = (float)cx[s], fy = (float)cy[s];
                if (level != 0) { fx = fx * L.scale; fy = fy * L.scale; }
                uint32_t w;
                switch (lane) {
                    case 0: w = __float_as_uint(fx); break;
                    case 1: w = __float_as_uint(fy); break;
                    case 2: w = __float_as_uint(L.size); break;
                    case 3: w = __float_as_uint(angle); break;
                    case 4: w = __float_as_uint((float)(key[s] >> 24)); break;
                    case 5: w = (uint32_t)level; break;
                    default: w = 0xFFFFFFFFu; break;
                }
                ((uint32_t*)(P.kps + (size_t)frame * P.cap + idx))[lane] = w;
            }
        }
    }
}

struct Desc { uint32_t w[8]; };
static __device__ __forceinline__ Desc load_desc16(const uint8_t* p) {
    Desc d;
    const uint4 a = *(const uint4*)p, b = *(const uint4*)(p + 16);
    d.w[0] = a.x; d.w[1] = a.y; d.w[2] = a.z; d.w[3] = a.w; d.w[4] = b.x; d.w[5] = b.y; d.w[6] = b.z; d.w[7] = b.w;
    return d;
}

// ============================================================================================================
// M9  Frame::ComputeStereoMatches (Frame.cc:955-1133): rectified stereo association.  Left keypoints are independent:
//     one wave per left keypoint scans the right keypoints in index order (row band +-2*scale, octave +-1, disparity range),
//     keeps the first minimum Hamming distance, then refines with the 11x11 L1 patch match over +-5 px on the pyramid
//     level of the left keypoint (parabola sub-pixel).  A second kernel applies the 1.5*1.4*median SAD cull per frame.
// ============================================================================================================
struct StereoParams {
    DescLevel lvL[ORBX_MAX_LEVELS], lvR[ORBX_MAX_LEVELS];   // pyramid views (base, strides, w, h, scale)
    float invScale[ORBX_MAX_LEVELS];
    const orb_keypoint *kpsL, *kpsR; const uint8_t *descL, *descR; const int32_t *cntL, *cntR;
    int cap; float mb, mbf; float* uRight; float* depth; int32_t* sad; int nRows;
    int32_t* rowStart; int32_t* rowIdx; int rowCap;   // vRowIndices as CSR per frame: [nRows + 1], [rowCap] right keypoint indices (ascending per row)
};

static __device__ __forceinline__ int stereo_pix(const DescLevel& L, int frame, int y, int x) {
    // mvImagePyramid[level] is the ROI of a BORDER_REFLECT_101 parent: reads a few pixels outside the ROI see the reflection
    y = reflect101(y, L.h); x = reflect101(x, L.w);
    return L.base[(size_t)frame * L.frameStride + (size_t)y * L.rowStride + x];
}

// vRowIndices (Frame.cc:972-982): every right keypoint is listed in the rows [floor(y - r), ceil(y + r)], r = 2 * scale[octave], in
// increasing keypoint index.  One workgroup per frame: LDS counting sort over the image rows, insertion order restored per row.
static __global__ __launch_bounds__(256) void k_stereo_rows(StereoParams P, const int ldsCap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    int* cnt = (int*)orb_smem;             // [nRows] counts -> starts
    int* fill = cnt + P.nRows;             // [nRows]
    int* scratch = fill + P.nRows;         // [256]
    int* lbuf = scratch + 256;             // [ldsCap] the frame's row lists while they are filled and ordered (round 3): the per-row insertion sort
                                           // was a chain of dependent global loads and stores (0.21 ms per 512 frames, one workgroup per frame)
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int Nr = min(P.cntR[2 * frame], P.cap);
    const orb_keypoint* kps = P.kpsR + (size_t)frame * P.cap;
    int32_t* rs = P.rowStart + (size_t)frame * (P.nRows + 1);
    int32_t* ri = P.rowIdx + (size_t)frame * P.rowCap;
    for (int r = tid; r < P.nRows; r += 256) { cnt[r] = 0; fill[r] = 0; }
    __syncthreads();
    for (int i = tid; i < Nr; i += 256) {
        const orb_keypoint kp = kps[i];
        const float r = 2.0f * P.lvL[kp.octave].scale;
        const int maxr = min((int)ceilf(kp.y + r), P.nRows - 1), minr = max((int)floorf(kp.y - r), 0);
        for (int y = minr; y <= maxr; y++) atomicAdd(&cnt[y], 1);
    }
    __syncthreads();
    int total;
    {
        const int per = (P.nRows + 255) / 256, s0 = tid * per, s1 = min(s0 + per, P.nRows);
        int sum = 0;
        for (int k = s0; k < s1; k++) sum += cnt[k];
        scratch[tid] = sum;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int v = tid >= off ? scratch[tid - off] : 0;
            __syncthreads();
            scratch[tid] += v;
            __syncthreads();
        }
        total = scratch[255];
        int run = scratch[tid] - sum;
        for (int k = s0; k < s1; k++) { const int c = cnt[k]; cnt[k] = run; run += c; }
        if (tid == 255) rs[P.nRows] = min(run, P.rowCap);
    }
    __syncthreads();
    for (int r = tid; r < P.nRows; r += 256) rs[r] = min(cnt[r], P.rowCap);
    const bool staged = total <= ldsCap && total <= P.rowCap;   // (workgroup-uniform)
    auto fill_and_order = [&](int* buf) {
        for (int i = tid; i < Nr; i += 256) {
            const orb_keypoint kp = kps[i];
            const float r = 2.0f * P.lvL[kp.octave].scale;
            const int maxr = min((int)ceilf(kp.y + r), P.nRows - 1), minr = max((int)floorf(kp.y - r), 0);
            for (int y = minr; y <= maxr; y++) { const int p = cnt[y] + atomicAdd(&fill[y], 1); if (p < P.rowCap) buf[p] = i; }
        }
        __threadfence_block();
        __syncthreads();
        for (int r = tid; r < P.nRows; r += 256) {   // restore ascending keypoint order inside each row (rows hold a few dozen entries)
            const int s = cnt[r], m = min(fill[r], max(P.rowCap - s, 0));
            for (int a = 1; a < m; a++) {
                const int v = buf[s + a];
                int p = a - 1;
                while (p >= 0 && buf[s + p] > v) { buf[s + p + 1] = buf[s + p]; p--; }
                buf[s + p + 1] = v;
            }
        }
    };
    if (staged) {
        fill_and_order(lbuf);
        __syncthreads();
        for (int p = tid; p < total; p += 256) ri[p] = lbuf[p];
    } else {
        fill_and_order(ri);
    }
}

static __global__ __launch_bounds__(256) void k_stereo_match(StereoParams P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frame = blockIdx.y, iL = blockIdx.x * 4 + wave;
    const int N = min(P.cntL[2 * frame], P.cap);
    if (iL >= P.cap) return;
    float* uRo = P.uRight + (size_t)frame * P.cap + iL;
    float* dpo = P.depth + (size_t)frame * P.cap + iL;
    int32_t* sado = P.sad + (size_t)frame * P.cap + iL;
    if (iL >= N) { if (lane == 0) { *uRo = -1.0f; *dpo = -1.0f; *sado = -1; } return; }
    const orb_keypoint kpL = P.kpsL[(size_t)frame * P.cap + iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const float minZ = P.mb, minD = 0.f, maxD = P.mbf / minZ;
    const float minU = uL - maxD, maxU = uL - minD;
    float outU = -1.0f, outD = -1.0f; int outS = -1;
    const int rowL = (int)vL;   // vRowIndices[vL]
    uint32_t best = 0xFFFFFFFFu;
    if (!(maxU < 0) && rowL >= 0 && rowL < P.nRows) {
        const Desc dL = load_desc16(P.descL + ((size_t)frame * P.cap + iL) * 32);
        // candidates = vRowIndices[vL] (Frame.cc:1004), ascending right keypoint index
        const int32_t* rs = P.rowStart + (size_t)frame * (P.nRows + 1);
        const int32_t* ri = P.rowIdx + (size_t)frame * P.rowCap;
        const int c0 = rs[rowL], c1 = rs[rowL + 1];
        for (int base = c0; base < c1; base += 64) {
            const int c = base + lane;
            if (c < c1) {
                const int iR = ri[c];
                const orb_keypoint kpR = P.kpsR[(size_t)frame * P.cap + iR];
                if (!(kpR.octave < levelL - 1 || kpR.octave > levelL + 1) && kpR.x >= minU && kpR.x <= maxU) {
                    const Desc dR = load_desc16(P.descR + ((size_t)frame * P.cap + iR) * 32);
                    int dist = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) dist += __popc(dL.w[k] ^ dR.w[k]);
                    best = min(best, ((uint32_t)dist << 16) | (uint32_t)iR);   // first (lowest iR) minimum wins, strict '<'
                }
            }
        }
    }
    best = wave_min_u32(best);
    const int bestDist = best == 0xFFFFFFFFu ? 256 : (int)(best >> 16);
    if (bestDist < ORBM_TH_HIGH && bestDist < (ORBM_TH_HIGH + ORBM_TH_LOW) / 2) {
        const int bestIdxR = (int)(best & 0xFFFF);
        const float uR0 = P.kpsR[(size_t)frame * P.cap + bestIdxR].x;
        const float scaleFactor = P.invScale[levelL];
        const int scaleduL = (int)roundf(kpL.x * scaleFactor), scaledvL = (int)roundf(kpL.y * scaleFactor);
        const int scaleduR0 = (int)roundf(uR0 * scaleFactor);
        const DescLevel& PL = P.lvL[levelL];
        const DescLevel& PR = P.lvR[levelL];
        const int w = 5, Lw = 5;
        const float iniu = (float)(scaleduR0 + Lw - w), endu = (float)(scaleduR0 + Lw + w + 1);
        if (!(iniu < 0 || endu >= (float)PR.w)) {
            // lanes own patch pixels p = lane and lane + 64 (121 pixels)
            const int cLv = stereo_pix(PL, frame, scaledvL, scaleduL);
            int il[2], pa[2], pb[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int p = lane + 64 * t;
                pa[t] = p / 11; pb[t] = p - pa[t] * 11;
                il[t] = p < 121 ? stereo_pix(PL, frame, scaledvL - w + pa[t], scaleduL - w + pb[t]) - cLv : 0;
            }
            // the 11 right patches are shifted copies of one 11 x 21 strip of the right image: staged once per keypoint in LDS
            extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
            uint8_t* strip = orb_smem + wave * (11 * 24);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int p = lane + 64 * t;
                if (p < 11 * 21) { const int r = p / 21, c = p - r * 21; strip[r * 24 + c] = (uint8_t)stereo_pix(PR, frame, scaledvL - w + r, scaleduR0 - Lw - w + c); }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            float vDists[11];
            int bestSad = 2147483647, bestincR = 0;
            // a lane's contributions to the eleven SADs first, two per register (a SAD is at most 120 * 510 = 61 200 < 2^16, so the halves never carry;
            // the sums are unsigned: the upper half alone exceeds INT_MAX), then six wave sums (rounds 1-2: eleven butterflies of six dependent ds_bpermute each — the kernel was that chain)
            uint32_t part[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int inc = 0; inc < 11; inc++) {
                const int cRv = strip[w * 24 + inc + w];
                int sum = 0;
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int p = lane + 64 * t;
                    if (p < 121) {
                        const int v = (int)strip[pa[t] * 24 + inc + pb[t]] - cRv;
                        const int dd = il[t] - v;
                        sum += dd < 0 ? -dd : dd;
                    }
                }
                part[inc >> 1] |= (uint32_t)sum << (16 * (inc & 1));
            }
#pragma unroll
            for (int k = 0; k < 6; k++) part[k] = wave_sum_u32(part[k]);
#pragma unroll
            for (int inc = 0; inc < 11; inc++) {
                const int incR = inc - Lw;
                const float dist = (float)(int)((part[inc >> 1] >> (16 * (inc & 1))) & 0xFFFFu);
                if (dist < (float)bestSad) { bestSad = (int)dist; bestincR = incR; }
                vDists[inc] = dist;
            }
            __builtin_amdgcn_wave_barrier();
            if (!(bestincR == -Lw || bestincR == Lw)) {
                float dist1 = 0, dist2 = 0, dist3 = 0;
#pragma unroll
                for (int inc = 1; inc < 10; inc++)
                    if (inc == Lw + bestincR) { dist1 = vDists[inc - 1]; dist2 = vDists[inc]; dist3 = vDists[inc + 1]; }
                const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = PL.scale * ((float)scaleduR0 + (float)bestincR + deltaR);
                    float disparity = uL - bestuR;
                    if (disparity >= minD && disparity < maxD) {
                        if (disparity <= 0) { disparity = (float)0.01; bestuR = (float)((double)uL - 0.01); }
                        outD = P.mbf / disparity; outU = bestuR; outS = bestSad;
                    }
                }
            }
        }
    }
    if (lane == 0) { *uRo = outU; *dpo = outD; *sado = outS; }
}

// median SAD cull (Frame.cc:1119-1132): sort (dist, iL); median = element size/2; drop everything with dist >= 1.5*1.4*median
#ifndef STEREO_ROWS_LDS_MAX
#define STEREO_ROWS_LDS_MAX (64 * 1024)   // dynamic LDS of k_stereo_rows (tests build with 0: every frame takes the global-memory path)
#endif
#define STEREO_CULL_BINS 2048   // SAD >> 5: the patches are centre-subtracted (Frame.cc:1055-1075, IL - IL.at(w, w)), so a pixel contributes up to 510 and the
                                // centre itself 0: a SAD is at most 120 * 510 = 61 200 (bin 1 912); nothing is clamped into a shared last bin
static __global__ __launch_bounds__(256) void k_stereo_cull(StereoParams P) {
    // median of the valid SADs = the value at rank size / 2 of the ascending order (Frame.cc:1120-1123: sort of (dist, index) pairs — ties do not change
    // the VALUE at a rank), by a two-level histogram: 2 048 bins of 32 values, then the 32 values of the bin that holds the rank.  (Rounds 1-2 ranked
    // every value against every other from an LDS copy: 4 000 turns per thread at 1 000 key points, 0.18 ms per 512 frames — and cap + 2 words of LDS.)
    // 2 048 bins cover 65 535 >= the largest possible SAD, so min(s >> 5, BINS - 1) never merges distinct bins and `s & 31` is the rank inside one.
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    int* hist = (int*)orb_smem;               // [STEREO_CULL_BINS]
    int* sub = hist + STEREO_CULL_BINS;       // [32]
    int* scratch = sub + 32;                  // [256]
    int* ctl = scratch + 256;                 // [0] bin of the rank, [1] rank inside the bin, [2] the median
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int N = min(P.cntL[2 * frame], P.cap);
    const int32_t* sad = P.sad + (size_t)frame * P.cap;
    for (int k = tid; k < STEREO_CULL_BINS + 32; k += 256) hist[k] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 256) { const int s = sad[i]; if (s >= 0) atomicAdd(&hist[min(s >> 5, STEREO_CULL_BINS - 1)], 1); }
    __syncthreads();
    {
        constexpr int per = STEREO_CULL_BINS / 256;
        int sum = 0;
        for (int k = 0; k < per; k++) sum += hist[tid * per + k];
        scratch[tid] = sum;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int v = tid >= off ? scratch[tid - off] : 0;
            __syncthreads();
            scratch[tid] += v;
            __syncthreads();
        }
        const int size = scratch[255];
        if (size == 0) return;   // (the reference reads vDistIdx[0] unconditionally; nothing to cull here)  — workgroup-uniform
        const int target = size / 2;
        int run = scratch[tid] - sum;
        for (int k = 0; k < per; k++) {
            const int c = hist[tid * per + k];
            if (target >= run && target < run + c) { ctl[0] = tid * per + k; ctl[1] = target - run; }
            run += c;
        }
    }
    __syncthreads();
    const int bin = ctl[0];
    for (int i = tid; i < N; i += 256) { const int s = sad[i]; if (s >= 0 && min(s >> 5, STEREO_CULL_BINS - 1) == bin) atomicAdd(&sub[s & 31], 1); }
    __syncthreads();
    if (tid == 0) {
        int run = 0, v = 0;
        for (int k = 0; k < 32; k++) { if (ctl[1] >= run && ctl[1] < run + sub[k]) v = k; run += sub[k]; }
        ctl[2] = (bin << 5) | v;
    }
    __syncthreads();
    const float median = (float)ctl[2];
    const float thDist = 1.5f * 1.4f * median;
    for (int i = tid; i < N; i += 256) {
        const int s = sad[i];
        if (s >= 0 && !((float)s < thDist)) { P.uRight[(size_t)frame * P.cap + i] = -1.0f; P.depth[(size_t)frame * P.cap + i] = -1.0f; }
    }
}

// ============================================================================================================
// Host side
// ============================================================================================================
struct LevelHost {
    int w, h, stride; size_t planeOff, planeBytes;   // levels >= 1 live in the handle's pyramid slab
    int maxBX, maxBY, nCols, nRows, wCell, hCell, nIni; float hX;
    int candCap; size_t candOff; int selCap, selOff;
};

struct orbx_extractor {
    orbx_config cfg; int W, H, maxBatch, device;
    std::vector<float> scale, invScale, sigma2, invSigma2; std::vector<int> nfeat; int umax[16];
    LevelHost lv[ORBX_MAX_LEVELS];
    size_t pyrFrame = 0, candFrame = 0; int selFrame = 0, nodeCap = 0, maxKp = 0;
    int nTiles = 0, nTiles1 = 0, fastImgBytes = 0, octKeyOff = 0, octMerge = 0; size_t fastSmem = 0, octSmem = 0;
    hipStream_t stream = nullptr;
    int32_t* d_rowStart = nullptr; int32_t* d_rowIdx = nullptr; int rowCapAlloc = 0;   // ComputeStereoMatches row buckets (lazy)
    int4* d_chain = nullptr; int chainTiles = 0, chainBuf = 0; size_t chainSmem = 0;   // k_pyramid_chain's regions ([tiles][nlevels]); chainTiles = 0: not usable for this geometry
    int* d_coef = nullptr; size_t coefOff[ORBX_MAX_LEVELS] = {0};   // k_resize2 tables of every level >= 1
    size_t tileTabOff[ORBX_MAX_LEVELS] = {0};                        // k_resize2 staging footprints of every level >= 1 (inside d_coef)
    uint8_t* d_pyr = nullptr; uint32_t* d_cand = nullptr; int* d_candCount = nullptr; uint16_t* d_keyNode = nullptr;
    uint32_t *d_sel = nullptr, *d_selAux = nullptr; int *d_selCount = nullptr, *d_lapCount = nullptr;
    FastTile* d_tiles = nullptr; uint32_t* d_retry = nullptr; int* d_order = nullptr;
    uint32_t* h_retry = nullptr;   // pinned: [0] tiles the last two-pass FAST call listed for its second pass (copied back asynchronously), [1] tiles of that call,
                                   // [2] the largest FAST candidate count of any (frame, level) of the call before last (k_frame_order writes it)
    uint32_t fastCalls = 0; int fastLastTwoPass = 0, lastOrdered = 0, lastHeavy = 0;
    // single-image staging
    uint8_t* d_img = nullptr; int imgStride = 0; orb_keypoint* d_kps1 = nullptr; uint8_t* d_desc1 = nullptr; int32_t* d_counts1 = nullptr;
    // single-frame host call: the three outputs share ONE device block ([counts | keypoints | descriptors]; the pointers above point into it) so that
    // one copy brings them back, and both directions go through pinned staging buffers — the whole call is one graph launch (H2D, kernels, D2H) and
    // one synchronisation, with plain memcpy on the host side (a pageable-memory copy is staged and synchronous inside the runtime: three of them
    // and two synchronisations were half of the call's 205 us)
    uint8_t* d_out1 = nullptr; uint8_t* h_img = nullptr; uint8_t* h_out = nullptr; size_t out1Bytes = 0, kps1Off = 0, desc1Off = 0;
    // last call (debug taps / pyramid views)
    const uint8_t* lastImages = nullptr; size_t lastFrameStride = 0; int lastRowStride = 0, lastBatch = 0;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; bool timed = false, timing = false;
    // single-frame entry point: the launch sequence of one call is captured once into a HIP graph (per lapping window) and replayed —
    // a frame is ~11 kernels + 2 memsets of a few tens of microseconds each, i.e. launch bound
    hipGraphExec_t graphExec = nullptr; int graphLap0 = 0, graphLap1 = 0; int graphState = 0;   // 0 = not tried, 1 = usable, -1 = capture unsupported
    bool capturing = false;
    // Host-pyramid option of the single-frame entry point (orbx_set_host_pyramid; mvImagePyramid consumers): all levels in the reference's bordered
    // layout in one device slab (k_border_pyramid) and its pinned twin, filled on a second stream that forks behind the pyramid launch and joins at
    // the end of the call — the copy runs under FAST / octree / describe.  Allocated by the first orbx_set_host_pyramid(h, 1).
    int hostPyr = 0, graphHostPyr = 0; bool hostCall = false, hostPyrValid = false;
    uint8_t* d_bpyr = nullptr; uint8_t* h_bpyr = nullptr; size_t bpyrBytes = 0, bOff[ORBX_MAX_LEVELS] = {0};
    int bPitch[ORBX_MAX_LEVELS] = {0}, bRowStart[ORBX_MAX_LEVELS + 1] = {0};
    hipStream_t stream2 = nullptr; hipEvent_t evFork = nullptr, evJoin = nullptr;
    hipStream_t lastStream = nullptr;   // the stream of the last batch call (orbx_copy_level and the debug taps wait for it, not for the device)
    bool lastSingle = false;            // the last call was orbx_extract: d_kps1 / d_desc1 / d_counts1 hold its outputs (orbx_stereo_matches_last)
    float* d_stereo1 = nullptr; float* h_stereo1 = nullptr;   // [u_right | depth | work] x maxKp of orbx_stereo_matches_last, device and pinned
    std::string err;
};

// error text of the calling thread's last failed orbx_create (no handle exists to hold it): per thread, so that two threads creating extractors
// concurrently — the reference builds its left / right extractors from Tracking's constructor, hosts may do it anywhere — never share a std::string
static thread_local std::string g_create_err;

static inline int cvRoundF(float v) { return (int)lrintf(v); }
static inline int cvRoundD(double v) { return (int)lrint(v); }

#define HIPCHK(h, call)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                \
            return ORB_E_HIP;                                                                            \
        }                                                                                                \
    } while (0)

static int orbx_fail(orbx_extractor* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_err = msg;
    return code;
}

static void orbx_free(orbx_extractor* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    void* bufs[] = {h->d_chain, h->d_rowStart, h->d_rowIdx, h->d_coef, h->d_pyr, h->d_cand, h->d_candCount, h->d_keyNode, h->d_sel, h->d_selAux, h->d_selCount,
                    h->d_lapCount, h->d_tiles, h->d_retry, h->d_img, h->d_out1, h->d_bpyr, h->d_stereo1, h->d_order};
    for (void* p : bufs) if (p) (void)hipFree(p);
    if (h->h_bpyr) (void)hipHostFree(h->h_bpyr);
    if (h->h_stereo1) (void)hipHostFree(h->h_stereo1);
    if (h->evFork) (void)hipEventDestroy(h->evFork);
    if (h->evJoin) (void)hipEventDestroy(h->evJoin);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->h_img) (void)hipHostFree(h->h_img);
    if (h->h_retry) (void)hipHostFree(h->h_retry);
    if (h->h_out) (void)hipHostFree(h->h_out);
    for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->graphExec) (void)hipGraphExecDestroy(h->graphExec);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int orbx_create(const orbx_config* cfg, int width, int height, int max_batch, int device, orbx_handle* out) {
    if (!cfg || !out) return orbx_fail(nullptr, ORB_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->nlevels < 1 || cfg->nlevels > ORBX_MAX_LEVELS || cfg->nfeatures < 1 || !(cfg->scale_factor > 1.0f) ||
        width < 1 || height < 1 || width > 4095 + 2 * ORBX_MINB || height > 4095 + 2 * ORBX_MINB || max_batch < 1)
        return orbx_fail(nullptr, ORB_E_INVALID, "bad configuration");
    orbx_extractor* h = new orbx_extractor();
    h->cfg = *cfg; h->W = width; h->H = height; h->maxBatch = max_batch; h->device = device;
    const int nl = cfg->nlevels;
    // ---- scale tables, features per level, umax: ORBextractor.cc:408-468 (scaleFactor is a double member)
    const double sf = (double)cfg->scale_factor;
    h->scale.resize(nl); h->invScale.resize(nl); h->sigma2.resize(nl); h->invSigma2.resize(nl); h->nfeat.resize(nl);
    h->scale[0] = 1.0f; h->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) { h->scale[i] = (float)(h->scale[i - 1] * sf); h->sigma2[i] = h->scale[i] * h->scale[i]; }
    for (int i = 0; i < nl; i++) { h->invScale[i] = 1.0f / h->scale[i]; h->invSigma2[i] = 1.0f / h->sigma2[i]; }
    {
        float factor = (float)(1.0f / sf);
        float nDesired = cfg->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
        int sum = 0;
        for (int l = 0; l < nl - 1; l++) { h->nfeat[l] = cvRoundF(nDesired); sum += h->nfeat[l]; nDesired *= factor; }
        h->nfeat[nl - 1] = std::max(cfg->nfeatures - sum, 0);
        int v, v0, vmax = (int)std::floor(15 * std::sqrt(2.f) / 2 + 1), vmin = (int)std::ceil(15 * std::sqrt(2.f) / 2);
        for (v = 0; v <= vmax; ++v) h->umax[v] = cvRoundD(std::sqrt(225.0 - v * v));
        for (v = 15, v0 = 0; v >= vmin; --v) { while (h->umax[v0] == h->umax[v0 + 1]) ++v0; h->umax[v] = v0; ++v0; }
    }
    // ---- per-level geometry: ORBextractor.cc:1162-1163 (sizes), :769-785 (cells), :541-543 (octree roots)
    std::vector<FastTile> tiles, tiles1;   // two-cell-row tiles (batches) / one-cell-row tiles (few frames)
    size_t pyrOff = 0, candOff = 0; int selOff = 0, maxRows = 0, maxPitch = 0, nodeCap = 0, maxKp = 0;
    for (int l = 0; l < nl; l++) {
        LevelHost& L = h->lv[l];
        L.w = cvRoundF((float)width * h->invScale[l]);
        L.h = cvRoundF((float)height * h->invScale[l]);
        L.stride = (L.w + 63) & ~63;
        L.planeBytes = (size_t)L.stride * L.h;
        L.planeOff = pyrOff;
        if (l > 0) pyrOff += (L.planeBytes + 255) & ~(size_t)255;
        L.maxBX = L.w - ORBX_MINB; L.maxBY = L.h - ORBX_MINB;
        const float fw = (float)(L.maxBX - ORBX_MINB), fh = (float)(L.maxBY - ORBX_MINB);
        L.nCols = (int)(fw / 30.f); L.nRows = (int)(fh / 30.f);
        if (L.nCols < 1 || L.nRows < 1) { orbx_free(h); return orbx_fail(nullptr, ORB_E_INVALID, "image too small for the requested number of pyramid levels"); }
        L.wCell = (int)std::ceil(fw / L.nCols); L.hCell = (int)std::ceil(fh / L.nRows);
        L.nIni = (int)std::round(fw / (L.maxBY - ORBX_MINB));
        if (L.nIni < 1 || L.nCols > 4000) { orbx_free(h); return orbx_fail(nullptr, ORB_E_INVALID, "unsupported aspect ratio (octree needs width/height >= 0.5)"); }
        L.hX = fw / L.nIni;
        // candidate bound: NMS survivors per cell <= ceil(w/2)*ceil(h/2)
        int cap = 0;
        const int detWtot = L.w - 2 * ORBX_EDGE, detHtot = L.h - 2 * ORBX_EDGE;
        for (int i = 0; i < L.nRows; i++) {
            const int hh = std::min(L.hCell, detHtot - i * L.hCell);
            if (hh <= 0) continue;
            for (int j = 0; j < L.nCols; j++) {
                const int ww = std::min(L.wCell, detWtot - j * L.wCell);
                if (ww > 0) cap += ((ww + 1) / 2) * ((hh + 1) / 2);
            }
        }
        L.candCap = std::max(cap, 1); L.candOff = candOff; candOff += (size_t)((L.candCap + 3) & ~3);
        L.selCap = std::max(h->nfeat[l], 4 * L.nIni) + 4; L.selOff = selOff; selOff += L.selCap;
        nodeCap = std::max(nodeCap, L.selCap); maxKp += L.selCap;
        if (L.selCap > 60000) { orbx_free(h); return orbx_fail(nullptr, ORB_E_INVALID, "nfeatures per level too large"); }
        // FAST tiles: whole cells, <= FAST_TW px wide, FAST_CROWS cell rows high (the last tile of a level takes what is left)
        const int cpt = std::max(1, std::min(FAST_MAXCELLS / FAST_CROWS, FAST_TW / L.wCell));
        const int nT = (L.nCols + cpt - 1) / cpt, per = (L.nCols + nT - 1) / nT;
        int validRows = 0;
        for (int i = 0; i < L.nRows; i++) if (ORBX_MINB + i * L.hCell < L.maxBY - 3) validRows = i + 1;
        // Two cell rows per tile where they fit the 64 detection rows of k_fast's row mask, and only on the finer half of the pyramid: corner
        // density grows with the level (on the benchmark's frames a quarter of the pixels of levels 6-7 are FAST corners at threshold 7), and a
        // two-row tile's corner list holds 23 % of its pixels — past that the tile takes the whole-tile fallback, which is correct but slow
        // (measured: two-row tiles on all levels 1.12 instead of 0.80 ms per 512 frames, all of it fallback tiles of levels 6-7).
        const int crows = (FAST_CROWS * L.hCell <= 64 && 2 * l < nl) ? FAST_CROWS : 1;
        for (int i = 0; i < validRows; i += crows) {
            const int ncr = std::min(crows, validRows - i);
            for (int c0 = 0; c0 < L.nCols; c0 += per) {
                const int n = std::min(per, L.nCols - c0);
                if (ORBX_MINB + c0 * L.wCell >= L.maxBX - 6) continue;
                tiles.push_back(FastTile{(short)(l | (ncr << 8)), (short)i, (short)c0, (short)n});
                maxPitch = std::max(maxPitch, ((n * L.wCell + 6 + 3) + 15) & ~15);
            }
        }
        // the same level as one-cell-row tiles: calls with few frames (the single-frame drop-in call) want many short workgroups — with one
        // frame the 134 two-row tiles are one round of workgroups that live twice as long (host API: 192 -> 204 us per 752x480 frame)
        for (int i = 0; i < validRows; i++)
            for (int c0 = 0; c0 < L.nCols; c0 += per) {
                if (ORBX_MINB + c0 * L.wCell >= L.maxBX - 6) continue;
                tiles1.push_back(FastTile{(short)(l | (1 << 8)), (short)i, (short)c0, (short)std::min(per, L.nCols - c0)});
            }
        maxRows = std::max(maxRows, crows * L.hCell + 6 + crows);
        if (L.hCell > 64) { orbx_free(h); return orbx_fail(nullptr, ORB_E_INVALID, "FAST cell higher than 64 rows"); }
        if (L.wCell > 127) { orbx_free(h); return orbx_fail(nullptr, ORB_E_INVALID, "FAST cell wider than 127 columns"); }   // (k_octree's reciprocal tables)
    }
    h->pyrFrame = pyrOff; h->candFrame = candOff; h->selFrame = selOff; h->nodeCap = nodeCap; h->maxKp = maxKp;
    h->nTiles = (int)tiles.size(); h->nTiles1 = (int)tiles1.size();
    if (maxPitch > FAST_PITCH) { orbx_free(h); return orbx_fail(nullptr, ORB_E_INVALID, "FAST tile wider than FAST_PITCH"); }
    h->fastImgBytes = (maxRows * FAST_PITCH + 15) & ~15;
    h->fastSmem = (size_t)h->fastImgBytes + 4 * FAST_Q1W * 2 + FAST_Q2CAP * 2 + FAST_TW + (8 + FAST_MAXCELLS) * 4;
    // 100 B per node with the second child-count buffer, 84 without (+ 16 for the alignment of the 16-byte move records)
    h->octMerge = (size_t)(256 + 16) * 4 + (size_t)nodeCap * 100 + 16 + 15 <= 150 * 1024 ? 1 : 0;
    h->octKeyOff = (int)(((size_t)(256 + 16) * 4 + (size_t)nodeCap * (h->octMerge ? 100 : 84) + 16 + 15) & ~(size_t)15);
    h->octSmem = (size_t)h->octKeyOff + (size_t)OCT_KEYCAP * 6;   // with the key cache; launches without it pass octKeyOff bytes
    if (h->octSmem > 150 * 1024) h->octSmem = (size_t)h->octKeyOff;   // node arrays of a very large nFeatures leave no room: no cache
    if (h->fastSmem > 64 * 1024 || h->octSmem > 150 * 1024) { orbx_free(h); return orbx_fail(nullptr, ORB_E_INVALID, "configuration exceeds the LDS budget"); }

    if (hipSetDevice(device) != hipSuccess) { orbx_free(h); return orbx_fail(nullptr, ORB_E_HIP, "hipSetDevice failed"); }
    if (orb_lds_optin((const void*)k_octree<OCT_T_BATCH>, h->octSmem) != ORB_OK || orb_lds_optin((const void*)k_octree<OCT_T_SINGLE>, h->octSmem) != ORB_OK) {   // (on `device`: the opt-in is per device and kernel)
        orbx_free(h); return orbx_fail(nullptr, ORB_E_HIP, "hipFuncSetAttribute(k_octree) failed");
    }
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::string m = std::string(#call) + ": " + hipGetErrorString(e_); orbx_free(h); return orbx_fail(nullptr, e_ == hipErrorOutOfMemory ? ORB_E_NOMEM : ORB_E_HIP, m); } } while (0)
    CK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (auto& e : h->ev) CK(hipEventCreate(&e));
    const size_t B = (size_t)max_batch;
    CK(hipMalloc((void**)&h->d_pyr, std::max<size_t>(B * h->pyrFrame, 256)));
    {   // per-level cv::resize coefficient tables (xs | xw | ys | yw), identical arithmetic to the in-kernel resize_coef of k_resize
        std::vector<int> tab;
        for (int l = 1; l < nl; l++) {
            const int sw = h->lv[l - 1].w, sh = h->lv[l - 1].h, dw = h->lv[l].w, dh = h->lv[l].h;
            const double sx = 1. / ((double)dw / sw), sy = 1. / ((double)dh / sh);
            h->coefOff[l] = tab.size();
            tab.resize(tab.size() + 2 * (size_t)dw + 2 * (size_t)dh);
            int* xs = tab.data() + h->coefOff[l]; int* xw = xs + dw; int* ys = xw + dw; int* yw = ys + dh;
            for (int x = 0; x < dw; x++) { int s0, w0, w1; resize_coef(x, sx, sw, true, s0, w0, w1); xs[x] = s0; xw[x] = (w0 & 0xFFFF) | (w1 << 16); }
            for (int y = 0; y < dh; y++) { int s0, w0, w1; resize_coef(y, sy, sh, false, s0, w0, w1); ys[y] = s0; yw[y] = (w0 & 0xFFFF) | (w1 << 16); }
            // staging footprints of the 64 x 32 destination tiles (8-byte aligned pairs).  For the levels k_resize2 serves (scale <= 1.3) every
            // footprint is checked here, once, against the exact tables: it must cover both taps of every column / row of its tile and fit the
            // kernel's LDS tile — what the kernel otherwise takes on trust (a miss would read, an overflow would write, outside the staged tile)
            const int tX = (dw + RS_TW - 1) / RS_TW, tY = (dh + R2_TH - 1) / R2_TH;
            std::vector<int> fp;
            bool covered = true;
            for (int t = 0; t < tX; t++) {
                int lo, ext; resize2_footprint(t * RS_TW, RS_TW, dw, sw, sx, true, lo, ext); fp.push_back(lo); fp.push_back(ext);
                for (int x = t * RS_TW; x < std::min(t * RS_TW + RS_TW, dw); x++)
                    covered = covered && xs[x] >= lo && std::min(xs[x] + 1, sw - 1) < lo + 4 * ext;
                covered = covered && ext >= 1 && 4 * ext <= RS_PITCH;
            }
            for (int t = 0; t < tY; t++) {
                int lo, ext; resize2_footprint(t * R2_TH, R2_TH, dh, sh, sy, false, lo, ext); fp.push_back(lo); fp.push_back(ext);
                for (int y = t * R2_TH; y < std::min(t * R2_TH + R2_TH, dh); y++) {
                    const int y0 = std::min(std::max(ys[y], 0), sh - 1), y1 = std::min(std::max(ys[y] + 1, 0), sh - 1);
                    covered = covered && y0 >= lo && y1 < lo + ext;
                }
                covered = covered && ext >= 1 && ext <= R2_ROWS;
            }
            if (!covered && sx <= 1.3 && sy <= 1.3) {
                orbx_free(h);
                return orbx_fail(nullptr, ORB_E_INVALID, "k_resize2: a staging footprint does not cover its tile (level " + std::to_string(l) + ")");
            }
            if (tab.size() & 1) tab.push_back(0);
            h->tileTabOff[l] = tab.size();
            tab.insert(tab.end(), fp.begin(), fp.end());
        }
        CK(hipMalloc((void**)&h->d_coef, std::max<size_t>(tab.size() * 4, 256)));
        if (!tab.empty()) CK(hipMemcpy(h->d_coef, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        // k_pyramid_chain's regions (see the kernel): tiles of ~16 x 16 pixels of the top level; every level is partitioned among the tiles
        // (share t of a length n = [t n / T, (t + 1) n / T)), a tile's region of level l = bounding box of its share and of the footprint of its
        // region of level l + 1.  Not usable (chainTiles = 0: the launches fall back to the per-level kernels) if a region or the whole set
        // exceeds the LDS budget, or with a single level.
        if (nl >= 2) {
            const int top = nl - 1;
            const int TX = std::max(1, h->lv[top].w / 16), TY = std::max(1, h->lv[top].h / 16);
            std::vector<int4> reg((size_t)TX * TY * nl);
            int maxArea = 0, maxCoef = 0;
            for (int ty = 0; ty < TY; ty++)
                for (int tx = 0; tx < TX; tx++) {
                    int4* R = reg.data() + ((size_t)ty * TX + tx) * nl;
                    int cx0 = 0, cx1 = 0, cy0 = 0, cy1 = 0, ncoef = 0;    // region of the level above, [cx0, cx1) x [cy0, cy1)
                    for (int l = top; l >= 0; l--) {
                        const int w = h->lv[l].w, hh = h->lv[l].h;
                        int x0 = (int)((long long)tx * w / TX), x1 = (int)((long long)(tx + 1) * w / TX);
                        int y0 = (int)((long long)ty * hh / TY), y1 = (int)((long long)(ty + 1) * hh / TY);
                        if (l == 0) { x0 = w; x1 = 0; y0 = hh; y1 = 0; }              // the image itself is not produced: footprint only
                        if (l < top) {
                            const int* xs = tab.data() + h->coefOff[l + 1]; const int* ys = xs + 2 * h->lv[l + 1].w;
                            const int fx0 = xs[cx0], fx1 = std::min(xs[cx1 - 1] + 1, w - 1) + 1;
                            const int fy0 = std::min(std::max(ys[cy0], 0), hh - 1), fy1 = std::min(std::max(ys[cy1 - 1] + 1, 0), hh - 1) + 1;
                            x0 = std::min(x0, fx0); x1 = std::max(x1, fx1); y0 = std::min(y0, fy0); y1 = std::max(y1, fy1);
                        }
                        R[l] = make_int4(x0, y0, x1 - x0, y1 - y0);
                        maxArea = std::max(maxArea, (x1 - x0) * (y1 - y0));
                        if (l >= 1) ncoef += 2 * ((x1 - x0) + (y1 - y0));
                        cx0 = x0; cx1 = x1; cy0 = y0; cy1 = y1;
                    }
                    maxCoef = std::max(maxCoef, ncoef);
                }
            h->chainBuf = (maxArea + 15) & ~15;
            h->chainSmem = (size_t)2 * h->chainBuf + (size_t)maxCoef * 4;
            if (h->chainSmem <= 64 * 1024) {
                h->chainTiles = TX * TY;
                CK(hipMalloc((void**)&h->d_chain, reg.size() * sizeof(int4)));
                CK(hipMemcpy(h->d_chain, reg.data(), reg.size() * sizeof(int4), hipMemcpyHostToDevice));
            }
        }
    }
    CK(hipMalloc((void**)&h->d_cand, B * h->candFrame * 4));
    CK(hipMalloc((void**)&h->d_keyNode, B * h->candFrame * 2));
    CK(hipMalloc((void**)&h->d_candCount, B * nl * 4));
    CK(hipMemset(h->d_candCount, 0, B * nl * 4));
    CK(hipMalloc((void**)&h->d_order, B * 4));
    CK(hipMalloc((void**)&h->d_retry, ((size_t)B * tiles.size() * 2 + 2) * 4));   // k_fast's retry list: count, then two words per (frame, two-row tile) at most
    CK(hipMalloc((void**)&h->d_sel, B * h->selFrame * 4));
    CK(hipMalloc((void**)&h->d_selAux, B * h->selFrame * 4));
    // (+ ORBX_MAX_LEVELS entries: k_describe reads a frame's counts as one fixed-size block, whatever nlevels is)
    CK(hipMalloc((void**)&h->d_selCount, (B * nl + ORBX_MAX_LEVELS) * 4));
    CK(hipMalloc((void**)&h->d_lapCount, (B * nl + ORBX_MAX_LEVELS) * 4));
    CK(hipMemset(h->d_selCount, 0, (B * nl + ORBX_MAX_LEVELS) * 4));
    CK(hipMemset(h->d_lapCount, 0, (B * nl + ORBX_MAX_LEVELS) * 4));
    tiles.insert(tiles.end(), tiles1.begin(), tiles1.end());   // one allocation: [two-row list | one-row list]
    CK(hipMalloc((void**)&h->d_tiles, tiles.size() * sizeof(FastTile)));
    CK(hipMemcpy(h->d_tiles, tiles.data(), tiles.size() * sizeof(FastTile), hipMemcpyHostToDevice));
    h->imgStride = (width + 63) & ~63;
    CK(hipMalloc((void**)&h->d_img, (size_t)h->imgStride * height));
    h->kps1Off = 64;
    h->desc1Off = (h->kps1Off + (size_t)maxKp * sizeof(orb_keypoint) + 63) & ~(size_t)63;
    h->out1Bytes = h->desc1Off + (size_t)maxKp * 32;
    CK(hipMalloc((void**)&h->d_out1, h->out1Bytes));
    h->d_counts1 = (int32_t*)h->d_out1; h->d_kps1 = (orb_keypoint*)(h->d_out1 + h->kps1Off); h->d_desc1 = h->d_out1 + h->desc1Off;
    CK(hipHostMalloc((void**)&h->h_img, (size_t)h->imgStride * height, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&h->h_retry, 16, hipHostMallocDefault));
    h->h_retry[0] = 0; h->h_retry[1] = 1; h->h_retry[2] = 0; h->h_retry[3] = 0;
    CK(hipHostMalloc((void**)&h->h_out, h->out1Bytes, hipHostMallocDefault));
    {
        // c_umax / c_patternf / c_icmask do not depend on the configuration (HALF_PATCH_SIZE = 15, the 256 test pairs): uploaded by the FIRST handle of a
        // device, under a mutex — a later orbx_create (another camera's extractor, an adapter re-creating its handle for a new image size, any host
        // thread) must not rewrite constant memory that the kernels of a live handle are reading
        static std::mutex constMu;
        static bool constDone[64] = {false};
        std::lock_guard<std::mutex> lock(constMu);
        if (device < 0 || device >= 64 || !constDone[device]) {
            CK(hipMemcpyToSymbol(HIP_SYMBOL(c_umax), h->umax, sizeof(h->umax)));
            float pf[1024];
            for (int i = 0; i < 256; i++) {   // h_pattern: x0, y0, x1, y1 per pair (ORBextractor.cc:149-406)
                pf[4 * i] = (float)h_pattern[4 * i]; pf[4 * i + 1] = (float)h_pattern[4 * i + 2];
                pf[4 * i + 2] = (float)h_pattern[4 * i + 1]; pf[4 * i + 3] = (float)h_pattern[4 * i + 3];
            }
            CK(hipMemcpyToSymbol(HIP_SYMBOL(c_patternf), pf, sizeof(pf)));
            uint32_t icmask[16][12];
            for (int v = 0; v < 16; v++)
                for (int k = 0; k < 12; k++) {
                    uint32_t m = 0;
                    for (int t = 0; t < 4; t++) { const int u = 4 * k + t - 21; if (u >= -h->umax[v] && u <= h->umax[v]) m |= 0xFFu << (8 * t); }
                    icmask[v][k] = m;
                }
            CK(hipMemcpyToSymbol(HIP_SYMBOL(c_icmask), icmask, sizeof(icmask)));
            if (device >= 0 && device < 64) constDone[device] = true;
        }
    }
#undef CK
    *out = h;
    return ORB_OK;
}

extern "C" void orbx_destroy(orbx_handle h) { orbx_free(h); }
extern "C" const char* orbx_last_error(orbx_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }
extern "C" int orbx_max_keypoints(orbx_handle h) { return h ? h->maxKp : ORB_E_INVALID; }

extern "C" int orbx_get_tables(orbx_handle h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int32_t* fpl) {
    if (!h) return ORB_E_INVALID;
    for (int i = 0; i < h->cfg.nlevels; i++) {
        if (scale) scale[i] = h->scale[i];
        if (inv_scale) inv_scale[i] = h->invScale[i];
        if (sigma2) sigma2[i] = h->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = h->invSigma2[i];
        if (fpl) fpl[i] = h->nfeat[i];
    }
    return ORB_OK;
}

static void level_view(orbx_extractor* h, int l, const uint8_t*& base, size_t& frameStride, int& rowStride) {
    if (l == 0) { base = h->lastImages; frameStride = h->lastFrameStride; rowStride = h->lastRowStride; }
    else { base = h->d_pyr + h->lv[l].planeOff; frameStride = h->pyrFrame; rowStride = h->lv[l].stride; }
}

extern "C" int orbx_extract_batch_dev(orbx_handle h, const uint8_t* d_images, int batch, size_t frame_stride, int row_stride,
                                      int lap0, int lap1, orb_keypoint* d_kps, uint8_t* d_desc, int cap_per_frame,
                                      int32_t* d_counts, void* stream_) {
    if (!h) return ORB_E_INVALID;
    if (!d_images) return orbx_fail(h, ORB_E_EMPTY_IMAGE, "empty image");
    if (batch < 1 || batch > h->maxBatch) return orbx_fail(h, ORB_E_INVALID, "batch exceeds max_batch");
    if (row_stride < h->W || (row_stride & 3) || ((uintptr_t)d_images & 3) || (frame_stride & 3) || !d_kps || !d_desc || !d_counts || cap_per_frame < 1)
        return orbx_fail(h, ORB_E_INVALID, "bad batch arguments (row stride / base must be 4-byte aligned)");
    hipStream_t st = (hipStream_t)stream_;  // NULL = the HIP default stream (what torch uses unless told otherwise)
    HIPCHK(h, hipSetDevice(h->device));
    const int nl = h->cfg.nlevels;
    h->lastImages = d_images; h->lastFrameStride = frame_stride; h->lastRowStride = row_stride; h->lastBatch = batch;
    h->lastStream = st; h->lastSingle = h->hostCall; h->hostPyrValid = false;
    if (ORBX_EVENTS_ON(h)) HIPCHK(h, hipEventRecord(h->ev[0], st));
#ifndef ORBX_FRAME_ORDER
#define ORBX_FRAME_ORDER 1   // 0: frames in batch order (experiments)
#endif
#ifndef PYR_CHAIN_MAX_BATCH
#define PYR_CHAIN_MAX_BATCH 1   // calls of up to this many frames build the pyramid in ONE launch (k_pyramid_chain): the single-frame entry point
#endif
#ifndef ORBX_EXP_DUP
#define ORBX_EXP_DUP 0   // experiment only (tools/exp.py build): launch a stage twice — 1 pyramid, 2 octree, 4 describe — to read its MARGINAL cost in the
#endif                   // three-stream step (every one of them is idempotent) next to its standalone time
    // frames in descending order of the previous call's candidate counts (k_frame_order; it also clears the counters) — batches only
    const bool ordered = ORBX_FRAME_ORDER && batch >= 2 && batch <= ORDER_MAX_BATCH && !h->capturing;
    // the FAST plan of this call (the pyramid's first launch carries its bookkeeping, see below)
    const bool tall = batch >= FAST_TALL_MIN_BATCH;
    const int nTiles = tall ? h->nTiles : h->nTiles1;
    const int iniTh = std::min(std::max(h->cfg.ini_th_fast, 0), 255), minTh = std::min(std::max(h->cfg.min_th_fast, 0), 255);
    // Two passes for batches of FAST_TWO_PASS_MIN_BATCH frames or more (two dependent launches cost a small batch more than the first pass saves:
    // k_fast 0.036 vs 0.030 ms at 8 frames, 0.046 vs 0.042 at 16, 0.065 vs 0.063 at 32, 0.102 vs 0.105 at 64, 0.171 vs 0.184 at 128) — while they pay.  Both forms give the same key points; which is faster depends on the frames: the first pass saves the corners between the two
    // thresholds, every listed tile pays staging and pre-test a second time.  On the benchmark's frames 13 % of the tiles are listed (k_fast
    // 0.70 -> 0.63 ms per 512 frames), on sparsely textured ones (the same synthetic scene at 1280x720) most are (0.78 -> 0.90 ms).  The listed
    // share of the handle's previous two-pass call (written back by that call's k_octree, read here without waiting: a stale figure only delays the
    // switch) decides; in one-pass mode every FAST_PROBE_EVERY-th call takes the two passes to measure again.
    bool two = !FAST_XCD && FAST_TWO_PASS && tall && batch >= FAST_TWO_PASS_MIN_BATCH && iniTh > minTh && nTiles <= 65535 && batch <= 65535;
    if (two) {
        const volatile uint32_t* hr = h->h_retry;
        const bool pays = (double)hr[0] <= FAST_TWO_PASS_MAX_LISTED * (double)hr[1];
        two = pays || (h->fastCalls % FAST_PROBE_EVERY) == 0;
        h->fastCalls++;
    }
    h->fastLastTwoPass = two;
    // frame order + cleared counters + the empty second-pass list: one extra workgroup of the level-1 k_resize2 launch when there is one
    // (scale factors <= 1.3 and more than one level), else launches of their own
    const bool sep1 = nl > 1 && 1. / ((double)h->lv[1].w / h->lv[0].w) <= 1.3 && 1. / ((double)h->lv[1].h / h->lv[0].h) <= 1.3;
    const bool chain = h->chainTiles > 0 && batch <= PYR_CHAIN_MAX_BATCH;
#ifndef ORBX_FOLD_ORDER
#define ORBX_FOLD_ORDER 1   // 0: k_frame_order and the list's two memsets as launches of their own (experiments)
#endif
    const bool folded = ORBX_FOLD_ORDER && ordered && sep1 && !chain && R2_XCD && !(ORBX_EXP_DUP & 1);
    if (!folded) {
        if (ordered) hipLaunchKernelGGL(k_frame_order, dim3(1), dim3(ORDER_T), (size_t)batch * 4 + 16, st, h->d_candCount, nl, batch, h->d_order, h->h_retry + 2);
        else HIPCHK(h, hipMemsetAsync(h->d_candCount, 0, (size_t)batch * nl * 4, st));
    }
    // E1 pyramid
    if (chain) {
        PyrChainParams C;
        memset(&C, 0, sizeof(C));
        level_view(h, 0, C.img, C.imgFrame, C.imgStride);
        C.pyr = h->d_pyr; C.pyrFrame = h->pyrFrame; C.coef = h->d_coef; C.regions = h->d_chain; C.nlevels = nl; C.bufBytes = h->chainBuf;
        for (int l = 0; l < nl; l++) { C.planeOff[l] = h->lv[l].planeOff; C.stride[l] = h->lv[l].stride; C.w[l] = h->lv[l].w; C.h[l] = h->lv[l].h; C.coefOff[l] = (int)h->coefOff[l]; }
        hipLaunchKernelGGL(k_pyramid_chain, dim3(h->chainTiles, batch), dim3(PYC_T), h->chainSmem, st, C);
    }
    for (int rep = 0; !chain && rep < ((ORBX_EXP_DUP & 1) ? 2 : 1); rep++)
    for (int l = 1; l < nl; l++) {
        ResizeParams R;
        level_view(h, l - 1, R.src, R.sFrame, R.sStride);
        R.sw = h->lv[l - 1].w; R.sh = h->lv[l - 1].h;
        R.dst = h->d_pyr + h->lv[l].planeOff; R.dFrame = h->pyrFrame; R.dStride = h->lv[l].stride;
        R.dw = h->lv[l].w; R.dh = h->lv[l].h;
        R.scale_x = 1. / ((double)R.dw / R.sw); R.scale_y = 1. / ((double)R.dh / R.sh);
        R.coef = h->d_coef + h->coefOff[l];
        R.tileTab = h->d_coef + h->tileTabOff[l];
        if (R.scale_x <= 1.3 && R.scale_y <= 1.3) {   // separable tile kernel (its LDS footprint is sized for scale <= 1.3)
            R.tilesX = (R.dw + RS_TW - 1) / RS_TW; R.tilesY = (R.dh + R2_TH - 1) / R2_TH; R.batch = batch;
            R.unitsMagic = xcd_units_magic(R.tilesX * ((R.tilesY + R2_PAIR - 1) / R2_PAIR), batch);
            R.reverse = (l & 1) == 0;
            R.tilesXMagic = R.tilesX >= 2 ? (uint32_t)(0x100000000ull / (unsigned long long)R.tilesX + 1ull) : 0u;
#if R2_XCD
            const bool carry = folded && l == 1;           // + the bookkeeping workgroup (LDS: batch + 1 words <= R2_SMEM for ORDER_MAX_BATCH frames)
            static_assert((size_t)ORDER_MAX_BATCH * 4 + 16 <= R2_SMEM, "frame_order_body's keys fit k_resize2's LDS block");
            R.ordCand = carry ? h->d_candCount : nullptr; R.ordOut = h->d_order; R.ordHostMax = h->h_retry + 2; R.ordLevels = nl;
            R.ordRetry = carry && two ? h->d_retry : nullptr; R.ordTiles = (uint32_t)nTiles * (uint32_t)batch;
            const dim3 g2(R.tilesX * ((R.tilesY + R2_PAIR - 1) / R2_PAIR) * 8 * ((batch + 7) / 8) + (carry ? 8 : 0));
            if (carry) hipLaunchKernelGGL(k_resize2<true>, g2, dim3(256), R2_SMEM, st, R);
            else hipLaunchKernelGGL(k_resize2<false>, g2, dim3(256), R2_SMEM, st, R);
#else
            dim3 grid(R.tilesX, (R.tilesY + R2_PAIR - 1) / R2_PAIR, batch);
            hipLaunchKernelGGL(k_resize2<false>, grid, dim3(256), R2_SMEM, st, R);
#endif
        } else {
            dim3 grid((R.dw + RS_TW - 1) / RS_TW, (R.dh + RS_TH - 1) / RS_TH, batch);
            hipLaunchKernelGGL(k_resize, grid, dim3(256), RS_PITCH * RS_ROWS + (2 * RS_TW + 2 * RS_TH) * 4, st, R);
        }
    }
    if (ORBX_EVENTS_ON(h)) HIPCHK(h, hipEventRecord(h->ev[1], st));
    if (h->hostCall && h->hostPyr && h->d_bpyr) {
        // mvImagePyramid for the host: fork behind the pyramid, border frame + ONE copy of the slab on the second stream; orbx_extract joins
        BorderParams Bp;
        memset(&Bp, 0, sizeof(Bp));
        for (int l = 0; l < nl; l++) {
            size_t fs;
            level_view(h, l, Bp.src[l], fs, Bp.sStride[l]);
            Bp.w[l] = h->lv[l].w; Bp.h[l] = h->lv[l].h; Bp.dOff[l] = h->bOff[l]; Bp.dPitch[l] = h->bPitch[l]; Bp.rowStart[l] = h->bRowStart[l];
        }
        Bp.rowStart[nl] = h->bRowStart[nl]; Bp.dst = h->d_bpyr; Bp.nlevels = nl;
        HIPCHK(h, hipEventRecord(h->evFork, st));
        HIPCHK(h, hipStreamWaitEvent(h->stream2, h->evFork, 0));
        hipLaunchKernelGGL(k_border_pyramid, dim3(h->bRowStart[nl]), dim3(256), 0, h->stream2, Bp);
        HIPCHK(h, hipMemcpyAsync(h->h_bpyr, h->d_bpyr, h->bpyrBytes, hipMemcpyDeviceToHost, h->stream2));
        HIPCHK(h, hipEventRecord(h->evJoin, h->stream2));
        h->hostPyrValid = true;
    }
    // E2 FAST
    {
        FastParams F;
        memset(&F, 0, sizeof(F));
        for (int l = 0; l < nl; l++) {
            FastLevel& fl = F.lv[l]; const LevelHost& L = h->lv[l];
            level_view(h, l, fl.base, fl.frameStride, fl.rowStride);
            fl.w = L.w; fl.h = L.h; fl.nCols = L.nCols; fl.nRows = L.nRows; fl.wCell = L.wCell; fl.hCell = L.hCell;
            fl.wCellMagic = (65536 + L.wCell - 1) / L.wCell;   // wCell = ceil(fw / floor(fw / 30)) < 60
            fl.candOff = L.candOff; fl.candCap = L.candCap;
        }
        F.tiles = tall ? h->d_tiles : h->d_tiles + h->nTiles; F.cand = h->d_cand; F.candFrame = h->candFrame; F.candCount = h->d_candCount; F.nlevels = nl;
        F.iniTh = iniTh; F.minTh = minTh;
        F.imgBytes = h->fastImgBytes;
        F.nTiles = nTiles; F.batch = batch; F.retry = h->d_retry; F.order = ordered ? h->d_order : nullptr;
#if FAST_XCD
        hipLaunchKernelGGL((k_fast<2>), dim3(nTiles * 8 * ((batch + 7) / 8)), dim3(256), h->fastSmem, st, F);
#else
        if (two) {
            // [0] the list's count, [1] the tiles of THIS call: both words travel back in one copy, so the pair the next call's policy (and
            // orbx_last_fast_passes) reads always belongs to one call, whatever batch sizes alternate on the handle
            if (!folded) {   // (else the pyramid's bookkeeping workgroup has written both)
                HIPCHK(h, hipMemsetAsync(h->d_retry, 0, 4, st));
                HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)(h->d_retry + 1), (int)((uint32_t)nTiles * (uint32_t)batch), 1, st));
            }
            hipLaunchKernelGGL((k_fast<0>), dim3(nTiles, batch), dim3(256), h->fastSmem, st, F);
            {
                // the second pass's grid: the previous two-pass call's list length (+ 25 % + 1 024) if that call had the same tile count, else every tile
                const volatile uint32_t* hr2 = h->h_retry;
                const uint32_t all = (uint32_t)nTiles * (uint32_t)batch;
                uint32_t g1 = all;
                if (hr2[1] == all) g1 = std::min(all, std::max(2048u, hr2[0] + hr2[0] / 4u + 1024u));
#ifdef FAST_P1_GRID   // tests: a fixed, tiny second-pass grid — every workgroup then walks several list entries
                g1 = std::min(all, (uint32_t)(FAST_P1_GRID));
#endif
                hipLaunchKernelGGL((k_fast<1>), dim3(g1), dim3(256), h->fastSmem, st, F);
            }
            // (the pair travels back to h_retry in k_octree's first workgroup — no copy launch)
        } else {
            hipLaunchKernelGGL((k_fast<2>), dim3(nTiles, batch), dim3(256), h->fastSmem, st, F);
        }
#endif
    }
    if (ORBX_EVENTS_ON(h)) HIPCHK(h, hipEventRecord(h->ev[2], st));
    // E3 octree
    {
        OctParams O;
        memset(&O, 0, sizeof(O));
        for (int l = 0; l < nl; l++) {
            OctLevel& ol = O.lv[l]; const LevelHost& L = h->lv[l];
            ol.W = L.maxBX - ORBX_MINB; ol.H = L.maxBY - ORBX_MINB; ol.N = h->nfeat[l]; ol.nIni = L.nIni; ol.hX = L.hX;
            ol.nCols = L.nCols; ol.wCell = L.wCell; ol.hCell = L.hCell; ol.candOff = L.candOff; ol.candCap = L.candCap;
            ol.wCellM19 = ((1 << 19) + L.wCell - 1) / L.wCell; ol.hCellM19 = ((1 << 19) + L.hCell - 1) / L.hCell;
            ol.selOff = L.selOff; ol.selCap = L.selCap; ol.scale = h->scale[l];
        }
        O.cand = h->d_cand; O.candFrame = h->candFrame; O.candCount = h->d_candCount; O.nlevels = nl; O.keyNode = h->d_keyNode;
        O.sel = h->d_sel; O.selAux = h->d_selAux; O.selFrame = h->selFrame; O.selCount = h->d_selCount; O.lapCount = h->d_lapCount;
        const bool cache = batch <= OCT_CACHE_MAX_BATCH && h->octSmem > (size_t)h->octKeyOff;
        O.nodeCap = h->nodeCap; O.merge = h->octMerge; O.lap0 = lap0; O.lap1 = lap1; O.keyCap = cache ? OCT_KEYCAP : 0; O.keyOff = h->octKeyOff;
        O.order = ordered ? h->d_order : nullptr;
        O.gridLevels = nl; O.heavyMode = 0; O.heavyMin = OCT_HEAVY_MIN;
        O.retry = h->d_retry; O.retryHost = two && !h->capturing ? h->h_retry : nullptr;
        // A problem of tens of thousands of candidates (every pixel a corner) runs ~3x longer than the whole launch of ordinary ones on 256 threads
        // and no ordering shortens ONE workgroup: when the call before last had such a problem (k_frame_order's word; a stale word only picks the
        // other of two correct plans) they get a launch of their own with OCT_T_HEAVY threads, over the levels that can hold one.
        int heavyLevels = 0;
        while (heavyLevels < nl && h->lv[heavyLevels].candCap >= OCT_HEAVY_MIN) heavyLevels++;
        const bool heavy = OCT_HEAVY_PASS && ordered && heavyLevels > 0 && ((const volatile uint32_t*)h->h_retry)[2] >= (uint32_t)OCT_HEAVY_MIN;
        h->lastOrdered = ordered ? 1 : 0; h->lastHeavy = heavy ? 1 : 0;
        if (heavy) {
            OctParams Oh = O;
            Oh.heavyMode = 1; Oh.gridLevels = heavyLevels;
            hipLaunchKernelGGL(k_octree<OCT_T_HEAVY>, dim3(heavyLevels * batch), dim3(OCT_T_HEAVY), (size_t)h->octKeyOff, st, Oh);   // (first: the long ones)
            O.heavyMode = 2;
        }
        for (int rep = 0; rep < ((ORBX_EXP_DUP & 2) ? 2 : 1); rep++)
            if (batch == 1) hipLaunchKernelGGL(k_octree<OCT_T_SINGLE>, dim3(nl * batch), dim3(OCT_T_SINGLE), cache ? h->octSmem : (size_t)h->octKeyOff, st, O);
            else hipLaunchKernelGGL(k_octree<OCT_T_BATCH>, dim3(nl * batch), dim3(OCT_T_BATCH), cache ? h->octSmem : (size_t)h->octKeyOff, st, O);
    }
    if (ORBX_EVENTS_ON(h)) HIPCHK(h, hipEventRecord(h->ev[3], st));
    // E5-E8 orientation + blur + descriptors + assembly
    {
        DescParams D;
        memset(&D, 0, sizeof(D));
        for (int l = 0; l < nl; l++) {
            DescLevel& dl = D.lv[l]; const LevelHost& L = h->lv[l];
            level_view(h, l, dl.base, dl.frameStride, dl.rowStride);
            dl.w = L.w; dl.h = L.h; dl.selOff = L.selOff; dl.selCap = L.selCap; dl.scale = h->scale[l]; dl.size = (float)(int)(31 * h->scale[l]);
        }
        D.sel = h->d_sel; D.selAux = h->d_selAux; D.selFrame = h->selFrame; D.selCount = h->d_selCount; D.lapCount = h->d_lapCount;
        D.nlevels = nl; D.kps = d_kps; D.desc = d_desc; D.cap = cap_per_frame; D.counts = d_counts;
        D.unitStart[0] = 0;
        for (int l = 0; l < nl; l++) D.unitStart[l + 1] = D.unitStart[l] + (h->lv[l].selCap + 3) / 4;
        for (int l = nl + 1; l <= ORBX_MAX_LEVELS; l++) D.unitStart[l] = INT_MAX;
        D.groups = D.unitStart[nl]; D.batch = batch; D.groupsMagic = xcd_units_magic(D.groups * 2, batch);
#if DESC_KPW == 2
        for (int rep = 0; rep < ((ORBX_EXP_DUP & 4) ? 2 : 1); rep++)
            hipLaunchKernelGGL(k_describe2, dim3(D.groups * 2 * 8 * ((batch + 7) / 8)), dim3(64), DESC2_WAVE_BYTES + 48, st, D);
#else
        hipLaunchKernelGGL(k_describe, dim3(D.groups * (4 / DESC_WPB) * 8 * ((batch + 7) / 8)), dim3(64 * DESC_WPB), DESC_WPB * DESC_WAVE_STRIDE + 96, st, D);
#endif
    }
    if (ORBX_EVENTS_ON(h)) HIPCHK(h, hipEventRecord(h->ev[4], st));
    if (ORBX_EVENTS_ON(h)) h->timed = true;
    if (h->hostPyrValid) HIPCHK(h, hipStreamWaitEvent(st, h->evJoin, 0));   // the host-pyramid branch joins (inside a capture: the graph's second leaf)
    HIPCHK(h, hipGetLastError());
    return ORB_OK;
}

extern "C" int orbx_set_host_pyramid(orbx_handle h, int keep) {
    if (!h) return ORB_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    if (keep && !h->d_bpyr) {
        const int nl = h->cfg.nlevels;
        size_t off = 0; int rows = 0;
        for (int l = 0; l < nl; l++) {
            h->bPitch[l] = (h->lv[l].w + 2 * ORBX_EDGE + 63) & ~63;
            h->bOff[l] = off; h->bRowStart[l] = rows;
            off += (size_t)h->bPitch[l] * (h->lv[l].h + 2 * ORBX_EDGE); rows += h->lv[l].h + 2 * ORBX_EDGE;
            if (h->lv[l].w <= ORBX_EDGE || h->lv[l].h <= ORBX_EDGE) return orbx_fail(h, ORB_E_INVALID, "pyramid level smaller than its border");
        }
        h->bRowStart[nl] = rows; h->bpyrBytes = off;
        HIPCHK(h, hipMalloc((void**)&h->d_bpyr, off));
        HIPCHK(h, hipHostMalloc((void**)&h->h_bpyr, off, hipHostMallocDefault));
        if (!h->stream2) HIPCHK(h, hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
        if (!h->evFork) HIPCHK(h, hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming));
        if (!h->evJoin) HIPCHK(h, hipEventCreateWithFlags(&h->evJoin, hipEventDisableTiming));
    }
    h->hostPyr = keep ? 1 : 0;
    return ORB_OK;
}

extern "C" int orbx_host_pyramid_level(orbx_handle h, int level, const uint8_t** ptr, int* w, int* hgt, int* stride) {
    if (!h || level < 0 || level >= h->cfg.nlevels) return ORB_E_INVALID;
    if (!h->hostPyrValid || !h->lastSingle) return orbx_fail(h, ORB_E_INVALID, "no host pyramid: orbx_set_host_pyramid(h, 1) before orbx_extract");
    if (ptr) *ptr = h->h_bpyr + h->bOff[level] + (size_t)ORBX_EDGE * h->bPitch[level] + ORBX_EDGE;
    if (w) *w = h->lv[level].w;
    if (hgt) *hgt = h->lv[level].h;
    if (stride) *stride = h->bPitch[level];
    return ORB_OK;
}

extern "C" int orbx_extract_view(orbx_handle h, const uint8_t* image, int width, int height, int stride, int lap0, int lap1,
                                 const orb_keypoint** kps, const uint8_t** desc, int* n_out, int* mono_index) {
    if (!h) return ORB_E_INVALID;
    if (n_out) *n_out = 0;
    if (mono_index) *mono_index = 0;
    if (kps) *kps = nullptr;
    if (desc) *desc = nullptr;
    if (!image || width <= 0 || height <= 0) return orbx_fail(h, ORB_E_EMPTY_IMAGE, "empty image");
    if (width != h->W || height != h->H || stride < width) return orbx_fail(h, ORB_E_INVALID, "image size differs from the handle's");
    HIPCHK(h, hipSetDevice(h->device));
    // the image into the pinned staging buffer (device pitch), then H2D + kernels + D2H as one graph launch and ONE synchronisation
    if (stride == h->imgStride) memcpy(h->h_img, image, (size_t)stride * (height - 1) + width);
    else for (int y = 0; y < height; y++) memcpy(h->h_img + (size_t)y * h->imgStride, image + (size_t)y * stride, (size_t)width);
    const size_t imgBytes = (size_t)h->imgStride * height;
    int rc = ORB_OK;
    h->hostCall = true;
    if (h->graphState >= 0 && (!h->graphExec || h->graphLap0 != lap0 || h->graphLap1 != lap1 || h->graphHostPyr != h->hostPyr)) {
        if (h->graphExec) { (void)hipGraphExecDestroy(h->graphExec); h->graphExec = nullptr; }
        if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            h->capturing = true;
            hipError_t em = hipMemcpyAsync(h->d_img, h->h_img, imgBytes, hipMemcpyHostToDevice, h->stream);
            rc = orbx_extract_batch_dev(h, h->d_img, 1, imgBytes, h->imgStride, lap0, lap1, h->d_kps1, h->d_desc1, h->maxKp, h->d_counts1, h->stream);
            if (em == hipSuccess) em = hipMemcpyAsync(h->h_out, h->d_out1, h->out1Bytes, hipMemcpyDeviceToHost, h->stream);
            h->capturing = false;
            hipGraph_t g = nullptr;
            const hipError_t ec = hipStreamEndCapture(h->stream, &g);
            if (rc == ORB_OK && em == hipSuccess && ec == hipSuccess && g && hipGraphInstantiate(&h->graphExec, g, nullptr, nullptr, 0) == hipSuccess) {
                h->graphLap0 = lap0; h->graphLap1 = lap1; h->graphHostPyr = h->hostPyr; h->graphState = 1;
            } else { h->graphExec = nullptr; h->graphState = -1; }
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            rc = ORB_OK;
        } else { h->graphState = -1; (void)hipGetLastError(); }
    }
    if (h->graphExec) {
        h->timed = false;   // per-kernel events are not part of the graph
        const hipError_t e = hipGraphLaunch(h->graphExec, h->stream);
        h->hostPyrValid = h->hostPyr != 0; h->lastStream = h->stream; h->lastSingle = true;   // (what the captured call set, for the replay)
        if (e != hipSuccess) { h->hostCall = false; HIPCHK(h, e); }
    } else {
        hipError_t e = hipMemcpyAsync(h->d_img, h->h_img, imgBytes, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            rc = orbx_extract_batch_dev(h, h->d_img, 1, imgBytes, h->imgStride, lap0, lap1, h->d_kps1, h->d_desc1, h->maxKp, h->d_counts1, h->stream);
            if (rc != ORB_OK) { h->hostCall = false; return rc; }
            e = hipMemcpyAsync(h->h_out, h->d_out1, h->out1Bytes, hipMemcpyDeviceToHost, h->stream);
        }
        if (e != hipSuccess) { h->hostCall = false; HIPCHK(h, e); }
    }
    h->hostCall = false;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int32_t counts[2];
    memcpy(counts, h->h_out, sizeof(counts));
    if (n_out) *n_out = counts[0];
    if (mono_index) *mono_index = counts[1];
    if (kps) *kps = (const orb_keypoint*)(h->h_out + h->kps1Off);
    if (desc) *desc = h->h_out + h->desc1Off;
    return ORB_OK;
}

extern "C" int orbx_extract(orbx_handle h, const uint8_t* image, int width, int height, int stride, int lap0, int lap1,
                            orb_keypoint* kps, uint8_t* desc, int cap, int* n_out, int* mono_index) {
    const orb_keypoint* k = nullptr; const uint8_t* d = nullptr;
    int n = 0;
    const int rc = orbx_extract_view(h, image, width, height, stride, lap0, lap1, &k, &d, &n, mono_index);
    if (n_out) *n_out = n;
    if (rc != ORB_OK) return rc;
    if (n > cap) return orbx_fail(h, ORB_E_CAPACITY, "output capacity too small");
    if (n > 0) {
        if (!kps || !desc) return orbx_fail(h, ORB_E_INVALID, "null output");
        memcpy(kps, k, (size_t)n * sizeof(orb_keypoint));
        memcpy(desc, d, (size_t)n * 32);
    }
    return ORB_OK;
}

extern "C" int orbx_stereo_matches(orbx_handle left, orbx_handle right, const orb_keypoint* d_kps_l, const uint8_t* d_desc_l,
                                   const int32_t* d_counts_l, const orb_keypoint* d_kps_r, const uint8_t* d_desc_r, const int32_t* d_counts_r,
                                   int cap_per_frame, int batch, float mb, float mbf, float* d_u_right, float* d_depth, int32_t* d_work,
                                   void* stream) {
    if (!left || !right) return ORB_E_INVALID;
    orbx_extractor* h = left;
    if (!d_kps_l || !d_desc_l || !d_counts_l || !d_kps_r || !d_desc_r || !d_counts_r || !d_u_right || !d_depth || !d_work || cap_per_frame < 1 ||
        cap_per_frame > 65535 || batch < 1 || batch > left->lastBatch || batch > right->lastBatch || left->W != right->W || left->H != right->H ||
        left->cfg.nlevels != right->cfg.nlevels || !left->lastImages || !right->lastImages || !(mb > 0))
        return orbx_fail(h, ORB_E_INVALID, "bad stereo arguments (both handles must have extracted this batch)");
    HIPCHK(h, hipSetDevice(h->device));
    StereoParams S;
    memset(&S, 0, sizeof(S));
    for (int l = 0; l < h->cfg.nlevels; l++) {
        for (int side = 0; side < 2; side++) {
            orbx_extractor* e = side ? right : left;
            DescLevel& dl = side ? S.lvR[l] : S.lvL[l];
            level_view(e, l, dl.base, dl.frameStride, dl.rowStride);
            dl.w = e->lv[l].w; dl.h = e->lv[l].h; dl.scale = e->scale[l];
        }
        S.invScale[l] = h->invScale[l];
    }
    S.kpsL = d_kps_l; S.kpsR = d_kps_r; S.descL = d_desc_l; S.descR = d_desc_r; S.cntL = d_counts_l; S.cntR = d_counts_r;
    S.cap = cap_per_frame; S.mb = mb; S.mbf = mbf; S.uRight = d_u_right; S.depth = d_depth; S.sad = d_work; S.nRows = h->H;
    // a keypoint is listed in at most 2 * ceil(2 * scale[top level]) + 3 rows: the slab can never overflow
    const int rowCap = cap_per_frame * (2 * (int)std::ceil(2.0f * h->scale[h->cfg.nlevels - 1]) + 3);
    if (!h->d_rowStart || h->rowCapAlloc < rowCap) {   // lazily sized for max_batch frames
        if (h->d_rowStart) { (void)hipFree(h->d_rowStart); (void)hipFree(h->d_rowIdx); h->d_rowStart = nullptr; h->d_rowIdx = nullptr; }
        HIPCHK(h, hipMalloc((void**)&h->d_rowStart, (size_t)h->maxBatch * (h->H + 1) * 4));
        HIPCHK(h, hipMalloc((void**)&h->d_rowIdx, (size_t)h->maxBatch * rowCap * 4));
        h->rowCapAlloc = rowCap;
    }
    S.rowStart = h->d_rowStart; S.rowIdx = h->d_rowIdx; S.rowCap = rowCap;
    hipStream_t st = (hipStream_t)stream;
    const int rowsLds = std::max(0, std::min(rowCap, (STEREO_ROWS_LDS_MAX - (2 * h->H + 256) * 4) / 4));   // row lists staged in LDS if a frame's fit (they do: ~5 per key point)
    hipLaunchKernelGGL(k_stereo_rows, dim3(batch), dim3(256), (size_t)(2 * h->H + 256 + rowsLds) * 4, st, S, rowsLds);
    hipLaunchKernelGGL(k_stereo_match, dim3((cap_per_frame + 3) / 4, batch), dim3(256), 4 * 11 * 24, st, S);
    hipLaunchKernelGGL(k_stereo_cull, dim3(batch), dim3(256), (size_t)(STEREO_CULL_BINS + 32 + 256 + 4) * 4, st, S);
    HIPCHK(h, hipGetLastError());
    return ORB_OK;
}

extern "C" int orbx_pyramid_level(orbx_handle h, int frame, int level, const uint8_t** d_ptr, int* w, int* hgt, int* stride) {
    if (!h || level < 0 || level >= h->cfg.nlevels || frame < 0 || frame >= h->lastBatch || !h->lastImages) return ORB_E_INVALID;
    const uint8_t* base; size_t fs; int rs;
    level_view(h, level, base, fs, rs);
    if (d_ptr) *d_ptr = base + (size_t)frame * fs;
    if (w) *w = h->lv[level].w;
    if (hgt) *hgt = h->lv[level].h;
    if (stride) *stride = rs;
    return ORB_OK;
}

extern "C" int orbx_copy_level(orbx_handle h, int frame, int level, int border, uint8_t* out) {
    const uint8_t* p; int w, hh, rs;
    int rc = orbx_pyramid_level(h, frame, level, &p, &w, &hh, &rs);
    if (rc != ORB_OK || !out || border < 0 || border >= w || border >= hh) return ORB_E_INVALID;
    const int ow = w + 2 * border, oh = hh + 2 * border;
    if (h->hostPyrValid && h->lastSingle && frame == 0 && border <= ORBX_EDGE) {   // the bordered slab of the last orbx_extract is on the host already
        const uint8_t* src = h->h_bpyr + h->bOff[level] + (size_t)(ORBX_EDGE - border) * h->bPitch[level] + (ORBX_EDGE - border);
        for (int y = 0; y < oh; y++) memcpy(out + (size_t)y * ow, src + (size_t)y * h->bPitch[level], (size_t)ow);
        return ORB_OK;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->lastStream));   // the stream of the call that built the pyramid — not the device: other handles keep running
    uint8_t* in = out + (size_t)border * ow + border;
    HIPCHK(h, hipMemcpy2DAsync(in, ow, p, rs, w, hh, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (border == 0) return ORB_OK;
    // copyMakeBorder(BORDER_REFLECT_101), ORBextractor.cc:1173-1179: the side columns of every image row, then whole rows above and below
    auto refl = [](int p_, int len) { if (p_ < 0) p_ = -p_; if (p_ >= len) p_ = 2 * (len - 1) - p_; return p_; };
    for (int y = 0; y < hh; y++) {
        uint8_t* row = in + (size_t)y * ow;
        for (int x = 1; x <= border; x++) { row[-x] = row[refl(-x, w)]; row[w - 1 + x] = row[refl(w - 1 + x, w)]; }
    }
    for (int y = 0; y < oh; y++)
        if (y < border || y >= border + hh) memcpy(out + (size_t)y * ow, out + (size_t)(border + refl(y - border, hh)) * ow, (size_t)ow);
    return ORB_OK;
}

extern "C" int orbx_stereo_matches_last(orbx_handle left, orbx_handle right, float mb, float mbf, float* u_right, float* depth, int cap, int* n_left) {
    if (!left || !right) return ORB_E_INVALID;
    orbx_extractor* h = left;
    if (n_left) *n_left = 0;
    if (!left->lastSingle || !right->lastSingle || left->maxKp != right->maxKp || left->device != right->device)
        return orbx_fail(h, ORB_E_INVALID, "orbx_stereo_matches_last: both handles (same configuration) must have run orbx_extract last");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t K = (size_t)h->maxKp;
    if (!h->d_stereo1) {
        HIPCHK(h, hipMalloc((void**)&h->d_stereo1, 3 * K * 4));
        HIPCHK(h, hipHostMalloc((void**)&h->h_stereo1, 2 * K * 4, hipHostMallocDefault));
    }
    int32_t cnt[2];
    memcpy(cnt, h->h_out, sizeof(cnt));   // the pinned output block still holds the last call's counts
    const int n = cnt[0];
    if (n_left) *n_left = n;
    if (n > cap) return orbx_fail(h, ORB_E_CAPACITY, "output capacity too small");
    if (n <= 0) return ORB_OK;
    if (!u_right || !depth) return orbx_fail(h, ORB_E_INVALID, "null output");
    // (both orbx_extract calls synchronised their streams before returning: the right handle's slabs are complete)
    const int rc = orbx_stereo_matches(left, right, left->d_kps1, left->d_desc1, left->d_counts1, right->d_kps1, right->d_desc1, right->d_counts1,
                                       h->maxKp, 1, mb, mbf, h->d_stereo1, h->d_stereo1 + K, (int32_t*)(h->d_stereo1 + 2 * K), h->stream);
    if (rc != ORB_OK) return rc;
    HIPCHK(h, hipMemcpyAsync(h->h_stereo1, h->d_stereo1, 2 * K * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memcpy(u_right, h->h_stereo1, (size_t)n * 4);
    memcpy(depth, h->h_stereo1 + K, (size_t)n * 4);
    return ORB_OK;
}

extern "C" int orbx_debug_candidates(orbx_handle h, int frame, int level, int32_t* xys, int cap, int* n_out) {
    if (!h || level < 0 || level >= h->cfg.nlevels || frame < 0 || frame >= h->lastBatch || !n_out) return ORB_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->lastStream));
    int n = 0;
    HIPCHK(h, hipMemcpy(&n, h->d_candCount + (size_t)frame * h->cfg.nlevels + level, 4, hipMemcpyDeviceToHost));
    *n_out = n;
    const int m = std::min(std::min(n, cap), h->lv[level].candCap);
    if (m > 0 && xys) {
        std::vector<uint32_t> tmp(m);
        HIPCHK(h, hipMemcpy(tmp.data(), h->d_cand + (size_t)frame * h->candFrame + h->lv[level].candOff, (size_t)m * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < m; i++) { xys[3 * i] = tmp[i] & 0xFFF; xys[3 * i + 1] = (tmp[i] >> 12) & 0xFFF; xys[3 * i + 2] = tmp[i] >> 24; }
    }
    return ORB_OK;
}

extern "C" int orbx_debug_selected(orbx_handle h, int frame, int level, int32_t* xys, int cap, int* n_out) {
    if (!h || level < 0 || level >= h->cfg.nlevels || frame < 0 || frame >= h->lastBatch || !n_out) return ORB_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->lastStream));
    int n = 0;
    HIPCHK(h, hipMemcpy(&n, h->d_selCount + (size_t)frame * h->cfg.nlevels + level, 4, hipMemcpyDeviceToHost));
    *n_out = n;
    const int m = std::min(n, cap);
    if (m > 0 && xys) {
        std::vector<uint32_t> tmp(m);
        HIPCHK(h, hipMemcpy(tmp.data(), h->d_sel + (size_t)frame * h->selFrame + h->lv[level].selOff, (size_t)m * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < m; i++) { xys[3 * i] = tmp[i] & 0xFFF; xys[3 * i + 1] = (tmp[i] >> 12) & 0xFFF; xys[3 * i + 2] = tmp[i] >> 24; }
    }
    return ORB_OK;
}

#ifdef ORBX_PROF
extern "C" int orbx_debug_prof(unsigned long long* out32, int clear) {
    static unsigned long long z[2 * 8 * 256], r[2 * 8 * 256];
    if (hipDeviceSynchronize() != hipSuccess) return ORB_E_HIP;
    if (hipMemcpyFromSymbol(r, HIP_SYMBOL(g_prof), sizeof(r)) != hipSuccess) return ORB_E_HIP;
    for (int k = 0; k < 16; k++) { out32[k] = 0; for (int i = 0; i < 256; i++) out32[k] += r[k * 256 + i]; }
    if (clear && hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return ORB_E_HIP;
    return ORB_OK;
}
#endif

extern "C" int orbx_last_fast_passes(orbx_handle h, int* two_pass, uint32_t* listed, uint32_t* tiles) {
    if (!h) return ORB_E_INVALID;
    const volatile uint32_t* hr = h->h_retry;
    if (two_pass) *two_pass = h->fastLastTwoPass;
    if (listed) *listed = hr[0];
    if (tiles) *tiles = hr[1];
    return ORB_OK;
}

extern "C" int orbx_last_schedule(orbx_handle h, int* frames_ordered, int* heavy_octree_pass) {
    if (!h) return ORB_E_INVALID;
    if (frames_ordered) *frames_ordered = h->lastOrdered;
    if (heavy_octree_pass) *heavy_octree_pass = h->lastHeavy;
    return ORB_OK;
}

extern "C" int orbx_enable_timing(orbx_handle h, int on) {
    if (!h) return ORB_E_INVALID;
    h->timing = on != 0;
    if (!on) h->timed = false;
    return ORB_OK;
}
extern "C" int orbx_last_timing(orbx_handle h, float* ms5) {
    if (!h || !ms5 || !h->timed) return ORB_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventSynchronize(h->ev[4]));
    for (int i = 0; i < 4; i++) HIPCHK(h, hipEventElapsedTime(&ms5[i], h->ev[i], h->ev[i + 1]));
    HIPCHK(h, hipEventElapsedTime(&ms5[4], h->ev[0], h->ev[4]));
    return ORB_OK;
}
