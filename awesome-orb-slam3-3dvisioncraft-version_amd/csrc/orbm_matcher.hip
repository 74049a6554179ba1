// orbm_matcher.hip — stage 2 of the hot path on gfx950: ORBmatcher (reference src/ORBmatcher.cc) and the Frame grid
// helpers it depends on (src/Frame.cc:444-478, 755-862), over the flattened records of include/orbhip.h.
//
//   k_hamming        M1   ORBmatcher::DescriptorDistance for a Q x T tile            ORBmatcher.cc:2700-2716
//   k_knn2           M10  BFMatcher(NORM_HAMMING).knnMatch(k=2)                       Frame.cc:1300
//   k_grid_build     M2   Frame::AssignFeaturesToGrid / PosInGrid -> CSR              Frame.cc:444-478, 852-862
//   k_sbp_candidates2 M3  Frame::GetFeaturesInArea + Hamming, a 32-lane half-wave per query (k_sbp_candidates: one wave per query,
//                         the -DSBP_HALF=0 build and the body of the wide-window fallback)     Frame.cc:755-850
//   k_sbp_resolve    M4/M5 serial-order resolution of SearchByProjection (one wave per frame), rotation histogram
//                                                                                      ORBmatcher.cc:59-255, 2244-2509
//   k_bow            M6   SearchByBoW(KF,F): one workgroup per pair, one wave per shared vocabulary node  :323-587
//
// Results are identical to the reference's serial loops: candidate enumeration order, strict-'<' best/second updates,
// "skip keypoints claimed earlier in this call" and the rotation-histogram quirks are all reproduced.
// 256-bit Hamming = 8 x v_bcnt_u32 on two 16-byte loads per descriptor.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/orbhip.h"
#include "lds_optin.inc"

#define GRID_CELLS (ORBM_GRID_COLS * ORBM_GRID_ROWS)
#define SBP_CAPC 64                 // cached candidates per query (more -> the resolver re-enumerates that query inline)
#define SBP_WORK_PER_Q (SBP_CAPC + 2)  // u32 words of workspace per query: count, bin, list[SBP_CAPC]

struct Desc { uint32_t w[8]; };

static __device__ __forceinline__ Desc load_desc(const uint8_t* p) {
    Desc d;
    const uint4 a = *(const uint4*)p, b = *(const uint4*)(p + 16);
    d.w[0] = a.x; d.w[1] = a.y; d.w[2] = a.z; d.w[3] = a.w; d.w[4] = b.x; d.w[5] = b.y; d.w[6] = b.z; d.w[7] = b.w;
    return d;
}
static __device__ __forceinline__ int hamming(const Desc& a, const Desc& b) {
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += __popc(a.w[i] ^ b.w[i]);
    return s;
}
static __device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    return v;
}

// ============================================================================================================
// M1 / M10  brute-force Hamming tiles
// ============================================================================================================
// block = 256 train descriptors (one per thread, in registers) x 16 queries staged in LDS (broadcast reads);
// each of the 16 output rows is written coalesced.
static __global__ __launch_bounds__(256) void k_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    uint32_t* qs = (uint32_t*)orb_smem;  // [16][8]
    const int b = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * 16;
    q += (size_t)b * nq * 32; t += (size_t)b * nt * 32; out += (size_t)b * nq * nt;
    if (threadIdx.x < 128) {
        const int qi = i0 + (threadIdx.x >> 3);
        qs[threadIdx.x] = qi < nq ? ((const uint32_t*)q)[(size_t)qi * 8 + (threadIdx.x & 7)] : 0;
    }
    __syncthreads();
    if (j >= nt) return;
    const Desc d = load_desc(t + (size_t)j * 32);
    for (int r = 0; r < 16 && i0 + r < nq; r++) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) s += __popc(d.w[k] ^ qs[r * 8 + k]);
        out[(size_t)(i0 + r) * nt + j] = (uint16_t)s;
    }
}

// thread = query (descriptor in registers); train descriptors stream through LDS in tiles of 64, scanned in
// ascending index order with strict '<' so that ties keep the lower train index (cv::BFMatcher semantics).
static __global__ __launch_bounds__(256) void k_knn2(const uint8_t* q, const int32_t* nqv, int cap_q, const uint8_t* t,
                                                     const int32_t* ntv, int cap_t, int cstride, int32_t* out_idx, int32_t* out_dist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    uint32_t* ts = (uint32_t*)orb_smem;  // [64][8]
    const int b = blockIdx.y;
    const int nq = min(nqv[(size_t)b * cstride], cap_q), nt = min(ntv[(size_t)b * cstride], cap_t);
    const int i = blockIdx.x * 256 + threadIdx.x;
    q += (size_t)b * cap_q * 32; t += (size_t)b * cap_t * 32;
    Desc d;
    if (i < nq) d = load_desc(q + (size_t)i * 32);
    else { for (int k = 0; k < 8; k++) d.w[k] = 0; }
    int d0 = 256, d1 = 256, i0 = -1, i1 = -1;
    for (int base = 0; base < nt; base += 64) {
        const int n = min(64, nt - base);
        __syncthreads();
        for (int w = threadIdx.x; w < n * 8; w += 256) ts[w] = ((const uint32_t*)t)[(size_t)base * 8 + w];
        __syncthreads();
        for (int j = 0; j < n; j++) {
            int s = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) s += __popc(d.w[k] ^ ts[j * 8 + k]);
            if (s < d0) { d1 = d0; i1 = i0; d0 = s; i0 = base + j; }
            else if (s < d1) { d1 = s; i1 = base + j; }
        }
    }
    if (i < cap_q) {
        int32_t* oi = out_idx + ((size_t)b * cap_q + i) * 2;
        int32_t* od = out_dist + ((size_t)b * cap_q + i) * 2;
        const bool ok = i < nq;
        oi[0] = ok ? i0 : -1; oi[1] = ok ? i1 : -1; od[0] = ok ? d0 : 256; od[1] = ok ? d1 : 256;
    }
}

// ============================================================================================================
// M2  grid build: counting sort into a CSR whose cells list keypoint indices in ascending (= insertion) order
// ============================================================================================================
static __device__ __forceinline__ bool pos_in_grid(const orb_keypoint& kp, const orbm_grid_params& g, int& cell) {
    // Frame::PosInGrid, Frame.cc:852-862 (C round(): half away from zero)
    const int px = (int)roundf((kp.x - g.min_x) * g.grid_w_inv);
    const int py = (int)roundf((kp.y - g.min_y) * g.grid_h_inv);
    if (px < 0 || px >= ORBM_GRID_COLS || py < 0 || py >= ORBM_GRID_ROWS) return false;
    cell = px * ORBM_GRID_ROWS + py;
    return true;
}

// cells = GRID_CELLS (single camera) or 2*GRID_CELLS (fisheye rig: keypoints >= nleft go to the second half = mGridRight, Frame.cc:470-476)
static __global__ __launch_bounds__(256) void k_grid_build(const orb_keypoint* kps, const int32_t* nkp, int cstride, int cap_k,
                                                          orbm_grid_params g, int32_t* grid_start, int32_t* grid_idx, const int32_t* nleft_p,
                                                          int cells) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    int* cnt = (int*)orb_smem;            // [cells] counts -> starts
    int* fill = cnt + cells;              // [cells]
    int* scratch = fill + cells;          // [256]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(nkp[(size_t)b * cstride], cap_k);
    const int nleft = nleft_p ? nleft_p[b] : n;
    kps += (size_t)b * cap_k; grid_start += (size_t)b * (cells + 1); grid_idx += (size_t)b * cap_k;
    for (int c = tid; c < cells; c += 256) { cnt[c] = 0; fill[c] = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        int cell;
        if (pos_in_grid(kps[i], g, cell)) atomicAdd(&cnt[cell + (i >= nleft ? GRID_CELLS : 0)], 1);
    }
    __syncthreads();
    // exclusive scan of the counts: cells/256 per thread
    {
        const int per = cells / 256;
        int sum = 0;
        for (int k = 0; k < per; k++) sum += cnt[tid * per + k];
        scratch[tid] = sum;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int v = tid >= off ? scratch[tid - off] : 0;
            __syncthreads();
            scratch[tid] += v;
            __syncthreads();
        }
        int run = scratch[tid] - sum;
        for (int k = 0; k < per; k++) { const int c = cnt[tid * per + k]; cnt[tid * per + k] = run; run += c; }
        if (tid == 255) grid_start[cells] = run;
    }
    __syncthreads();
    for (int c = tid; c < cells; c += 256) grid_start[c] = cnt[c];
    for (int i = tid; i < n; i += 256) {
        int cell;
        if (pos_in_grid(kps[i], g, cell)) { cell += i >= nleft ? GRID_CELLS : 0; grid_idx[cnt[cell] + atomicAdd(&fill[cell], 1)] = i; }
    }
    __threadfence_block();
    __syncthreads();
    // restore insertion order inside each cell (cells hold a handful of entries)
    for (int c = tid; c < cells; c += 256) {
        const int s = cnt[c], m = fill[c];
        for (int a = 1; a < m; a++) {
            const int v = grid_idx[s + a];
            int p = a - 1;
            while (p >= 0 && grid_idx[s + p] > v) { grid_idx[s + p + 1] = grid_idx[s + p]; p--; }
            grid_idx[s + p + 1] = v;
        }
    }
}

// Frame::UndistortKeyPoints + AssignFeaturesToGrid of a frame in ONE workgroup (round 4): the two steps are a few microseconds of work per frame
// each, and as two launches (k_undistort: a thread per key point; k_grid_build: a 256-thread workgroup per frame walking the undistorted records
// again) they cost 0.036 + 0.036 ms per 512 frames — launch, fill and drain of two kernels that never hold more than one wave per SIMD.  Here a
// thread undistorts its key point, writes the record and keeps its cell; counts, the exclusive scan over the 64 x 48 cells (three cells per thread,
// DPP wave scans) and the fill follow in LDS.  Same outputs as the two calls, bit for bit.
#include "frame_undistort.inc"
#define UG_T 1024
static __global__ __launch_bounds__(UG_T) void k_undistort_grid(const orb_keypoint* kps, const int32_t* nkp, const int cstride, const int cap_k, const orbf_camera cam,
                                                               const orbm_grid_params g, orb_keypoint* kps_un, int32_t* grid_start, int32_t* grid_idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    int* cnt = (int*)orb_smem;            // [GRID_CELLS] counts -> starts
    int* fill = cnt + GRID_CELLS;         // [GRID_CELLS]
    int* wtot = fill + GRID_CELLS;        // [16] per-wave totals of the scan
    uint16_t* lst = (uint16_t*)(wtot + 16);   // [cap_k] the CSR index list, sorted per cell before it is written out
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = min(nkp[(size_t)b * cstride], cap_k);
    kps += (size_t)b * cap_k; kps_un += (size_t)b * cap_k; grid_start += (size_t)b * (GRID_CELLS + 1); grid_idx += (size_t)b * cap_k;
    for (int c = tid; c < GRID_CELLS; c += UG_T) { cnt[c] = 0; fill[c] = 0; }
    __syncthreads();
    constexpr int KP = 4;                 // key points per thread: cap_k <= KP * UG_T (checked on the host)
    int cell[KP];
#pragma unroll
    for (int k = 0; k < KP; k++) {
        const int i = tid + k * UG_T;
        cell[k] = -1;
        if (i < n) {
            orb_keypoint kp = kps[i];
            if (cam.dist[0] != 0.0f) undistort_point(cam, kp.x, kp.y, &kp.x, &kp.y);   // Frame.cc:879-883: k1 == 0 -> mvKeysUn = mvKeys
            kps_un[i] = kp;
            int c;
            if (pos_in_grid(kp, g, c)) { cell[k] = c; atomicAdd(&cnt[c], 1); }
        }
    }
    __syncthreads();
    {   // exclusive scan of the counts: three consecutive cells per thread, a DPP scan per wave, the sixteen wave totals through LDS
        constexpr int per = GRID_CELLS / UG_T;
        static_assert(per * UG_T == GRID_CELLS, "cells per thread");
        int c3[per], sum = 0;
#pragma unroll
        for (int k = 0; k < per; k++) { c3[k] = cnt[tid * per + k]; sum += c3[k]; }
        int incl = sum;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        int run = incl - sum;
        for (int w = 0; w < wv; w++) run += wtot[w];
#pragma unroll
        for (int k = 0; k < per; k++) { cnt[tid * per + k] = run; grid_start[tid * per + k] = run; run += c3[k]; }
        if (tid == UG_T - 1) grid_start[GRID_CELLS] = run;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KP; k++)
        if (cell[k] >= 0) lst[cnt[cell[k]] + atomicAdd(&fill[cell[k]], 1)] = (uint16_t)(tid + k * UG_T);
    __syncthreads();
    // insertion order inside each cell = ascending index (cells hold a handful of entries), then the list goes out coalesced
    for (int c = tid; c < GRID_CELLS; c += UG_T) {
        const int s = cnt[c], m = fill[c];
        for (int a = 1; a < m; a++) {
            const uint16_t v = lst[s + a];
            int p = a - 1;
            while (p >= 0 && lst[s + p] > v) { lst[s + p + 1] = lst[s + p]; p--; }
            lst[s + p + 1] = v;
        }
    }
    __syncthreads();
    const int tot = cnt[GRID_CELLS - 1] + fill[GRID_CELLS - 1];
    for (int i = tid; i < tot; i += UG_T) grid_idx[i] = lst[i];
}

// ============================================================================================================
// M3-M5  windowed projection search
// ============================================================================================================
struct SbpArgs {
    const orb_keypoint* kps; const uint8_t* desc; const float* u_right; const uint8_t* occupied0;
    const int32_t* nkp; int cstride, cap_k;
    const int32_t* grid_start; const int32_t* grid_idx;
    const orbm_query* queries; const uint8_t* qdesc; const int32_t* nq; int cap_q;
    orbm_search_params prm;
    int32_t* q_match; int32_t* kp_match; int32_t* nmatches;
    uint32_t* work;
    int cells;                  // grid cells per frame: GRID_CELLS, or 2*GRID_CELLS for a fisheye rig (second half = right camera)
    const int32_t* kp_link;     // rig: global index of each keypoint's stereo partner or -1 (mvLeftToRightMatch / mvRightToLeftMatch)
    int chi2_gate;              // Fuse: reprojection gate of ORBmatcher.cc:1791-1815
    float inv_sigma2[16];
    int32_t* q_dist;
    int32_t* serial_flag;       // [batch] written by k_sbp_frame: 1 = this frame is left to k_sbp_candidates_flagged -> k_sbp_resolve; nullptr: they take every frame
};

// One entry of the window's enumeration (position p of the frame's grid-index list, or -1): the tests of GetFeaturesInArea and the candidate filters
// that do not depend on matches made during the call.
static __device__ __forceinline__ void window_entry(const SbpArgs& A, const orbm_query& Q, const Desc& qd, const int n, const int p, const float r,
                                                    const int minLevel, const int maxLevel, const bool bCheckLevels, const orb_keypoint* kps,
                                                    const uint8_t* desc, const float* ur, const uint8_t* occ0, const int32_t* gi, bool& pass, bool& area,
                                                    int& idx, int& dist, int& oct) {
        if (p >= 0) {
            idx = gi[p];
            if (idx >= 0 && idx < n) {
                const orb_keypoint kp = kps[idx];
                oct = kp.octave;
                pass = true;
                if (bCheckLevels) {
                    if (oct < minLevel) pass = false;
                    if (maxLevel >= 0 && oct > maxLevel) pass = false;
                }
                const float distx = kp.x - Q.u, disty = kp.y - Q.v;
                if (!(fabsf(distx) < r && fabsf(disty) < r)) pass = false;
                area = pass;
                if (pass && occ0 && occ0[idx]) pass = false;
                if (pass && (Q.flags & ORBM_Q_STEREO) && ur) {
                    const float uR = ur[idx];
                    if (uR > 0 && fabsf(Q.u_right - uR) > r) pass = false;
                }
                if (pass && A.chi2_gate) {   // ORBmatcher.cc:1791-1815 (float e2 * float sigma, compared as double)
                    const float ex = Q.u - kp.x, ey = Q.v - kp.y;
                    const float s2 = A.inv_sigma2[oct & 15];
                    if (ur && ur[idx] >= 0) {
                        const float er = Q.u_right - ur[idx];
                        const float e2 = ex * ex + ey * ey + er * er;
                        if ((double)(e2 * s2) > 7.8) pass = false;
                    } else {
                        const float e2 = ex * ex + ey * ey;
                        if ((double)(e2 * s2) > 5.99) pass = false;
                    }
                }
                if (pass) dist = hamming(qd, load_desc(desc + (size_t)idx * 32));
            }
        }
}

// Wave-level enumeration of Frame::GetFeaturesInArea(u, v, radius, minLevel, maxLevel) in the reference's order
// (ix outer, iy inner, insertion order inside a cell == CSR order inside one grid column segment), with the
// candidate filters that do not depend on matches made during the call.  sink(pass, idx, dist, octave, inArea) is called
// wave-uniformly for every 64-entry chunk (lane = entry).
template <class Sink>
static __device__ __forceinline__ void enumerate_window(const SbpArgs& A, int b, const orbm_query& Q, const Desc& qd, int n, Sink&& sink) {
    const orbm_grid_params& g = A.prm.grid;
    const float r = Q.radius;
    // Frame.cc:779-806
    const int nMinCellX = max(0, (int)floorf((Q.u - g.min_x - r) * g.grid_w_inv));
    if (nMinCellX >= ORBM_GRID_COLS) return;
    const int nMaxCellX = min(ORBM_GRID_COLS - 1, (int)ceilf((Q.u - g.min_x + r) * g.grid_w_inv));
    if (nMaxCellX < 0) return;
    const int nMinCellY = max(0, (int)floorf((Q.v - g.min_y - r) * g.grid_h_inv));
    if (nMinCellY >= ORBM_GRID_ROWS) return;
    const int nMaxCellY = min(ORBM_GRID_ROWS - 1, (int)ceilf((Q.v - g.min_y + r) * g.grid_h_inv));
    if (nMaxCellY < 0) return;
    const int minLevel = Q.min_level, maxLevel = Q.max_level;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);   // Frame.cc:810
    const int lane = threadIdx.x & 63;
    const orb_keypoint* kps = A.kps + (size_t)b * A.cap_k;
    const uint8_t* desc = A.desc + (size_t)b * A.cap_k * 32;
    const float* ur = A.u_right ? A.u_right + (size_t)b * A.cap_k : nullptr;
    // the initial occupancy is a call constant (a keypoint that starts blocked can never be matched, so it stays blocked) — except on a rig
    // with stereo links, where a partner copy overwrites a blocked keypoint unconditionally: then only the resolver's live state applies
    const uint8_t* occ0 = (A.occupied0 && !A.kp_link) ? A.occupied0 + (size_t)b * A.cap_k : nullptr;
    const int32_t* gs = A.grid_start + (size_t)b * (A.cells + 1) + ((Q.flags & ORBM_Q_RIGHT) ? GRID_CELLS : 0);
    const int32_t* gi = A.grid_idx + (size_t)b * A.cap_k;
    if (nMaxCellY < nMinCellY) return;
    // The window's grid columns are contiguous CSR ranges [s_c, e_c).  Instead of walking them one after the other (a chain of dependent
    // global loads per column, a handful of entries each), their lengths are scanned across the wave and the concatenation — which is
    // exactly the reference's enumeration order, ix outer / iy inner / insertion order inside a cell — is processed 64 entries at a time.
    const int ncol = nMaxCellX - nMinCellX + 1;   // <= 64
    int cs = 0, ce = 0;
    if (lane < ncol) { cs = gs[(nMinCellX + lane) * ORBM_GRID_ROWS + nMinCellY]; ce = gs[(nMinCellX + lane) * ORBM_GRID_ROWS + nMaxCellY + 1]; }
    int incl = ce - cs;
    for (int off = 1; off < 64; off <<= 1) {
        if (off >= ncol) break;   // wave-uniform: lanes >= ncol hold zero lengths, the scan only has to cover the window's columns
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    const int total = __builtin_amdgcn_readlane(incl, ncol - 1);
    const int excl = incl - (ce - cs);
    for (int base = 0; base < total; base += 64) {
        const int t = base + lane;
        int p = -1;
        for (int c = 0; c < ncol; c++) {   // wave-uniform walk over the columns: lane t belongs to the column whose scan interval holds it
            const int xc = __builtin_amdgcn_readlane(excl, c), ic = __builtin_amdgcn_readlane(incl, c), sc = __builtin_amdgcn_readlane(cs, c);
            if (t >= xc && t < ic) p = sc + (t - xc);
            if (ic >= base + 64) break;       // later columns start beyond this chunk
        }
        bool pass = false, area = false;   // area: returned by GetFeaturesInArea (level + box); pass: also survives the call-constant filters
        int idx = 0, dist = 256, oct = 0;
        window_entry(A, Q, qd, n, p, r, minLevel, maxLevel, bCheckLevels, kps, desc, ur, occ0, gi, pass, area, idx, dist, oct);
        sink(pass, idx, dist, oct, area);
    }
}

#ifndef SBP_HALF
#define SBP_HALF 1
#endif
// Two queries per wave (32 lanes each).  A search window holds 10-30 grid entries, so a wave per query is mostly idle lanes on a chain of dependent
// global loads; halving the number of waves halves the kernel when it is latency bound.  Windows wider than 32 grid columns (any half of the wave) send
// the whole wave down the one-query-at-a-time path of k_sbp_candidates.
static __device__ __forceinline__ void sort_and_store_list(uint32_t* lst, uint32_t* w, const int count, const bool anyArea, const int lane) {
    // full-wave form (one query): sort the <= 64 cached entries by (distance, enumeration position) and write them out once
    const int nl = min(count, SBP_CAPC);
    uint32_t ent = lane < nl ? lst[lane] : 0u;
    if (count > 1 && count <= SBP_CAPC) {
        unsigned long long key = ~0ull;
        if (lane < count) key = ((unsigned long long)((((ent >> 16) & 0x1FFu) << 6) | (uint32_t)lane) << 32) | ent;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
            if ((k >> 1) >= count) break;   // wave-uniform
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const unsigned long long other = __shfl_xor(key, j);
                const bool keepMin = ((lane & j) == 0) == ((lane & k) == 0);
                key = keepMin ? (key < other ? key : other) : (key > other ? key : other);
            }
        }
        ent = (uint32_t)key;
    }
    if (lane < nl) w[2 + lane] = ent;
    if (lane == 0) { w[0] = (uint32_t)count | (anyArea ? 0x80000000u : 0u); w[1] = 0xFFFFFFFFu; }
}

static __device__ __forceinline__ void sbp_candidates2_block(const SbpArgs& A, const int b, const int qblock) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int half = lane >> 5, hl = lane & 31;
    uint32_t* lst2 = (uint32_t*)orb_smem + wv * 2 * SBP_CAPC;       // two lists of SBP_CAPC entries per wave
    // the query record and its descriptor do not depend on the counts: they are fetched in the same memory round trip (clamping the index
    // with nq instead of cap_q makes the compiler wait for nq first — one more dependent round trip on a kernel that is a chain of them)
    const int q0 = (qblock * 4 + wv) * 2;
    const int q = q0 + half;
    const size_t qslot = (size_t)b * A.cap_q + min(q, A.cap_q - 1);
    const orbm_query Q = A.queries[qslot];
    const Desc qd = load_desc(A.qdesc + qslot * 32);
    const int nq = min(A.nq[b], A.cap_q);
    const int n = min(A.nkp[(size_t)b * A.cstride], A.cap_k);
    if (q0 >= nq) return;
    const bool live = q < nq;
    const bool valid = live && (Q.flags & ORBM_Q_VALID);
    // window geometry (Frame.cc:779-806), per half
    const orbm_grid_params& g = A.prm.grid;
    const float r = Q.radius;
    const int nMinCellX = max(0, (int)floorf((Q.u - g.min_x - r) * g.grid_w_inv));
    const int nMaxCellX = min(ORBM_GRID_COLS - 1, (int)ceilf((Q.u - g.min_x + r) * g.grid_w_inv));
    const int nMinCellY = max(0, (int)floorf((Q.v - g.min_y - r) * g.grid_h_inv));
    const int nMaxCellY = min(ORBM_GRID_ROWS - 1, (int)ceilf((Q.v - g.min_y + r) * g.grid_h_inv));
    const bool window = valid && nMinCellX < ORBM_GRID_COLS && nMaxCellX >= 0 && nMinCellY < ORBM_GRID_ROWS && nMaxCellY >= 0 && nMaxCellY >= nMinCellY &&
                        nMaxCellX >= nMinCellX;
    const int ncol = window ? nMaxCellX - nMinCellX + 1 : 0;
    if (__ballot(ncol > 32)) {
        // rare: a window wider than half a wave.  Both queries take the one-query path, one after the other (wave-uniform)
        for (int hsel = 0; hsel < 2; hsel++) {
            const int qq = q0 + hsel;
            if (qq >= nq) break;
            const orbm_query QQ = A.queries[(size_t)b * A.cap_q + qq];
            uint32_t* w = A.work + ((size_t)b * A.cap_q + qq) * SBP_WORK_PER_Q;
            uint32_t* lst = lst2;
            int count = 0;
            bool anyArea = false;
            if (QQ.flags & ORBM_Q_VALID) {
                const Desc qd = load_desc(A.qdesc + ((size_t)b * A.cap_q + qq) * 32);
                enumerate_window(A, b, QQ, qd, n, [&](bool pass, int idx, int dist, int oct, bool area) {
                    const unsigned long long m = __ballot(pass);
                    if (pass) {
                        const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
                        if (pos < SBP_CAPC) lst[pos] = (uint32_t)idx | ((uint32_t)dist << 16) | ((uint32_t)(oct & 0x3F) << 25);
                    }
                    count += __popcll(m);
                    anyArea |= __ballot(area) != 0ull;
                });
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            sort_and_store_list(lst, w, count, anyArea, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        return;
    }
    uint32_t* lst = lst2 + half * SBP_CAPC;
    uint32_t* w = A.work + qslot * SBP_WORK_PER_Q;
    const int minLevel = Q.min_level, maxLevel = Q.max_level;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    const orb_keypoint* kps = A.kps + (size_t)b * A.cap_k;
    const uint8_t* desc = A.desc + (size_t)b * A.cap_k * 32;
    const float* ur = A.u_right ? A.u_right + (size_t)b * A.cap_k : nullptr;
    const uint8_t* occ0 = (A.occupied0 && !A.kp_link) ? A.occupied0 + (size_t)b * A.cap_k : nullptr;
    const int32_t* gs = A.grid_start + (size_t)b * (A.cells + 1) + ((Q.flags & ORBM_Q_RIGHT) ? GRID_CELLS : 0);
    const int32_t* gi = A.grid_idx + (size_t)b * A.cap_k;
    // column ranges of the window (lane hl = column), prefix scan inside the half
    int cs = 0, ce = 0;
    if (hl < ncol) { cs = gs[(nMinCellX + hl) * ORBM_GRID_ROWS + nMinCellY]; ce = gs[(nMinCellX + hl) * ORBM_GRID_ROWS + nMaxCellY + 1]; }
    int incl = ce - cs;
    // inclusive scan inside each 32-lane half, register-only: row_shr:1 / 2 / 4 / 8 inside the rows of 16 lanes, then row_bcast:15 carries a half's first
    // row into its second (rows 1 and 3) — five DPP adds where five __shfl_up steps are five dependent ds_bpermute round trips.  All 64 lanes are active.
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);
    const int excl = incl - (ce - cs);
    const int total = __shfl(incl, half * 32 + 31);     // lanes >= ncol carry the last column's inclusive sum
    int count = 0;
    bool anyArea = false;
    for (int base = 0; __ballot(base < total); base += 32) {
        const int t = base + hl;
        // column of entry t: the first column whose inclusive sum exceeds t (binary search over the half's 32 lanes, 5 cross-lane reads)
        int lo = 0, hi = 31;
#pragma unroll
        for (int st = 0; st < 5; st++) {
            const int mid = (lo + hi) >> 1;
            const int v = __shfl(incl, half * 32 + mid);
            if (v > t) hi = mid; else lo = mid + 1;
        }
        const int sc = __shfl(cs, half * 32 + lo), xc = __shfl(excl, half * 32 + lo);
        const int p = (t < total) ? sc + (t - xc) : -1;
        bool pass = false, area = false;
        int idx = 0, dist = 256, oct = 0;
        window_entry(A, Q, qd, n, p, r, minLevel, maxLevel, bCheckLevels, kps, desc, ur, occ0, gi, pass, area, idx, dist, oct);
        const unsigned long long m = __ballot(pass);
        const uint32_t hm = (uint32_t)(m >> (32 * half));
        if (pass) {
            const int pos = count + __popc(hm & ((1u << hl) - 1u));
            if (pos < SBP_CAPC) lst[pos] = (uint32_t)idx | ((uint32_t)dist << 16) | ((uint32_t)(oct & 0x3F) << 25);
        }
        count += __popc(hm);
        anyArea |= ((uint32_t)(__ballot(area) >> (32 * half))) != 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (__ballot(count > 32 && count <= SBP_CAPC)) {
        // rare: a list longer than half a wave needs the 64-lane network: the halves take turns
        for (int hsel = 0; hsel < 2; hsel++) {
            const int ch = __shfl(count, hsel * 32);
            const int aa = __shfl((int)anyArea, hsel * 32);
            const int qq = q0 + hsel;
            if (qq < nq) sort_and_store_list(lst2 + hsel * SBP_CAPC, A.work + ((size_t)b * A.cap_q + qq) * SBP_WORK_PER_Q, ch, aa != 0, lane);
        }
        return;
    }
    // half-wave sort of (dist << 6 | pos) << 32 | entry; lists longer than SBP_CAPC are left unsorted (the resolver re-enumerates those queries)
    const int nl = min(count, SBP_CAPC);
    uint32_t ent = hl < nl ? lst[hl] : 0u;
    uint32_t ent2 = (hl + 32 < nl) ? lst[hl + 32] : 0u;      // only when count > SBP_CAPC (unsorted copy-out)
    {
        const bool needSort = count > 1 && count <= 32;
        unsigned long long key = ~0ull;
        if (needSort && hl < count) key = ((unsigned long long)((((ent >> 16) & 0x1FFu) << 6) | (uint32_t)hl) << 32) | ent;
#pragma unroll
        for (int k = 2; k <= 32; k <<= 1) {
            if (!__ballot(needSort && (k >> 1) < count)) break;   // wave-uniform: neither half needs this stage
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const unsigned long long other = __shfl_xor(key, j);
                const bool keepMin = ((hl & j) == 0) == ((hl & k) == 0);
                key = keepMin ? (key < other ? key : other) : (key > other ? key : other);
            }
        }
        if (needSort) ent = (uint32_t)key;
    }
    if (live) {
        if (hl < nl) w[2 + hl] = ent;
        if (hl + 32 < nl) w[2 + hl + 32] = ent2;
        if (hl == 0) { w[0] = (uint32_t)count | (anyArea ? 0x80000000u : 0u); w[1] = 0xFFFFFFFFu; }
    }
}

static __global__ __launch_bounds__(256) void k_sbp_candidates2(SbpArgs A) { sbp_candidates2_block(A, blockIdx.y, blockIdx.x); }

static __global__ __launch_bounds__(256) void k_sbp_candidates(SbpArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    uint32_t* lst = (uint32_t*)orb_smem + (threadIdx.x >> 6) * SBP_CAPC;   // this wave's compacted list: sorted in LDS, written to the workspace once
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nq = min(A.nq[b], A.cap_q);
    if (q >= nq) return;
    const int n = min(A.nkp[(size_t)b * A.cstride], A.cap_k);
    const orbm_query Q = A.queries[(size_t)b * A.cap_q + q];
    uint32_t* w = A.work + ((size_t)b * A.cap_q + q) * SBP_WORK_PER_Q;
    int count = 0;
    bool anyArea = false;   // vIndices.empty() of the reference refers to this, not to the filtered list
    if (Q.flags & ORBM_Q_VALID) {
        const Desc qd = load_desc(A.qdesc + ((size_t)b * A.cap_q + q) * 32);
        enumerate_window(A, b, Q, qd, n, [&](bool pass, int idx, int dist, int oct, bool area) {
            const unsigned long long m = __ballot(pass);
            if (pass) {
                const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
                if (pos < SBP_CAPC) lst[pos] = (uint32_t)idx | ((uint32_t)dist << 16) | ((uint32_t)(oct & 0x3F) << 25);
            }
            count += __popcll(m);
            anyArea |= __ballot(area) != 0ull;
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nl = min(count, SBP_CAPC);
        uint32_t ent = lane < nl ? lst[lane] : 0u;
        if (count > 1 && count <= SBP_CAPC) {
            // sort the cached list by (distance, enumeration position): the resolver then takes "first unblocked" instead of min-reducing.
            // Wave-wide bitonic sort of (dist << 6 | pos) << 32 | entry.
            unsigned long long key = ~0ull;
            if (lane < count) key = ((unsigned long long)((((ent >> 16) & 0x1FFu) << 6) | (uint32_t)lane) << 32) | ent;
            // the network only has to cover the first 2^m >= count lanes (the others hold ~0 and would stay at the end anyway): a typical list of
            // <= 8 entries takes 6 of the 21 compare-exchange steps, each a dependent cross-lane shuffle in a latency-bound kernel
#pragma unroll
            for (int k = 2; k <= 64; k <<= 1) {
                if ((k >> 1) >= count) break;   // wave-uniform
#pragma unroll
                for (int j = k >> 1; j > 0; j >>= 1) {
                    const unsigned long long other = __shfl_xor(key, j);
                    const bool keepMin = ((lane & j) == 0) == ((lane & k) == 0);
                    key = keepMin ? (key < other ? key : other) : (key > other ? key : other);
                }
            }
            ent = (uint32_t)key;
        }
        if (lane < nl) w[2 + lane] = ent;
    }
    if (lane == 0) { w[0] = (uint32_t)count | (anyArea ? 0x80000000u : 0u); w[1] = 0xFFFFFFFFu; }
}

// M12 Fuse (search half): queries are independent -> one wave per query, first minimum of (dist, enumeration position).
static __global__ __launch_bounds__(256) void k_fuse(SbpArgs A) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= A.cap_q) return;
    const int nq = min(A.nq[b], A.cap_q);
    int bestIdx = -1, bestDist = 256;
    if (q < nq) {
        const int n = min(A.nkp[(size_t)b * A.cstride], A.cap_k);
        const orbm_query Q = A.queries[(size_t)b * A.cap_q + q];
        if (Q.flags & ORBM_Q_VALID) {
            const Desc qd = load_desc(A.qdesc + ((size_t)b * A.cap_q + q) * 32);
            uint32_t k1 = 0xFFFFFFFFu;
            int i1 = -1, seen = 0;
            enumerate_window(A, b, Q, qd, n, [&](bool pass, int idx, int dist, int oct, bool) {
                const unsigned long long m = __ballot(pass);
                if (pass) {
                    const uint32_t key = ((uint32_t)dist << 20) | (uint32_t)((seen + __popcll(m & ((1ull << lane) - 1ull))) & 0xFFFFF);
                    if (key < k1) { k1 = key; i1 = idx; }
                }
                seen += __popcll(m);
            });
            const uint32_t m1 = wave_min_u32(k1);
            if (m1 != 0xFFFFFFFFu) {
                const int l1 = __ffsll((long long)__ballot(k1 == m1)) - 1;
                bestIdx = __shfl(i1, l1);
                bestDist = (int)(m1 >> 20);
            }
        }
    }
    if (lane == 0) {
        const bool ok = bestIdx >= 0 && bestDist <= A.prm.th_dist;
        A.q_match[(size_t)b * A.cap_q + q] = ok ? bestIdx : -1;
        A.q_dist[(size_t)b * A.cap_q + q] = bestDist;
        if (ok) atomicAdd(&A.nmatches[b], 1);
    }
}

// One wave per frame walks the queries in index order (ORBmatcher.cc:65 / :2265 loop order) and applies the
// reference's accept rules against the live occupancy; distances come from k_sbp_candidates.
// The serial chain touches LDS only: queries are staged 64 at a time (count, HAS_OBS flag and the first SBP_STAGE candidates
// of each, loaded lane-parallel), and the rotation-histogram bins are computed after the loop (they do not feed back).
#ifndef SBP_FUSED_FRAME
#define SBP_FUSED_FRAME 1          // 0: every frame takes k_sbp_candidates2 -> k_sbp_resolve (the round-1..3 form: the A/B build and the CPU tier's second pass)
#endif
#define SBP_STAGE 16
struct __attribute__((packed, aligned(4))) SbpRow4 { uint32_t a, b, c, d; };   // 4 list entries; rows of the work buffer are 8-byte aligned
static __global__ __launch_bounds__(64) void k_sbp_resolve(SbpArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (A.serial_flag && A.serial_flag[b] == 0) return;   // resolved by k_sbp_frame
    const int n = min(A.nkp[(size_t)b * A.cstride], A.cap_k);
    const int nq = min(A.nq[b], A.cap_q);
    int* hist = (int*)orb_smem;                       // [32]
    int* ctl = hist + 32;                             // [8]
    uint32_t* sEnt = (uint32_t*)(ctl + 8);            // [64][SBP_STAGE]
    int* sCnt = (int*)(sEnt + 64 * SBP_STAGE);        // [64] count | hasObs << 30
    uint8_t* occ = (uint8_t*)(sCnt + 64);             // [cap_k] holder has Observations()>0
    const bool initMode = A.prm.mode == ORBM_MODE_INIT;
    uint16_t* mdist = (uint16_t*)(occ + ((A.cap_k + 15) & ~15));   // [cap_k] vMatchedDistance (INIT only; 0xFFFF = INT_MAX)
    uint16_t* holder = mdist + A.cap_k;                            // [cap_k] vnMatches21      (INIT only; 0xFFFF = -1)
    int32_t* q_match = A.q_match + (size_t)b * A.cap_q;
    int32_t* kp_match = A.kp_match + (size_t)b * A.cap_k;
    const uint8_t* occ0 = A.occupied0 ? A.occupied0 + (size_t)b * A.cap_k : nullptr;
    for (int i = lane; i < A.cap_k; i += 64) { occ[i] = (i < n && occ0 && occ0[i]) ? 1 : 0; kp_match[i] = -1; }
    if (initMode) for (int i = lane; i < A.cap_k; i += 64) { mdist[i] = 0xFFFFu; holder[i] = 0xFFFFu; }
    for (int i = lane; i < A.cap_q; i += 64) q_match[i] = -1;
    if (lane < 32) hist[lane] = 0;
    __syncthreads();
    const int mode = A.prm.mode, th = A.prm.th_dist;
    const float ratio = A.prm.nn_ratio;
    const bool ori = mode == ORBM_MODE_BEST_ONLY && A.prm.check_orientation;
    const orb_keypoint* kps = A.kps + (size_t)b * A.cap_k;
    const orbm_query* queries = A.queries + (size_t)b * A.cap_q;
    int nmatches = 0;
    bool skipTwin = false;
    const int32_t* link = A.kp_link ? A.kp_link + (size_t)b * A.cap_k : nullptr;
    for (int q0 = 0; q0 < nq; q0 += 64) {
        {   // stage this block of queries (lane-parallel loads)
            const int q = q0 + lane;
            int c = 0, obs = 0, fl = 0;
            if (q < nq) {
                const uint32_t* w = A.work + ((size_t)b * A.cap_q + q) * SBP_WORK_PER_Q;
                c = (int)(w[0] & 0x7FFFFFFFu);
                const uint32_t qf = queries[q].flags;
                obs = (qf & ORBM_Q_HAS_OBS) ? 1 : 0;
                // bit 29: GetFeaturesInArea returned something; bit 28: right-camera twin of the previous query; bit 27: right camera
                fl = (int)((w[0] >> 31) << 29) | ((qf & ORBM_Q_TWIN) ? 1 << 28 : 0) | ((qf & ORBM_Q_RIGHT) ? 1 << 27 : 0);
                // up to SBP_STAGE entries, four 16-byte loads in flight (entries past the count are never read back)
                const int m = min(c, SBP_STAGE);
                SbpRow4 v[SBP_STAGE / 4];
#pragma unroll
                for (int u = 0; u < SBP_STAGE / 4; u++) v[u] = (4 * u < m) ? *(const SbpRow4*)(w + 2 + 4 * u) : SbpRow4{0u, 0u, 0u, 0u};
#pragma unroll
                for (int u = 0; u < SBP_STAGE / 4; u++)
                    if (4 * u < m) *(uint4*)&sEnt[lane * SBP_STAGE + 4 * u] = make_uint4(v[u].a, v[u].b, v[u].c, v[u].d);
            }
            sCnt[lane] = c | (obs << 30) | fl;
        }
        __syncthreads();
        const int qend = min(64, nq - q0);
        // accept rule of one query given its best / second-best unblocked candidates (updates the rig twin-skip state)
        auto decide = [&](uint32_t eb1, bool have2, uint32_t eb2, bool rightCam) -> bool {
            const int bestDist = (int)((eb1 >> 16) & 0x1FF);
            if (bestDist > th) return false;
            if (mode == ORBM_MODE_LOCAL_MAP) {
                const int bestLevel = (int)((eb1 >> 25) & 0x3F);
                const int bestDist2 = have2 ? (int)((eb2 >> 16) & 0x1FF) : 256;
                const int bestLevel2 = have2 ? (int)((eb2 >> 25) & 0x3F) : -1;
                // ORBmatcher.cc:160-178
                if (bestLevel == bestLevel2 && (float)bestDist > ratio * (float)bestDist2) { if (!rightCam) skipTwin = true; return false; }
                return bestLevel != bestLevel2 || (float)bestDist <= ratio * (float)bestDist2;
            }
            if (initMode) {   // ORBmatcher.cc:914-918: bestDist < (float)bestDist2*mfNNratio, bestDist2 = INT_MAX when alone
                const float bestDist2 = have2 ? (float)(int)((eb2 >> 16) & 0x1FF) : (float)INT_MAX;
                return (float)bestDist < bestDist2 * ratio;
            }
            return true;  // ORBmatcher.cc:2372
        };
        // one query resolved by the whole wave (any candidate count, every mode)
        auto resolve_one = [&](const int i) {
            const int q = q0 + i;
            const int cw = sCnt[i];
            const int count = cw & 0x07FFFFFF;
            // rig twins (ORBmatcher.cc:166-167 / :2332): a `continue` taken while handling the left camera skips the right camera too
            const bool twin = (cw >> 28) & 1, rightCam = (cw >> 27) & 1;
            if (!twin) skipTwin = false;
            if (twin && skipTwin) return;
            if (mode == ORBM_MODE_BEST_ONLY && !rightCam && !((cw >> 29) & 1)) skipTwin = true;   // left window empty
            if (count == 0) return;
            // best / second-best among the candidates that are not blocked by the live state.  The cached list is sorted by
            // (distance, enumeration position) — the order in which the reference's strict-'<' scan would rank them — so the best is
            // simply the first unblocked entry and the second-best the next one: one ballot instead of wave-wide min reductions.
            uint32_t eb1 = 0, eb2 = 0;
            bool have2 = false;
            if (count <= SBP_CAPC) {
                uint32_t e = 0;
                bool unb = false;
                if (lane < count) {
                    e = count <= SBP_STAGE ? sEnt[i * SBP_STAGE + lane] : A.work[((size_t)b * A.cap_q + q) * SBP_WORK_PER_Q + 2 + lane];
                    unb = !(initMode ? (uint32_t)mdist[e & 0xFFFF] <= ((e >> 16) & 0x1FF) : occ[e & 0xFFFF] != 0);
                }
                const unsigned long long um = __ballot(unb);
                if (um == 0ull) return;  // every candidate already holds an observed point
                eb1 = (uint32_t)__builtin_amdgcn_readlane((int)e, __ffsll((long long)um) - 1);   // wave-uniform lane index -> v_readlane, no LDS round trip
                const unsigned long long um2 = um & (um - 1ull);
                if (um2) { have2 = true; eb2 = (uint32_t)__builtin_amdgcn_readlane((int)e, __ffsll((long long)um2) - 1); }
            } else {  // rare: more candidates than the cache holds -> re-enumerate this query against the live state
                uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu, e1 = 0, e2 = 0;   // per-lane two smallest keys; key = dist<<20 | enumeration position
                const orbm_query Q = queries[q];
                const Desc qd = load_desc(A.qdesc + ((size_t)b * A.cap_q + q) * 32);
                int seen = 0;
                enumerate_window(A, b, Q, qd, n, [&](bool pass, int idx, int dist, int oct, bool) {
                    const unsigned long long m = __ballot(pass);
                    if (pass && !(initMode ? (int)mdist[idx] <= dist : occ[idx] != 0)) {
                        const uint32_t pos = (uint32_t)(seen + __popcll(m & ((1ull << lane) - 1ull)));
                        const uint32_t key = ((uint32_t)dist << 20) | (pos & 0xFFFFF);
                        const uint32_t e = (uint32_t)idx | ((uint32_t)dist << 16) | ((uint32_t)(oct & 0x3F) << 25);
                        if (key < k1) { k2 = k1; e2 = e1; k1 = key; e1 = e; }
                        else if (key < k2) { k2 = key; e2 = e; }
                    }
                    seen += __popcll(m);
                });
                const uint32_t m1 = wave_min_u32(k1);
                if (m1 == 0xFFFFFFFFu) return;
                const bool iBest = k1 == m1;
                eb1 = __shfl(e1, __ffsll((long long)__ballot(iBest)) - 1);
                const uint32_t c2 = iBest ? k2 : k1;
                const uint32_t m2 = wave_min_u32(c2);
                const unsigned long long bm2 = __ballot(c2 == m2 && m2 != 0xFFFFFFFFu);
                if (bm2) { have2 = true; eb2 = __shfl(iBest ? e2 : e1, __ffsll((long long)bm2) - 1); }
            }
            const int bestDist = (int)((eb1 >> 16) & 0x1FF), bestIdx = (int)(eb1 & 0xFFFF);
            const bool accept = decide(eb1, have2, eb2, rightCam);
            if (accept && initMode) {
                const int prev = holder[bestIdx];   // vnMatches21[bestIdx2] (:920-924): the displaced F1 keypoint loses its match
                if (prev != 0xFFFF) nmatches--;
                nmatches++;
                __syncthreads();
                if (lane == 0) {
                    if (prev != 0xFFFF) q_match[prev] = -1;
                    q_match[q] = bestIdx;
                    holder[bestIdx] = (uint16_t)q;
                    mdist[bestIdx] = (uint16_t)bestDist;
                    int bin = 30;
                    if (A.prm.check_orientation) {   // :933-943, factor = HISTO_LENGTH/360.0f
                        float rot = queries[q].angle - kps[bestIdx].angle;
                        if (rot < 0.0f) rot += 360.0f;
                        bin = (int)roundf(rot * (ORBM_HISTO_LENGTH / 360.0f));
                        if (bin == ORBM_HISTO_LENGTH) bin = 0;
                        bin = max(0, min(bin, 29));
                        hist[bin]++;
                    }
                    A.work[((size_t)b * A.cap_q + q) * SBP_WORK_PER_Q + 1] = (uint32_t)bin;
                }
                __syncthreads();
            } else if (accept) {
                nmatches++;
                // rig, local-map search: the match is copied to the keypoint's stereo partner in the other camera (:172-176, :239-243)
                const int partner = (link && mode == ORBM_MODE_LOCAL_MAP) ? link[bestIdx] : -1;
                if (partner >= 0) nmatches++;
                if (lane == 0) {
                    occ[bestIdx] = (uint8_t)((cw >> 30) & 1);
                    kp_match[bestIdx] = q;
                    q_match[q] = bestIdx;
                    if (partner >= 0) { occ[partner] = (uint8_t)((cw >> 30) & 1); kp_match[partner] = q; }
                }
                __syncthreads();  // single-wave block: orders lane 0's LDS write before the next query's reads
            }
                };
        for (int i = 0; i < qend;) {
            // Batched fast path (occupancy modes): up to 8 consecutive queries with <= SBP_STAGE cached candidates each are packed over the wave by
            // their list lengths (a query of c candidates takes max(c, 1) lanes, 64 lanes in all); ONE read of the occupancy serves all of them.
            // stk[k] = first lane of query k of the batch (scalar registers), stk[k] = total for k >= nb.
            int nb = 0, wq = 65;
            int stk[9];
            stk[0] = 0;
            if (!initMode) {
                if (lane < 8 && i + lane < qend) { const int cq = sCnt[i + lane] & 0x07FFFFFF; wq = cq > SBP_STAGE ? 65 : max(cq, 1); }
                bool open = true;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int end = stk[k] + __builtin_amdgcn_readlane(wq, k);
                    open = open && end <= 64;
                    stk[k + 1] = open ? end : stk[k];
                    nb += open ? 1 : 0;
                }
            }
            if (nb < 2) { resolve_one(i); i++; continue; }
            const int total = stk[8];
            int g = 0, myst = 0, myend = total;
#pragma unroll
            for (int k = 1; k < 8; k++) {
                if (lane >= stk[k]) { g++; myst = stk[k]; }
                if (lane < stk[8 - k]) myend = stk[8 - k];
            }
            g = min(g, nb - 1);
            const bool inb = lane < total;
            const int sl = lane - myst;
            const int cwl = inb ? sCnt[i + g] : 0;
            const bool valid = inb && sl < (cwl & 0x07FFFFFF);
            const uint32_t e = valid ? sEnt[(i + g) * SBP_STAGE + sl] : 0u;
            const int cidx = (int)(e & 0xFFFF);
            bool unb = valid && occ[cidx] == 0;
            // Optimistic parallel resolution: all queries of the batch decide at once against the occupancy at the batch start.  A committed accept
            // changes a later query's decision only through that query's best / second-best unblocked candidate (the list is sorted: blocking
            // anything behind them changes nothing), or when both accept the same keypoint (store order).  Such a (k < g) pair is a clash: the
            // batch is committed up to the first clashing query and the next batch starts there.  The 8 x 8 (k, g) pairs are tested one per lane.
            // Rig twins, whose `continue` rules chain, take the serial loop below.
            if (link == nullptr && !(__ballot(inb && ((cwl >> 28) & 1)))) {
                const int width = myend - myst;
                const unsigned long long wmask = width >= 64 ? ~0ull : ((1ull << width) - 1ull);
                const bool head = inb && sl == 0;
                // pair lane L = 8 g2 + k2: first lanes of queries k2 and g2 (independent of the decisions: issued before them)
                const int k2 = lane & 7, g2 = lane >> 3;
                int sk = 0, sg = 0;
#pragma unroll
                for (int k = 1; k < 8; k++) { sk = k2 == k ? stk[k] : sk; sg = g2 == k ? stk[k] : sg; }
                const bool pairOn = k2 < g2 && g2 < nb;
                const unsigned long long um = (__ballot(unb) >> myst) & wmask;
                const bool have1 = um != 0ull;
                const unsigned long long um2 = um & (um - 1ull);
                const bool have2 = um2 != 0ull;
                const uint32_t eb1 = (uint32_t)__shfl((int)e, myst + (have1 ? __ffsll((long long)um) - 1 : 0));
                const uint32_t eb2 = (uint32_t)__shfl((int)e, myst + (have2 ? __ffsll((long long)um2) - 1 : 0));
                bool accept = false, skipAfter = false;
                if (inb) {
                    if (mode == ORBM_MODE_BEST_ONLY && !((cwl >> 29) & 1)) skipAfter = true;   // left window empty (no twins here: rightCam == 0)
                    if (have1) {
                        const int bestDist = (int)((eb1 >> 16) & 0x1FF);
                        if (bestDist <= th) {
                            if (mode == ORBM_MODE_LOCAL_MAP) {
                                const int bestLevel = (int)((eb1 >> 25) & 0x3F);
                                const int bestDist2 = have2 ? (int)((eb2 >> 16) & 0x1FF) : 256;
                                const int bestLevel2 = have2 ? (int)((eb2 >> 25) & 0x3F) : -1;
                                if (bestLevel == bestLevel2 && (float)bestDist > ratio * (float)bestDist2) skipAfter = true;
                                else accept = bestLevel != bestLevel2 || (float)bestDist <= ratio * (float)bestDist2;
                            } else accept = true;
                        }
                    }
                }
                const int aidx = accept ? (int)(eb1 & 0xFFFF) : -1;
                const int bi1 = have1 ? (int)(eb1 & 0xFFFF) : -2, bi2 = have2 ? (int)(eb2 & 0xFFFF) : -2;
                const int ak = __shfl(aidx, sk & 63), b1 = __shfl(bi1, sg & 63), b2 = __shfl(bi2, sg & 63);
                const unsigned long long cm = __ballot(pairOn && ak >= 0 && (ak == b1 || ak == b2));
                const int np = cm ? (__ffsll((long long)cm) - 1) >> 3 : nb;   // pair lanes are ordered by g2; query 0 never clashes
                const bool lead = head && accept && g < np;
                if (lead) {
                    const int q = q0 + i + g;
                    occ[aidx] = (uint8_t)((cwl >> 30) & 1);
                    kp_match[aidx] = q;
                    q_match[q] = aidx;
                }
                nmatches += __popcll(__ballot(lead));
                skipTwin = __ballot(head && g == np - 1 && skipAfter) != 0ull;
                __syncthreads();
                i += np;
                continue;
            }
            for (int j = 0, stj = 0, wj = 0; j < nb; j++, stj += wj) {
                wj = __builtin_amdgcn_readlane(wq, j);
                const int q = q0 + i + j;
                const int cw = __builtin_amdgcn_readlane(cwl, stj);
                const bool twin = (cw >> 28) & 1, rightCam = (cw >> 27) & 1;
                if (!twin) skipTwin = false;
                if (twin && skipTwin) continue;
                if (mode == ORBM_MODE_BEST_ONLY && !rightCam && !((cw >> 29) & 1)) skipTwin = true;
                const unsigned long long um = (__ballot(unb) >> stj) & (wj >= 64 ? ~0ull : ((1ull << wj) - 1ull));
                if (um == 0ull) continue;   // no candidates, or every candidate already holds an observed point
                const uint32_t eb1 = (uint32_t)__builtin_amdgcn_readlane((int)e, stj + __ffsll((long long)um) - 1);
                const unsigned long long um2 = um & (um - 1ull);
                const bool have2 = um2 != 0ull;
                const uint32_t eb2 = have2 ? (uint32_t)__builtin_amdgcn_readlane((int)e, stj + __ffsll((long long)um2) - 1) : 0u;
                if (!decide(eb1, have2, eb2, rightCam)) continue;
                const int bestIdx = (int)(eb1 & 0xFFFF), obs = (cw >> 30) & 1;
                nmatches++;
                const int partner = (link && mode == ORBM_MODE_LOCAL_MAP) ? link[bestIdx] : -1;
                if (partner >= 0) nmatches++;
                if (lane == 0) {
                    occ[bestIdx] = (uint8_t)obs;
                    kp_match[bestIdx] = q;
                    q_match[q] = bestIdx;
                    if (partner >= 0) { occ[partner] = (uint8_t)obs; kp_match[partner] = q; }
                }
                if (valid && (cidx == bestIdx || cidx == partner)) unb = obs == 0;   // what the later queries of this batch will see
            }
            __syncthreads();   // lane 0's occupancy writes are ordered before the next batch's reads
            i += nb;
        }
        __syncthreads();  // the staging area is rewritten by the next block
    }
    __threadfence_block();
    __syncthreads();
    if (initMode) {
        for (int i = lane; i < n; i += 64) kp_match[i] = holder[i] == 0xFFFF ? -1 : (int)holder[i];
        if (A.prm.check_orientation) {   // :949-973: every i1 accepted at some point sits in its bin, even if displaced later
            if (lane == 0) {
                int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
                for (int i = 0; i < ORBM_HISTO_LENGTH; i++) {
                    const int s = hist[i];
                    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                    else if (s > max3) { max3 = s; ind3 = i; }
                }
                if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
                else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
                ctl[0] = ind1; ctl[1] = ind2; ctl[2] = ind3; ctl[3] = 0;
            }
            __syncthreads();
            const int ind1 = ctl[0], ind2 = ctl[1], ind3 = ctl[2];
            for (int q = lane; q < nq; q += 64) {
                if (q_match[q] >= 0) {
                    const int bin = (int)A.work[((size_t)b * A.cap_q + q) * SBP_WORK_PER_Q + 1];
                    if (bin != ind1 && bin != ind2 && bin != ind3) { q_match[q] = -1; atomicAdd(&ctl[3], 1); }
                }
            }
            __threadfence_block();
            __syncthreads();
            nmatches -= ctl[3];
        }
        if (lane == 0) A.nmatches[b] = nmatches;
        return;
    }
    if (ori) {
        // rotation histogram (ORBmatcher.cc:2387-2395: factor = 1/HISTO_LENGTH quirk, C round()) of every accepted match
        for (int q = lane; q < nq; q += 64) {
            const int idx = q_match[q];
            if (idx >= 0) {
                float rot = queries[q].angle - kps[idx].angle;
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * (1.0f / ORBM_HISTO_LENGTH));
                if (bin == ORBM_HISTO_LENGTH) bin = 0;
                bin = max(0, min(bin, 31));
                atomicAdd(&hist[bin], 1);
                A.work[((size_t)b * A.cap_q + q) * SBP_WORK_PER_Q + 1] = (uint32_t)bin;
            }
        }
        __threadfence_block();
        __syncthreads();
        if (lane == 0) {  // ComputeThreeMaxima, ORBmatcher.cc:2654-2695
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < ORBM_HISTO_LENGTH; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
            ctl[0] = ind1; ctl[1] = ind2; ctl[2] = ind3; ctl[3] = 0;
        }
        __syncthreads();
        const int ind1 = ctl[0], ind2 = ctl[1], ind3 = ctl[2];
        for (int q = lane; q < nq; q += 64) {
            if (q_match[q] >= 0) {
                const int bin = (int)A.work[((size_t)b * A.cap_q + q) * SBP_WORK_PER_Q + 1];
                if (bin != ind1 && bin != ind2 && bin != ind3) {
                    kp_match[q_match[q]] = -2;      // CurrentFrame.mvpMapPoints[...] = NULL, :2499 (-2: claimed during the call, then culled)
                    atomicAdd(&ctl[3], 1);
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        nmatches -= ctl[3];
    }
    // a query's match is reported only while its keypoint still holds it
    for (int q = lane; q < nq; q += 64)
        if (q_match[q] >= 0 && kp_match[q_match[q]] != q) q_match[q] = -1;
    if (lane == 0) A.nmatches[b] = nmatches;
}

// ------------------------------------------------------------------------------------------------------------
// Round 4: the whole projection search of a frame in ONE workgroup — k_sbp_frame.
//
// (1) The frame is staged once in LDS: the grid in CSR order (position p -> key point index, octave, x, y side by side), the 64 x 48 CSR heads and
//     the descriptors (rows of 9 dwords: an odd pitch spreads random row reads over the banks) — ~60 bytes per key point + 6 KB.
//     k_sbp_candidates2 walked query -> CSR heads -> index -> record -> descriptor as a chain of dependent L2 gathers (0.21 ms per 512 frames at
//     3 % of the VALU, traffic 2.6-4.9 x algorithmic); from LDS a link costs ~100 cycles, so the chain is given to ONE LANE per query: every query of
//     the frame walks its window at once, in the reference's order (grid columns outer, CSR order inside a column = iy inner, insertion order inside
//     a cell: Frame.cc:779-850).  That order IS ascending CSR position, so a candidate's rank in the serial strict-'<' scans of ORBmatcher.cc:137-158 /
//     :2355-2368 is the rank of its key  dist << 22 | p << 6 | octave  — a plain unsigned compare.  A lane keeps its SBPF_SD smallest keys sorted in
//     registers; what falls out of that list goes, unordered, to the query's row of the workspace (SBPF_ROW = 128 keys).
// (2) The serial accept loop as parallel fixed-point rounds.  The loop's state is the occupancy: candidate p is blocked for query q iff an EARLIER
//     query q' < q accepted it while holding an observed point (ORBmatcher.cc:125-127 / :2347-2349; the call's initial occupancy is filtered in (1)).
//     So a query's decision d[q] = F_q(d[0..q-1]) depends on the earlier decisions only, through blk[p] = the smallest accepting-and-observed query
//     of p: best / second-best = the two smallest keys of q with blk[p] >= q.  Jacobi iteration d_{r+1}[q] = F_q(d_r) on this strictly
//     lower-triangular system reaches its unique fixed point — the serial result, by induction over q: after round r every query whose dependency
//     chain is shorter than r is final — and stops after the first round that changes nothing (at most nq + 1 rounds; 8-9 on the benchmark's
//     frames, where the one-wave walk of k_sbp_resolve takes ~1000 dependent steps).  blk is double-buffered and its entries carry the round that
//     wrote them ((0xFFFF - r) << 16 | q under atomicMin: a newer round overrides, an entry of an older round reads as "nobody"), so a round is ONE
//     workgroup barrier.  A query whose kept keys are all claimed reads on in its workspace row (two smallest unblocked keys of an unordered set, read
//     by the whole wave); one that reads past SBPF_SD + SBPF_ROW candidates raises the frame's flag (the one-wave walk below redoes it).  All exact.
// (3) What the walk leaves behind is reconstructed from the fixed point: kp_match[c] = the LAST accepter of c (only accepters without an observed
//     point can share a key point), nmatches = the number of accepters, then the rotation-histogram cull and the "a query's match is reported
//     only while its key point still holds it" filter exactly as k_sbp_resolve applies them.
// Queries beyond the workgroup's 1024 threads (cap_q = nFeatures + 64 is a little more than 1024 at nFeatures = 1000) keep their lists in LDS
// instead of registers (TAIL): the register budget of the common case stays that of one query per thread.
// A frame this form does not cover — rig twins / right-camera queries, a list read past its kept keys — raises serial_flag[b] and is redone by the gated launches behind it
// (k_sbp_candidates_flagged -> k_sbp_resolve).  INIT mode, rigs with stereo links and frames beyond the LDS limits never come here (sbp_launch).
struct SbpfFrame { const uint16_t* gs; const uint32_t* ge; const float* gx; const float* gy; const uint32_t* dsc; const uint8_t* occ0; const float* ur; int n; };
#define SBPF_DP 9          // descriptor row pitch in LDS, dwords (odd: random rows spread over the banks)
// One lane's walk over the window of its query in the reference's order (Frame.cc:779-850) on the staged frame; sink(p, idx, oct, dist) sees every
// candidate that GetFeaturesInArea returns and that passes the call-constant filters (initial occupancy, stereo gate), with its Hamming distance.
template <class Sink>
static __device__ __forceinline__ void sbpf_walk(const SbpfFrame& F, const orbm_grid_params& g, const orbm_query& Q, const Desc& qd, Sink&& sink) {
    const float r = Q.radius;
    // Frame.cc:779-806
    const int nMinCellX = max(0, (int)floorf((Q.u - g.min_x - r) * g.grid_w_inv));
    const int nMaxCellX = min(ORBM_GRID_COLS - 1, (int)ceilf((Q.u - g.min_x + r) * g.grid_w_inv));
    const int nMinCellY = max(0, (int)floorf((Q.v - g.min_y - r) * g.grid_h_inv));
    const int nMaxCellY = min(ORBM_GRID_ROWS - 1, (int)ceilf((Q.v - g.min_y + r) * g.grid_h_inv));
    if (nMinCellX >= ORBM_GRID_COLS || nMaxCellX < 0 || nMinCellY >= ORBM_GRID_ROWS || nMaxCellY < 0 || nMaxCellY < nMinCellY) return;
    const int minLevel = Q.min_level, maxLevel = Q.max_level;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);   // Frame.cc:810
    // the level filter of Frame.cc:826-832 (`octave < minLevel` -> skip; `maxLevel >= 0 && octave > maxLevel` -> skip, both only if bCheckLevels) as one
    // unsigned range test on the 16-bit octave: lo <= oct <= hi with lo / hi open where the reference does not test
    const int lo = bCheckLevels ? minLevel : -32768, hi = (bCheckLevels && maxLevel >= 0) ? maxLevel : 32767;
    const uint32_t span = (uint32_t)(hi - lo);               // hi < lo (an empty level window) wraps to a huge span only if lo > hi: handled below
    const bool noLevel = hi < lo;
    const bool stereoGate = (Q.flags & ORBM_Q_STEREO) && F.ur;
    // ONE loop over the window's entries: a grid column's cells [nMinCellY, nMaxCellY] are one contiguous CSR range (iy inner, insertion order
    // inside a cell), the columns follow each other (ix outer).  A lane's turn moves to its next column when the current one is used up AND takes an
    // entry if there is one, so a wave runs max over its lanes of (entries + empty columns) turns instead of a sum over columns of per-column maxima.
    int ix = nMinCellX - 1, p = 0, pe = 0;
    for (;;) {
        if (p >= pe) {
            if (++ix > nMaxCellX) break;
            p = F.gs[ix * ORBM_GRID_ROWS + nMinCellY]; pe = F.gs[ix * ORBM_GRID_ROWS + nMaxCellY + 1];
        }
        if (p >= pe) continue;
        const uint32_t e = F.ge[p];
        const float x = F.gx[p], y = F.gy[p];
        const int pp = p++;
        const int idx = (int)(e & 0xFFFFu), oct = (int)(int16_t)(e >> 16);
        if (idx >= F.n) continue;
        if (noLevel || (uint32_t)(oct - lo) > span) continue;
        const float distx = x - Q.u, disty = y - Q.v;
        if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;
        // returned by GetFeaturesInArea; now the candidate filters that do not depend on matches made during the call:
        if (F.occ0 && F.occ0[idx]) continue;
        if (stereoGate) {
            const float uR = F.ur[idx];
            if (uR > 0 && fabsf(Q.u_right - uR) > r) continue;
        }
        const uint32_t* dr = F.dsc + (size_t)idx * SBPF_DP;
        int dist = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) dist += __popc(qd.w[w] ^ dr[w]);
        sink(pp, idx, oct, dist);
    }
}
#ifndef SBPF_EXP
#define SBPF_EXP 0      // timing experiments only (tools/exp.py build): 1 no reading on behind the kept keys, 2 one round, 4 no sorted insertion, 8 no window walk
#endif
#define SBPF_T 1024
#ifndef SBPF_WPE
#define SBPF_WPE 8         // waves per SIMD the register allocation aims at: 8 = two workgroups per CU (64 VGPRs)
#endif
#define SBPF_SD 8          // smallest keys kept per query (the benchmark's lists hold 3.4 entries on average, 16 at most)
#define SBPF_ROW 128       // keys kept behind them in the query's workspace row (k_sbp_frame's rows are SBPF_ROW words; the fallback kernels lay their own 66-word rows over them)
                           // (the th = 15 local-map search after a relocalisation holds up to ~90 candidates per query; a list read past list + row hands the frame to the one-wave walk)
#define SBPF_KEY(dist, p, oct) (((uint32_t)(dist) << 22) | ((uint32_t)(p) << 6) | ((uint32_t)(oct) & 0x3Fu))
#define SBPF_KEY_P(k) (((k) >> 6) & 0xFFFFu)
// a query's kept keys: in registers (one query per thread) or, for the queries beyond the workgroup's threads, in LDS (key-major: lanes = consecutive queries)
struct SbpfRegList {
    uint32_t e[SBPF_SD];
    __device__ __forceinline__ uint32_t get(const int j) const { return e[j]; }
    __device__ __forceinline__ void set(const int j, const uint32_t v) { e[j] = v; }
};
struct SbpfLdsList {
    uint32_t* base; int stride;
    __device__ __forceinline__ uint32_t get(const int j) const { return base[j * stride]; }
    __device__ __forceinline__ void set(const int j, const uint32_t v) { base[j * stride] = v; }
};
static __device__ __forceinline__ uint32_t sbpf_umed3(const uint32_t a, const uint32_t b, const uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIP_EMULATED)
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return max(min(a, b), min(max(a, b), c));
#endif
}
// phase (1) for one query: walk, keep the SBPF_SD smallest keys sorted, spill the rest to the query's workspace row.  -> count | obs << 30
template <class L>
static __device__ __forceinline__ int sbpf_collect(const SbpfFrame& F, const SbpArgs& A, const int b, const int q, const orbm_query& Q, L& lst) {
    const Desc qd = load_desc(A.qdesc + ((size_t)b * A.cap_q + q) * 32);
    uint32_t* row = A.work + ((size_t)b * A.cap_q + q) * SBPF_ROW;
    int count = 0;
    sbpf_walk(F, A.prm.grid, Q, qd, [&](const int p, const int, const int oct, const int dist) {
        const uint32_t ne = SBPF_KEY(dist, p, oct);
        if (SBPF_EXP & 4) { lst.set(0, ne); count++; return; }
        // the list is ascending with 0xFFFFFFFF in its empty slots (every key is smaller): inserting one key is e'[j] = median(e[j - 1], key, e[j]) —
        // min(max(e[j - 1], key), e[j]) — for every slot at once, one v_med3_u32 each; what leaves at the end is the larger of the last slot and the key
        if (count >= SBPF_SD && count - SBPF_SD < SBPF_ROW) row[count - SBPF_SD] = max(lst.get(SBPF_SD - 1), ne);   // (unordered: the row is a set)
#pragma unroll
        for (int j = SBPF_SD - 1; j > 0; j--) lst.set(j, sbpf_umed3(lst.get(j - 1), ne, lst.get(j)));
        lst.set(0, min(lst.get(0), ne));
        count++;
    });
    return count | ((Q.flags & ORBM_Q_HAS_OBS) ? 1 << 30 : 0);
}
// wave-wide minimum by DPP steps (register only; the result forms in lane 63): the cooperative row scans below take two per query
static __device__ __forceinline__ uint32_t sbpf_wave_min(uint32_t x) {
#define SBPF_WMIN(ctrl, rows) x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, ctrl, rows, 0xF, false))
    SBPF_WMIN(0x111, 0xF); SBPF_WMIN(0x112, 0xF); SBPF_WMIN(0x114, 0xF); SBPF_WMIN(0x118, 0xF);   // row_shr:1, 2, 4, 8
    SBPF_WMIN(0x142, 0xA); SBPF_WMIN(0x143, 0xC);                                                 // row_bcast:15 -> rows 1, 3; row_bcast:31 -> rows 2, 3
#undef SBPF_WMIN
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
// phase (2) for one round: every lane's decision against the blocked set `cur` of this round.  -> accepted CSR position or -1.
// Called by ALL 64 lanes of a wave (cnt = 0 for a lane without a query or without candidates): the lanes whose kept keys are (nearly) all claimed have
// their workspace rows read by the whole wave, one query after the other — 64 keys per coalesced load and 64 blocked tests at once, two wave minima —
// instead of each walking its row alone through a chain of dependent global loads (wide windows: 0.46 instead of 3.1 ms per 512 frames of the
// th = 15 local-map search, where a tenth of the queries reads on every round).
template <class L>
static __device__ __forceinline__ int sbpf_decide(const SbpArgs& A, const int b, const int q, const int cnt, const L& lst,
                                                 const uint32_t* cur, const uint32_t tagCur, int* overflow) {
    const int mode = A.prm.mode, th = A.prm.th_dist;
    const int lane = threadIdx.x & 63;
    auto blocked_for = [&](const uint32_t p, const int qq) { const uint32_t v = cur[p]; return (v >> 16) == tagCur && (int)(v & 0xFFFFu) < qq; };   // tagCur = 0x10000 in round 0: nobody
    uint32_t e[SBPF_SD];
    uint32_t ub = 0;
#pragma unroll
    for (int j = 0; j < SBPF_SD; j++) e[j] = j < cnt ? lst.get(j) : 0u;
#pragma unroll
    for (int j = 0; j < SBPF_SD; j++)
        if (j < cnt && !blocked_for(SBPF_KEY_P(e[j]), q)) ub |= 1u << j;
    uint32_t eb1 = 0, eb2 = 0;
    bool have1 = ub != 0u;
    const uint32_t ub2 = ub & (ub - 1u);
    bool have2 = ub2 != 0u;
    const int j1 = have1 ? __ffs((int)ub) - 1 : -1, j2 = have2 ? __ffs((int)ub2) - 1 : -1;
#pragma unroll
    for (int j = 0; j < SBPF_SD; j++) { eb1 = j == j1 ? e[j] : eb1; eb2 = j == j2 ? e[j] : eb2; }
    const bool wantSecond = mode == ORBM_MODE_LOCAL_MAP;
    const bool need = cnt > SBPF_SD && (wantSecond ? !have2 : !have1) && !(SBPF_EXP & 1);
    // (a) the kept keys are (nearly) all claimed and the row holds the rest of the list: every key there is larger than the kept ones, so the two
    //     smallest unblocked keys of the row (an unordered set) continue the list
    unsigned long long todo = __ballot(need && cnt <= SBPF_SD + SBPF_ROW);
    while (todo) {                                       // wave-uniform; rare in the tracking searches
        const int ls = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const int qL = __builtin_amdgcn_readlane(q, ls), nrow = __builtin_amdgcn_readlane(cnt, ls) - SBPF_SD;
        const uint32_t* row = A.work + ((size_t)b * A.cap_q + qL) * SBPF_ROW;
        uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
        for (int k = lane; k < nrow; k += 64) {          // (at most two turns)
            const uint32_t key = row[k];
            if (blocked_for(SBPF_KEY_P(key), qL)) continue;
            if (key < m1) { m2 = m1; m1 = key; } else if (key < m2) m2 = key;
        }
        const uint32_t M1 = sbpf_wave_min(m1);
        const uint32_t M2 = sbpf_wave_min((m1 == M1) ? m2 : m1);   // keys are unique: one lane held M1
        if (lane == ls) {
            if (!have1) { have1 = M1 != 0xFFFFFFFFu; eb1 = M1; have2 = M2 != 0xFFFFFFFFu; eb2 = M2; }
            else { have2 = M1 != 0xFFFFFFFFu; eb2 = M1; }
        }
    }
    // (b) more candidates than list + row hold (> 136 in one window: no search of the path comes near) and this round reads past them: the frame is
    //     handed to the one-wave walk, which works from the window itself.  (Walking the window again right here is exact too, but inlined into
    //     the round loop it pushed the kept keys into scratch — eight reloads per query and round, 0.091 -> 0.105 ms per 512 benchmark frames.)
    if (need && cnt > SBPF_SD + SBPF_ROW) *overflow = 1;
    if (!have1) return -1;
    const int bestDist = (int)(eb1 >> 22);
    if (bestDist > th) return -1;
    if (wantSecond) {                                    // ORBmatcher.cc:160-178
        const int bestLevel = (int)(eb1 & 0x3Fu);
        const int bestDist2 = have2 ? (int)(eb2 >> 22) : 256;
        const int bestLevel2 = have2 ? (int)(eb2 & 0x3Fu) : -1;
        if (bestLevel == bestLevel2 && (float)bestDist > A.prm.nn_ratio * (float)bestDist2) return -1;
    }
    return (int)SBPF_KEY_P(eb1);                         // ORBmatcher.cc:2372
}

template <bool TAIL>       // queries beyond SBPF_T (cap_q <= 2 * SBPF_T): lists in LDS
static __global__ __launch_bounds__(SBPF_T, SBPF_WPE) void k_sbp_frame(SbpArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(A.nkp[(size_t)b * A.cstride], A.cap_k);
    const int nq = min(A.nq[b], A.cap_q);
    const int capk4 = (A.cap_k + 3) & ~3;
    int* hist = (int*)orb_smem;                          // [32]
    int* ctl = hist + 32;                                // [8]  0-2 maxima, 3 culled, 4 "serial", 5 accepters
    int* chg = ctl + 8;                                  // [8]  "something changed" of round r in slot r % 3; "a list ran out" in slot 4 + r % 3
    uint32_t* blk0 = (uint32_t*)(chg + 8);               // [capk4] x 2, by CSR position
    uint32_t* blk1 = blk0 + capk4;
    int* km = (int*)(blk1 + capk4);                      // [capk4] by key point index: last accepter / -1 / -2
    float* gx = (float*)(km + capk4);                    // [capk4] x of the key point at CSR position p
    float* gy = gx + capk4;                              // [capk4] y
    uint32_t* ge = (uint32_t*)(gy + capk4);              // [capk4] index | octave << 16 of the key point at CSR position p
    uint32_t* dsc = ge + capk4;                          // [capk4][SBPF_DP] descriptors by key point index
    uint16_t* gs = (uint16_t*)(dsc + (size_t)capk4 * SBPF_DP);   // [GRID_CELLS + 2] CSR heads
    const int tq = TAIL ? A.cap_q - SBPF_T : 0;          // queries with an LDS list
    uint32_t* tl = (uint32_t*)(gs + GRID_CELLS + 2);     // [SBPF_SD][tq]
    const bool ori = A.prm.mode == ORBM_MODE_BEST_ONLY && A.prm.check_orientation;
    const orb_keypoint* kps = A.kps + (size_t)b * A.cap_k;
    const orbm_query* queries = A.queries + (size_t)b * A.cap_q;
    // ---- (1a) stage the frame
    if (tid < 32) hist[tid] = 0;
    if (tid < 8) ctl[tid] = 0;
    if (tid < 8) chg[tid] = 0;
    for (int i = tid; i < A.cap_k; i += SBPF_T) { blk0[i] = 0xFFFFFFFFu; blk1[i] = 0xFFFFFFFFu; km[i] = -1; }
    {
        // the grid in CSR order: position p -> (index, octave, x, y) side by side, so that a window entry is ONE round of independent LDS reads
        // (grid index -> record was a dependent pair); key points outside the grid appear in no cell and are not staged
        const int32_t* gsrc = A.grid_start + (size_t)b * (A.cells + 1);
        const int32_t* isrc = A.grid_idx + (size_t)b * A.cap_k;
        const int tot = min(gsrc[GRID_CELLS], n);
        for (int p = tid; p < tot; p += SBPF_T) {
            const int idx = isrc[p];
            if (idx >= 0 && idx < n) { const orb_keypoint kp = kps[idx]; gx[p] = kp.x; gy[p] = kp.y; ge[p] = (uint32_t)idx | ((uint32_t)(kp.octave & 0xFFFF) << 16); }
            else { gx[p] = 0.f; gy[p] = 0.f; ge[p] = 0xFFFFu; }     // never produced by k_grid_build; 0xFFFF >= n: skipped
        }
        for (int i = tid; i <= GRID_CELLS; i += SBPF_T) gs[i] = (uint16_t)min(gsrc[i], tot);
        const uint4* dsrc = (const uint4*)(A.desc + (size_t)b * A.cap_k * 32);
        for (int i = tid; i < 2 * n; i += SBPF_T) {
            const uint4 v = dsrc[i];
            uint32_t* d = dsc + (size_t)(i >> 1) * SBPF_DP + (i & 1) * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    const SbpfFrame F{gs, ge, gx, gy, dsc, A.occupied0 ? A.occupied0 + (size_t)b * A.cap_k : nullptr, A.u_right ? A.u_right + (size_t)b * A.cap_k : nullptr, n};
    // ---- (1b) every query walks its window
    SbpfRegList r0;
    SbpfLdsList r1{tl + tid, tq};
#pragma unroll
    for (int j = 0; j < SBPF_SD; j++) r0.e[j] = 0xFFFFFFFFu;
    if (TAIL && tid < tq) {
#pragma unroll
        for (int j = 0; j < SBPF_SD; j++) r1.set(j, 0xFFFFFFFFu);
    }
    int cntw0 = 0, cntw1 = 0, d0 = -1, d1 = -1;          // candidate count | obs << 30, decision (CSR position or -1), per slot
    bool bad = false;
    const int q1 = tid + SBPF_T;
    const bool has1 = TAIL && q1 < nq;                   // (tid < tq follows: nq <= cap_q)
    __syncthreads();
    if (tid < nq) {
        const orbm_query Q = queries[tid];
        if (Q.flags & (ORBM_Q_TWIN | ORBM_Q_RIGHT)) bad = true;
        if ((Q.flags & ORBM_Q_VALID) && !(SBPF_EXP & 8)) cntw0 = sbpf_collect(F, A, b, tid, Q, r0);
    }
    if (has1) {
        const orbm_query Q = queries[q1];
        if (Q.flags & (ORBM_Q_TWIN | ORBM_Q_RIGHT)) bad = true;
        if ((Q.flags & ORBM_Q_VALID) && !(SBPF_EXP & 8)) cntw1 = sbpf_collect(F, A, b, q1, Q, r1);
    }
    if (bad) ctl[4] = 1;
    // The workspace rows (global memory) are written by ONE lane each (sbpf_collect) and read back in the rounds below by the WHOLE wave
    // (sbpf_decide walks a row with all lanes): writes by one work-item, reads by others of the same workgroup.  What orders them is a release of
    // the writer and an acquire of the readers at workgroup scope around the barrier — stated explicitly, so that the pairing does not rest on
    // the barrier's implied fence or on how this compute unit's L1 happens to be shared (threadgroup-split mode would break an implicit one silently).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- (2) fixed-point rounds
    const int maxRounds = nq + 2;
    bool ran_out = false;
    for (int r = 0; r < maxRounds && !ctl[4]; r++) {
        const uint32_t* cur = (r & 1) ? blk1 : blk0;     // written in round r - 1 with tag 0x10000 - r
        uint32_t* nxt = (r & 1) ? blk0 : blk1;
        const uint32_t tagCur = 0x10000u - (uint32_t)r, tagNxt = (0xFFFFu - (uint32_t)r) << 16;
        bool changed = false;
        if (__ballot(cntw0 & 0x07FFFFFF)) {   // (wave-uniform: the decision function reads workspace rows with the whole wave)
            const int nd = sbpf_decide(A, b, tid, cntw0 & 0x07FFFFFF, r0, cur, tagCur, &chg[4 + r % 3]);
            if (nd != d0) { changed = true; d0 = nd; }
            if (nd >= 0 && ((cntw0 >> 30) & 1)) atomicMin(&nxt[nd], tagNxt | (uint32_t)tid);
        }
        if (TAIL && __ballot(cntw1 & 0x07FFFFFF)) {
            const int nd = sbpf_decide(A, b, q1, cntw1 & 0x07FFFFFF, r1, cur, tagCur, &chg[4 + r % 3]);
            if (nd != d1) { changed = true; d1 = nd; }
            if (nd >= 0 && ((cntw1 >> 30) & 1)) atomicMin(&nxt[nd], tagNxt | (uint32_t)q1);
        }
        if (changed) chg[r % 3] = 1;
        if (tid == 0) chg[(r + 1) % 3] = 0;
        __syncthreads();
        if (chg[4 + r % 3]) { ran_out = true; break; }   // workgroup-uniform, like the next line (the slots of round r are not written again before r + 3)
        if (!chg[r % 3] || (SBPF_EXP & 2)) break;
    }
    if (ctl[4] || ran_out) {                             // workgroup-uniform (read after a barrier in every path)
        if (tid == 0) A.serial_flag[b] = 1;
        return;
    }
    if (tid == 0) A.serial_flag[b] = 0;
    // ---- (3) what the serial walk leaves behind, from the fixed point (decisions: CSR position -> key point index)
    d0 = d0 >= 0 ? (int)(ge[d0] & 0xFFFFu) : -1;
    d1 = (TAIL && d1 >= 0) ? (int)(ge[d1] & 0xFFFFu) : -1;
    int acc = 0;
    if (d0 >= 0) { atomicMax(&km[d0], tid); acc++; }
    if (d1 >= 0) { atomicMax(&km[d1], q1); acc++; }
    {   // accepters of the frame: a wave sum by ballots of the bits (acc <= 2), one LDS atomic per wave
        const int wsum = __popcll(__ballot(acc & 1)) + 2 * __popcll(__ballot(acc >> 1));
        if ((tid & 63) == 0 && wsum) atomicAdd(&ctl[5], wsum);
    }
    __syncthreads();
    if (ori) {
        // rotation histogram (ORBmatcher.cc:2387-2395: factor = 1/HISTO_LENGTH quirk, C round()) of every accepted match
        auto bin_of = [&](const int q, const int idx) {
            float rot = queries[q].angle - kps[idx].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * (1.0f / ORBM_HISTO_LENGTH));
            if (bin == ORBM_HISTO_LENGTH) bin = 0;
            bin = max(0, min(bin, 31));
            atomicAdd(&hist[bin], 1);
            return bin;
        };
        const int bin0 = d0 >= 0 ? bin_of(tid, d0) : -1, bin1 = d1 >= 0 ? bin_of(q1, d1) : -1;
        __syncthreads();
        if (tid == 0) {  // ComputeThreeMaxima, ORBmatcher.cc:2654-2695
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < ORBM_HISTO_LENGTH; i++) {
                const int c = hist[i];
                if (c > max1) { max3 = max2; max2 = max1; max1 = c; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (c > max2) { max3 = max2; max2 = c; ind3 = ind2; ind2 = i; }
                else if (c > max3) { max3 = c; ind3 = i; }
            }
            if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
            ctl[0] = ind1; ctl[1] = ind2; ctl[2] = ind3;
        }
        __syncthreads();
        const int ind1 = ctl[0], ind2 = ctl[1], ind3 = ctl[2];
        // CurrentFrame.mvpMapPoints[...] = NULL, :2499 (-2: claimed during the call, then culled)
        if (d0 >= 0 && bin0 != ind1 && bin0 != ind2 && bin0 != ind3) { km[d0] = -2; atomicAdd(&ctl[3], 1); }
        if (d1 >= 0 && bin1 != ind1 && bin1 != ind2 && bin1 != ind3) { km[d1] = -2; atomicAdd(&ctl[3], 1); }
        __syncthreads();
    }
    int32_t* q_match = A.q_match + (size_t)b * A.cap_q;
    int32_t* kp_match = A.kp_match + (size_t)b * A.cap_k;
    // a query's match is reported only while its key point still holds it
    if (tid < A.cap_q) q_match[tid] = (d0 >= 0 && km[d0] == tid) ? d0 : -1;
    if (TAIL && q1 < A.cap_q) q_match[q1] = (d1 >= 0 && km[d1] == q1) ? d1 : -1;
    for (int i = tid; i < A.cap_k; i += SBPF_T) kp_match[i] = km[i];
    if (tid == 0) A.nmatches[b] = ctl[5] - ctl[3];
}

// The candidate lists of the frames k_sbp_frame flagged, for k_sbp_resolve: one workgroup per frame walks the frame's query blocks through the
// body of k_sbp_candidates2 (rare frames: the launch is a count check per frame otherwise).
static __global__ __launch_bounds__(256) void k_sbp_candidates_flagged(SbpArgs A) {
    const int b = blockIdx.x;
    if (A.serial_flag[b] == 0) return;
    const int nblk = (min(A.nq[b], A.cap_q) + 7) / 8;
    for (int qb = 0; qb < nblk; qb++) {
        sbp_candidates2_block(A, b, qb);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                 // the wave's LDS lists are rewritten by its next block
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ============================================================================================================
// M6  SearchByBoW(KeyFrame*, Frame&)  ORBmatcher.cc:323-587   and   SearchByBoW(KeyFrame*, KeyFrame*)  ORBmatcher.cc:984-1124
// ============================================================================================================
// KK = the key-frame / key-frame overload: the inner side has its own validity flags (`!pMP2 || pMP2->isBad()`, :1049-1053), the blocking
// state is vbMatched2, the acceptance is `bestDist1 < TH_LOW` (strict, :1072) and the result is indexed by the OUTER feature
// (vpMatches12[idx1], :1076); the rotation histogram therefore culls by idx1 (:1088, :1117).  fbin[] (LDS, per inner feature) is
// vbMatched2 and carries the match's bin; each inner feature is matched at most once, so the cull finds idx1's bin through match12[idx1].
struct BowArgs {
    orbm_bow_side kf, f;
    const uint8_t* kf_valid;
    const uint8_t* f_valid;   // KK only
    float nn_ratio; int check_orientation;
    int32_t* f_match; int32_t* nmatches;   // KK: f_match = match12 [batch][kf.cap_f]
};

template <bool KK>
static __global__ __launch_bounds__(256) void k_bow(BowArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* hist = (int*)orb_smem;                    // [32]
    int* ctl = hist + 32;                          // [8]
    int* pairF = ctl + 8;                          // [kf.cap_nodes]
    int8_t* fbin = (int8_t*)(pairF + A.kf.cap_nodes);  // [f.cap_f] rotation bin of an accepted match, -1 otherwise
    const int nkn = min(A.kf.n_nodes[b], A.kf.cap_nodes), nfn = min(A.f.n_nodes[b], A.f.cap_nodes);
    const int32_t* kid = A.kf.node_id + (size_t)b * A.kf.cap_nodes;
    const int32_t* kst = A.kf.node_start + (size_t)b * (A.kf.cap_nodes + 1);
    const int32_t* kfe = A.kf.feat_idx + (size_t)b * A.kf.cap_f;
    const int32_t* fid = A.f.node_id + (size_t)b * A.f.cap_nodes;
    const int32_t* fst = A.f.node_start + (size_t)b * (A.f.cap_nodes + 1);
    const int32_t* ffe = A.f.feat_idx + (size_t)b * A.f.cap_f;
    const uint8_t* kdesc = A.kf.desc + (size_t)b * A.kf.cap_f * 32;
    const uint8_t* fdesc = A.f.desc + (size_t)b * A.f.cap_f * 32;
    const float* kang = A.kf.angle + (size_t)b * A.kf.cap_f;
    const float* fang = A.f.angle + (size_t)b * A.f.cap_f;
    const uint8_t* kvalid = A.kf_valid + (size_t)b * A.kf.cap_f;
    int32_t* f_match = A.f_match + (size_t)b * (KK ? A.kf.cap_f : A.f.cap_f);
    const uint8_t* fvalid = KK ? A.f_valid + (size_t)b * A.f.cap_f : nullptr;
    const int nleft = (!KK && A.f.n_left) ? A.f.n_left[b] : -1;
    if (tid < 32) hist[tid] = 0;
    if (tid < 8) ctl[tid] = 0;
    for (int i = tid; i < A.f.cap_f; i += 256) { if (!KK) f_match[i] = -1; fbin[i] = -1; }
    if (KK) for (int i = tid; i < A.kf.cap_f; i += 256) f_match[i] = -1;
    // the merge walk of the two sorted FeatureVectors (ORBmatcher.cc:343, 553-560) = intersection of the id lists
    for (int k = tid; k < nkn; k += 256) {
        const int key = kid[k];
        int lo = 0, hi = nfn;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (fid[m] < key) lo = m + 1; else hi = m; }
        pairF[k] = (lo < nfn && fid[lo] == key) ? lo : -1;
    }
    __syncthreads();
    int myMatches = 0;
    for (int k = wave; k < nkn; k += 4) {
        const int fn = pairF[k];
        if (fn < 0) continue;
        const int fs = fst[fn], fe = fst[fn + 1];
        for (int iKF = kst[k]; iKF < kst[k + 1]; iKF++) {
            const int realIdxKF = kfe[iKF];
            if (!kvalid[realIdxKF]) continue;
            const Desc dKF = load_desc(kdesc + (size_t)realIdxKF * 32);
            // per-lane best/second keys (dist << 20 | position in the node) for the left camera's features (all of them for a single
            // camera) and, on a fisheye rig, separately for the right camera's (realIdxF >= Nleft), ORBmatcher.cc:411-436
            uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu, r1 = 0xFFFFFFFFu, r2 = 0xFFFFFFFFu;
            int idx1 = -1, idxr = -1;
            for (int base = fs; base < fe; base += 64) {
                const int p = base + lane;
                if (p < fe) {
                    const int realIdxF = ffe[p];
                    if (fbin[realIdxF] == -1 && (!KK || fvalid[realIdxF])) {   // !vpMapPointMatches[realIdxF] / !vbMatched2[idx2] && pMP2 good  (fbin != -1 <=> matched during this call; LDS only on the serial chain)
                        const uint32_t key = ((uint32_t)hamming(dKF, load_desc(fdesc + (size_t)realIdxF * 32)) << 20) | (uint32_t)(p - fs);
                        if (nleft < 0 || realIdxF < nleft) {
                            if (key < k1) { k2 = k1; k1 = key; idx1 = realIdxF; }
                            else if (key < k2) { k2 = key; }
                        } else {
                            if (key < r1) { r2 = r1; r1 = key; idxr = realIdxF; }
                            else if (key < r2) { r2 = key; }
                        }
                    }
                }
            }
            const uint32_t m1 = wave_min_u32(k1);
            if (m1 == 0xFFFFFFFFu) continue;   // bestDist1 stays 256 > TH_LOW: neither camera is matched (:453)
            const bool iBest = k1 == m1;
            const uint32_t m2 = wave_min_u32(iBest ? k2 : k1);
            const int l1 = __ffsll((long long)__ballot(iBest)) - 1;
            const int bestIdxF = __shfl(idx1, l1);
            const int bestDist1 = (int)(m1 >> 20), bestDist2 = m2 == 0xFFFFFFFFu ? 256 : (int)(m2 >> 20);
            if (KK ? bestDist1 >= ORBM_TH_LOW : bestDist1 > ORBM_TH_LOW) continue;   // `<= TH_LOW` :453 vs `< TH_LOW` :1072
            const bool accL = (float)bestDist1 < A.nn_ratio * (float)bestDist2;   // :455
            int bestIdxFR = -1;
            if (nleft >= 0) {   // right camera: accepted whenever bestDist1R <= TH_LOW (the ratio test is short-circuited by `|| true`, :509)
                const uint32_t mr = wave_min_u32(r1);
                if (mr != 0xFFFFFFFFu && (int)(mr >> 20) <= ORBM_TH_LOW) {
                    const int lr = __ffsll((long long)__ballot(r1 == mr)) - 1;
                    bestIdxFR = __shfl(idxr, lr);
                }
            }
            if (accL || bestIdxFR >= 0) {
                myMatches += (accL ? 1 : 0) + (bestIdxFR >= 0 ? 1 : 0);
                if (lane == 0) {
                    for (int side = 0; side < 2; side++) {
                        const int idxF = side == 0 ? (accL ? bestIdxF : -1) : bestIdxFR;
                        if (idxF < 0) continue;
                        if (KK) f_match[realIdxKF] = idxF; else f_match[idxF] = realIdxKF;
                        int bin = 30;  // "accepted, orientation unchecked"
                        if (A.check_orientation) {
                            float rot = kang[realIdxKF] - fang[idxF];
                            if (rot < 0.0f) rot += 360.0f;
                            bin = (int)roundf(rot * (1.0f / ORBM_HISTO_LENGTH));
                            if (bin == ORBM_HISTO_LENGTH) bin = 0;
                            bin = max(0, min(bin, 29));
                            atomicAdd(&hist[bin], 1);
                        }
                        fbin[idxF] = (int8_t)bin;
                    }
                }
                __threadfence_block();
                __builtin_amdgcn_wave_barrier();  // lane 0's LDS write is program-ordered before the next feature's reads
            }
        }
    }
    if (lane == 0 && myMatches) atomicAdd(&ctl[4], myMatches);
    __syncthreads();
    if (A.check_orientation) {
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < ORBM_HISTO_LENGTH; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
            ctl[0] = ind1; ctl[1] = ind2; ctl[2] = ind3;
        }
        __syncthreads();
        const int ind1 = ctl[0], ind2 = ctl[1], ind3 = ctl[2];
        if (KK) {
            for (int i = tid; i < A.kf.cap_f; i += 256) {
                const int j = f_match[i];
                if (j >= 0) {
                    const int bin = fbin[j];
                    if (bin != ind1 && bin != ind2 && bin != ind3) { f_match[i] = -1; atomicAdd(&ctl[5], 1); }
                }
            }
        } else {
            for (int j = tid; j < A.f.cap_f; j += 256) {
                const int bin = fbin[j];
                if (bin >= 0 && bin != ind1 && bin != ind2 && bin != ind3) { f_match[j] = -1; atomicAdd(&ctl[5], 1); }
            }
        }
        __syncthreads();
    }
    if (tid == 0) A.nmatches[b] = ctl[4] - ctl[5];
}

// ============================================================================================================
// M12  SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse)   ORBmatcher.cc:1138-1428 (pinhole, one camera)
// ============================================================================================================
// The reference never sets vbMatched2, so every KF1 feature is independent: its match is the gate-passing candidate of the shared
// vocabulary node with the smallest distance <= TH_LOW, the LAST one among equals (the `dist>bestDist -> continue` update rule).
// One workgroup per key-frame pair, one wave per shared node, lanes over the node's KF2 features.
// KB8 = key frames with KannalaBrandt8 cameras (one camera, or a fisheye rig with mpCamera2): the gate is KannalaBrandt8::epipolarConstrain =
// TriangulateMatches(...) > 0.0001f (KannalaBrandt8.cpp:235-238, 334-400; rule R4) with the (R12, t12, camera) combination the reference picks
// per candidate from (bRight1, bRight2) (ORBmatcher.cc:1280-1315); bStereo1 / bStereo2 are false for such key frames (mvuRight is not set
// by the fisheye Frame constructor), the epipole test only applies without a second camera (:1269).
struct TriArgs {
    orbm_tri_side k1, k2;
    const orbm_tri_pair* pairs;
    const orbm_tri_kb8_pair* kb8; const int32_t* nleft1; const int32_t* nleft2;   // KB8 only
    int only_stereo, coarse, check_orientation;
    int32_t* match12; int32_t* nmatches;
};
#include "kb8_geom.inc"

template <bool KB8>
static __global__ __launch_bounds__(256) void k_tri(TriArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* hist = (int*)orb_smem;                    // [32]
    int* ctl = hist + 32;                          // [8]
    int* pair2 = ctl + 8;                          // [k1.cap_nodes]
    int8_t* bin1 = (int8_t*)(pair2 + A.k1.cap_nodes);  // [k1.cap_f] rotation bin of a matched KF1 feature, -1 otherwise
    const int nn1 = min(A.k1.n_nodes[b], A.k1.cap_nodes), nn2 = min(A.k2.n_nodes[b], A.k2.cap_nodes);
    const int32_t* id1 = A.k1.node_id + (size_t)b * A.k1.cap_nodes;
    const int32_t* st1 = A.k1.node_start + (size_t)b * (A.k1.cap_nodes + 1);
    const int32_t* fe1 = A.k1.feat_idx + (size_t)b * A.k1.cap_f;
    const int32_t* id2 = A.k2.node_id + (size_t)b * A.k2.cap_nodes;
    const int32_t* st2 = A.k2.node_start + (size_t)b * (A.k2.cap_nodes + 1);
    const int32_t* fe2 = A.k2.feat_idx + (size_t)b * A.k2.cap_f;
    const orb_keypoint* kp1s = A.k1.kps + (size_t)b * A.k1.cap_f;
    const orb_keypoint* kp2s = A.k2.kps + (size_t)b * A.k2.cap_f;
    const uint8_t* d1s = A.k1.desc + (size_t)b * A.k1.cap_f * 32;
    const uint8_t* d2s = A.k2.desc + (size_t)b * A.k2.cap_f * 32;
    const float* ur1 = A.k1.u_right ? A.k1.u_right + (size_t)b * A.k1.cap_f : nullptr;
    const float* ur2 = A.k2.u_right ? A.k2.u_right + (size_t)b * A.k2.cap_f : nullptr;
    const uint8_t* mp1 = A.k1.has_mp + (size_t)b * A.k1.cap_f;
    const uint8_t* mp2 = A.k2.has_mp + (size_t)b * A.k2.cap_f;
    const orbm_tri_pair* const Pp = KB8 ? nullptr : A.pairs + b;          // exactly one of the two pair records exists
    const orbm_tri_kb8_pair* const Qp = KB8 ? A.kb8 + b : nullptr;
    const bool rig = KB8 && Qp->n_cams == 2;
    const int nl1 = rig ? A.nleft1[b] : -1, nl2 = rig ? A.nleft2[b] : -1;
    int32_t* match12 = A.match12 + (size_t)b * A.k1.cap_f;
    if (tid < 32) hist[tid] = 0;
    if (tid < 8) ctl[tid] = 0;
    for (int i = tid; i < A.k1.cap_f; i += 256) { match12[i] = -1; bin1[i] = -1; }
    for (int k = tid; k < nn1; k += 256) {   // merge walk of the two sorted FeatureVectors = intersection of the id lists
        const int key = id1[k];
        int lo = 0, hi = nn2;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (id2[m] < key) lo = m + 1; else hi = m; }
        pair2[k] = (lo < nn2 && id2[lo] == key) ? lo : -1;
    }
    __syncthreads();
    float F[9];
#pragma unroll
    for (int i = 0; i < 9; i++) F[i] = KB8 ? 0.f : Pp->F12[i];
    const float epx = KB8 ? Qp->ep[0] : Pp->ep[0], epy = KB8 ? Qp->ep[1] : Pp->ep[1];
    const float* sf2 = KB8 ? Qp->scale_factors_2 : Pp->scale_factors_2;
    const float* sg2 = KB8 ? Qp->level_sigma2_2 : Pp->level_sigma2_2;
    int myMatches = 0;
    for (int k = wave; k < nn1; k += 4) {
        const int n2 = pair2[k];
        if (n2 < 0) continue;
        const int s2 = st2[n2], e2 = st2[n2 + 1];
        for (int i1 = st1[k]; i1 < st1[k + 1]; i1++) {
            const int idx1 = fe1[i1];
            if (mp1[idx1]) continue;
            const bool bStereo1 = !KB8 && ur1 && ur1[idx1] >= 0;
            if (A.only_stereo && !bStereo1) continue;
            const orb_keypoint kp1 = kp1s[idx1];
            const int bRight1 = (nl1 >= 0 && idx1 >= nl1) ? 1 : 0;
            const Desc d1 = load_desc(d1s + (size_t)idx1 * 32);
            // Pinhole.cpp:162-165 epipolar line of kp1 in image 2
            const float la = kp1.x * F[0] + kp1.y * F[3] + F[6];
            const float lb = kp1.x * F[1] + kp1.y * F[4] + F[7];
            const float lc = kp1.x * F[2] + kp1.y * F[5] + F[8];
            const float den = la * la + lb * lb;
            uint32_t k1 = 0xFFFFFFFFu;
            int best2 = -1;
            for (int base = s2; base < e2; base += 64) {
                const int p = base + lane;
                if (p >= e2) continue;
                const int idx2 = fe2[p];
                if (mp2[idx2]) continue;
                const bool bStereo2 = !KB8 && ur2 && ur2[idx2] >= 0;
                if (A.only_stereo && !bStereo2) continue;
                const int dist = hamming(d1, load_desc(d2s + (size_t)idx2 * 32));
                if (dist > ORBM_TH_LOW) continue;
                const orb_keypoint kp2 = kp2s[idx2];
                if (!bStereo1 && !bStereo2 && !rig) {   // :1269-1277 too close to the epipole (not with a second camera)
                    const float distex = epx - kp2.x, distey = epy - kp2.y;
                    if (distex * distex + distey * distey < 100 * sf2[kp2.octave & 15]) continue;
                }
                bool ok = A.coarse != 0;
                if (KB8) {
                    if (!ok) {
                        const int bRight2 = (nl2 >= 0 && idx2 >= nl2) ? 1 : 0;
                        const int combo = bRight1 * 2 + bRight2;
                        float x3D[3];
                        ok = kb8_triangulate_matches(Qp->k1[bRight1], Qp->k2[bRight2], Qp->R12[combo], Qp->t12[combo], kp1, kp2, Qp->level_sigma2_1[kp1.octave & 15],
                                                     sg2[kp2.octave & 15], x3D) > 0.0001f;
                    }
                } else if (!ok && den != 0) {
                    const float num = la * kp2.x + lb * kp2.y + lc;
                    const float dsqr = num * num / den;
                    ok = (double)dsqr < 3.84 * (double)sg2[kp2.octave & 15];
                }
                if (!ok) continue;
                const uint32_t key = ((uint32_t)dist << 20) | (uint32_t)(0xFFFFF - ((p - s2) & 0xFFFFF));
                if (key < k1) { k1 = key; best2 = idx2; }
            }
            const uint32_t m1 = wave_min_u32(k1);
            if (m1 == 0xFFFFFFFFu) continue;
            const int l1 = __ffsll((long long)__ballot(k1 == m1)) - 1;
            const int bestIdx2 = __shfl(best2, l1);
            myMatches++;
            if (lane == 0) {
                match12[idx1] = bestIdx2;
                int bin = 30;
                if (A.check_orientation) {   // :1344-1353, factor = 1/HISTO_LENGTH quirk
                    float rot = kp1.angle - kp2s[bestIdx2].angle;
                    if (rot < 0.0f) rot += 360.0f;
                    bin = (int)roundf(rot * (1.0f / ORBM_HISTO_LENGTH));
                    if (bin == ORBM_HISTO_LENGTH) bin = 0;
                    bin = max(0, min(bin, 29));
                    atomicAdd(&hist[bin], 1);
                }
                bin1[idx1] = (int8_t)bin;
            }
        }
    }
    if (lane == 0 && myMatches) atomicAdd(&ctl[4], myMatches);
    __threadfence_block();
    __syncthreads();
    if (A.check_orientation) {
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < ORBM_HISTO_LENGTH; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
            ctl[0] = ind1; ctl[1] = ind2; ctl[2] = ind3;
        }
        __syncthreads();
        const int ind1 = ctl[0], ind2 = ctl[1], ind3 = ctl[2];
        for (int j = tid; j < A.k1.cap_f; j += 256) {
            const int bin = bin1[j];
            if (bin >= 0 && bin != ind1 && bin != ind2 && bin != ind3) { match12[j] = -1; atomicAdd(&ctl[5], 1); }
        }
        __syncthreads();
    }
    if (tid == 0) A.nmatches[b] = ctl[4] - ctl[5];
}

// ============================================================================================================
// C ABI
// ============================================================================================================
static int launch_status() { return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP; }

// ---- optional per-kernel device timing of the projection search (bench.py's roofline legs): HIP events recorded on the launch stream
// around k_grid_build / k_sbp_candidates2 / k_sbp_resolve while enabled.  Process-wide and not re-entrant: a measurement facility only.
// per calling thread: Tracking, LocalMapping and LoopClosing each drive their own matcher calls on their own streams; a thread that enables the
// timing gets events of its own and reads back its own last call, whatever the others do
// The events belong to the thread AND to the device that was current when they were made: they are destroyed with the thread (short-lived
// worker threads do not leak them) and made again when the thread has moved to another device.
struct MatcherTiming {
    bool on = false, made = false, have_grid = false, have_sbp = false;
    int device = -1;
    hipEvent_t ev[5];
    void drop() {
        if (made) for (auto& e : ev) (void)hipEventDestroy(e);
        made = have_grid = have_sbp = false;
    }
    ~MatcherTiming() { drop(); }
};
static thread_local MatcherTiming g_mt;
static bool mt_ready() {
    if (!g_mt.on) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (g_mt.made && g_mt.device != dev) g_mt.drop();
    if (!g_mt.made) {
        int n = 0;
        for (auto& e : g_mt.ev) { if (hipEventCreate(&e) != hipSuccess) { for (int i = 0; i < n; i++) (void)hipEventDestroy(g_mt.ev[i]); return false; } n++; }
        g_mt.made = true; g_mt.device = dev;
    }
    return true;
}
extern "C" int orbm_enable_timing(int on) { g_mt.on = on != 0; g_mt.have_grid = g_mt.have_sbp = false; return ORB_OK; }
extern "C" int orbm_last_timing(float* ms3) {
    if (!ms3) return ORB_E_INVALID;
    ms3[0] = ms3[1] = ms3[2] = 0.f;
    if (!g_mt.made) return ORB_OK;
    if (g_mt.have_grid) {
        if (hipEventSynchronize(g_mt.ev[1]) != hipSuccess || hipEventElapsedTime(&ms3[0], g_mt.ev[0], g_mt.ev[1]) != hipSuccess) return ORB_E_HIP;
    }
    if (g_mt.have_sbp) {
        if (hipEventSynchronize(g_mt.ev[4]) != hipSuccess || hipEventElapsedTime(&ms3[1], g_mt.ev[2], g_mt.ev[3]) != hipSuccess ||
            hipEventElapsedTime(&ms3[2], g_mt.ev[3], g_mt.ev[4]) != hipSuccess) return ORB_E_HIP;
    }
    return ORB_OK;
}

extern "C" int orbm_hamming(const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int batch, uint16_t* d_out, void* stream) {
    if (!d_q || !d_t || !d_out || nq < 1 || nt < 1 || batch < 1) return ORB_E_INVALID;
    hipLaunchKernelGGL(k_hamming, dim3((nt + 255) / 256, (nq + 15) / 16, batch), dim3(256), 512, (hipStream_t)stream, d_q, nq, d_t, nt, d_out);
    return launch_status();
}

extern "C" int orbm_knn2(const uint8_t* d_q, const int32_t* d_nq, int cap_q, const uint8_t* d_t, const int32_t* d_nt, int cap_t,
                         int count_stride, int batch, int32_t* d_out_idx, int32_t* d_out_dist, void* stream) {
    if (!d_q || !d_t || !d_nq || !d_nt || !d_out_idx || !d_out_dist || cap_q < 1 || cap_t < 1 || batch < 1 || count_stride < 1) return ORB_E_INVALID;
    hipLaunchKernelGGL(k_knn2, dim3((cap_q + 255) / 256, batch), dim3(256), 64 * 32, (hipStream_t)stream, d_q, d_nq, cap_q, d_t, d_nt, cap_t,
                       count_stride, d_out_idx, d_out_dist);
    return launch_status();
}

extern "C" int orbm_grid_build(const orb_keypoint* d_kps, const int32_t* d_nkp, int count_stride, int cap_k, int batch,
                               const orbm_grid_params* gp, int32_t* d_grid_start, int32_t* d_grid_idx, void* stream) {
    if (!d_kps || !d_nkp || !gp || !d_grid_start || !d_grid_idx || cap_k < 1 || batch < 1 || count_stride < 1) return ORB_E_INVALID;
    const bool timed = mt_ready();
    if (timed) (void)hipEventRecord(g_mt.ev[0], (hipStream_t)stream);
    hipLaunchKernelGGL(k_grid_build, dim3(batch), dim3(256), (2 * GRID_CELLS + 256) * 4, (hipStream_t)stream, d_kps, d_nkp, count_stride, cap_k,
                       *gp, d_grid_start, d_grid_idx, (const int32_t*)nullptr, GRID_CELLS);
    if (timed) { (void)hipEventRecord(g_mt.ev[1], (hipStream_t)stream); g_mt.have_grid = true; }
    return launch_status();
}

extern "C" int orbm_grid_build_rig(const orb_keypoint* d_kps, const int32_t* d_nkp, const int32_t* d_nleft, int count_stride, int cap_k, int batch,
                                   const orbm_grid_params* gp, int32_t* d_grid_start, int32_t* d_grid_idx, void* stream) {
    if (!d_kps || !d_nkp || !d_nleft || !gp || !d_grid_start || !d_grid_idx || cap_k < 1 || batch < 1 || count_stride < 1) return ORB_E_INVALID;
    hipLaunchKernelGGL(k_grid_build, dim3(batch), dim3(256), (4 * GRID_CELLS + 256) * 4, (hipStream_t)stream, d_kps, d_nkp, count_stride, cap_k,
                       *gp, d_grid_start, d_grid_idx, d_nleft, 2 * GRID_CELLS);
    return launch_status();
}

extern "C" int orbm_undistort_and_grid_build(const orb_keypoint* d_kps, const int32_t* d_nkp, int count_stride, int cap_k, int batch, const orbf_camera* cam,
                                             const orbm_grid_params* gp, orb_keypoint* d_kps_un, int32_t* d_grid_start, int32_t* d_grid_idx, void* stream) {
    if (!d_kps || !d_nkp || !cam || !gp || !d_kps_un || !d_grid_start || !d_grid_idx || cap_k < 1 || batch < 0 || count_stride < 1) return ORB_E_INVALID;
    if (batch == 0) return ORB_OK;
    if (cap_k > 4 * UG_T) {   // beyond the fused kernel's four key points per thread: the two separate steps
        const int rc = orbf_undistort_keypoints(d_kps, d_nkp, count_stride, cap_k, batch, cam, d_kps_un, stream);
        return rc != ORB_OK ? rc : orbm_grid_build(d_kps_un, d_nkp, count_stride, cap_k, batch, gp, d_grid_start, d_grid_idx, stream);
    }
    const bool timed = mt_ready();
    if (timed) (void)hipEventRecord(g_mt.ev[0], (hipStream_t)stream);
    hipLaunchKernelGGL(k_undistort_grid, dim3(batch), dim3(UG_T), (2 * GRID_CELLS + 16) * 4 + (((size_t)cap_k + 1) & ~(size_t)1) * 2, (hipStream_t)stream, d_kps, d_nkp,
                       count_stride, cap_k, *cam, *gp, d_kps_un, d_grid_start, d_grid_idx);
    if (timed) { (void)hipEventRecord(g_mt.ev[1], (hipStream_t)stream); g_mt.have_grid = true; }
    return launch_status();
}

// per query a row of SBP_WORK_PER_Q words, then one flag word per frame (k_sbp_frame -> k_sbp_resolve)
// (k_sbp_frame's rows are SBPF_ROW words; a frame it hands on is rewritten in the fallback kernels' own 66-word rows, which fit inside)
#define SBP_WORK_ROW (SBPF_ROW > SBP_WORK_PER_Q ? SBPF_ROW : SBP_WORK_PER_Q)
extern "C" size_t orbm_search_workspace_bytes(int batch, int cap_q) { return (size_t)batch * cap_q * SBP_WORK_ROW * 4 + (((size_t)batch * 4 + 15) & ~(size_t)15); }

static int sbp_launch(const orb_keypoint* d_kps, const uint8_t* d_desc, const float* d_u_right, const uint8_t* d_occupied0, const int32_t* d_kp_link,
                      int cells, const int32_t* d_nkp, int count_stride, int cap_k, const int32_t* d_grid_start, const int32_t* d_grid_idx,
                      const orbm_query* d_queries, const uint8_t* d_qdesc, const int32_t* d_nq, int cap_q, int batch,
                      const orbm_search_params* params, int32_t* d_q_match, int32_t* d_kp_match, int32_t* d_nmatches, void* d_work, void* stream) {
    if (!d_kps || !d_desc || !d_nkp || !d_grid_start || !d_grid_idx || !d_queries || !d_qdesc || !d_nq || !params || !d_q_match ||
        !d_kp_match || !d_nmatches || !d_work || cap_k < 1 || cap_k > 65535 || cap_q < 1 || batch < 1 || count_stride < 1)
        return ORB_E_INVALID;
    if (params->mode != ORBM_MODE_LOCAL_MAP && params->mode != ORBM_MODE_BEST_ONLY && params->mode != ORBM_MODE_INIT) return ORB_E_INVALID;
    if (params->mode == ORBM_MODE_INIT && (cap_q > 65534 || cells != GRID_CELLS)) return ORB_E_INVALID;
    const size_t smem = (32 + 8 + 64 * SBP_STAGE + 64) * 4 + (((size_t)cap_k + 15) & ~(size_t)15) + (params->mode == ORBM_MODE_INIT ? (size_t)cap_k * 4 : 0);
    if (smem > 64 * 1024) return ORB_E_INVALID;
    SbpArgs A;
    A.kps = d_kps; A.desc = d_desc; A.u_right = d_u_right; A.occupied0 = d_occupied0; A.nkp = d_nkp; A.cstride = count_stride; A.cap_k = cap_k;
    A.grid_start = d_grid_start; A.grid_idx = d_grid_idx; A.queries = d_queries; A.qdesc = d_qdesc; A.nq = d_nq; A.cap_q = cap_q;
    A.prm = *params; A.q_match = d_q_match; A.kp_match = d_kp_match; A.nmatches = d_nmatches; A.work = (uint32_t*)d_work;
    A.cells = cells; A.kp_link = d_kp_link;
    A.chi2_gate = 0; A.q_dist = nullptr; A.serial_flag = nullptr;
    for (int i = 0; i < 16; i++) A.inv_sigma2[i] = 0.f;
    // k_sbp_frame takes the occupancy modes of single-camera frames whose state fits its LDS layout and its K queries per thread; the frames it
    // flags, and every other call, take k_sbp_candidates2 -> k_sbp_resolve
    const int capk4 = (cap_k + 3) & ~3;
    const int tailq = cap_q > SBPF_T ? cap_q - SBPF_T : 0;
    const size_t smem_f = (32 + 8 + 8) * 4 + (size_t)capk4 * (12 + 12 + 4 * SBPF_DP) + (GRID_CELLS + 2) * 2 + (size_t)tailq * SBPF_SD * 4;
    // (> 64 KB of dynamic LDS: only where the device grants it to both instantiations — lds_optin.inc, once per device and kernel, thread safe;
    //  a device with a smaller LDS takes the rounds pair below like every call k_sbp_frame does not cover)
    const bool fused = SBP_FUSED_FRAME && params->mode != ORBM_MODE_INIT && !d_kp_link && cells == GRID_CELLS && smem_f <= 150 * 1024 &&
                       cap_q <= 2 * SBPF_T &&
                       orb_lds_optin(tailq ? (const void*)k_sbp_frame<true> : (const void*)k_sbp_frame<false>, smem_f) == ORB_OK;
    if (fused) A.serial_flag = (int32_t*)((uint32_t*)d_work + (size_t)batch * cap_q * SBP_WORK_ROW);
    const bool timed = mt_ready();
    if (timed) (void)hipEventRecord(g_mt.ev[2], (hipStream_t)stream);
    if (fused) {
        if (tailq) hipLaunchKernelGGL(k_sbp_frame<true>, dim3(batch), dim3(SBPF_T), smem_f, (hipStream_t)stream, A);
        else hipLaunchKernelGGL(k_sbp_frame<false>, dim3(batch), dim3(SBPF_T), smem_f, (hipStream_t)stream, A);
        if (timed) (void)hipEventRecord(g_mt.ev[3], (hipStream_t)stream);
        hipLaunchKernelGGL(k_sbp_candidates_flagged, dim3(batch), dim3(256), 8 * SBP_CAPC * 4, (hipStream_t)stream, A);
    } else {
#if SBP_HALF
        hipLaunchKernelGGL(k_sbp_candidates2, dim3((cap_q + 7) / 8, batch), dim3(256), 8 * SBP_CAPC * 4, (hipStream_t)stream, A);
#else
        hipLaunchKernelGGL(k_sbp_candidates, dim3((cap_q + 3) / 4, batch), dim3(256), 4 * SBP_CAPC * 4, (hipStream_t)stream, A);
#endif
        if (timed) (void)hipEventRecord(g_mt.ev[3], (hipStream_t)stream);
    }
    hipLaunchKernelGGL(k_sbp_resolve, dim3(batch), dim3(64), smem, (hipStream_t)stream, A);
    if (timed) { (void)hipEventRecord(g_mt.ev[4], (hipStream_t)stream); g_mt.have_sbp = true; }
    return launch_status();
}

extern "C" int orbm_search_by_projection(const orb_keypoint* d_kps, const uint8_t* d_desc, const float* d_u_right, const uint8_t* d_occupied0,
                                         const int32_t* d_nkp, int count_stride, int cap_k, const int32_t* d_grid_start,
                                         const int32_t* d_grid_idx, const orbm_query* d_queries, const uint8_t* d_qdesc,
                                         const int32_t* d_nq, int cap_q, int batch, const orbm_search_params* params,
                                         int32_t* d_q_match, int32_t* d_kp_match, int32_t* d_nmatches, void* d_work, void* stream) {
    return sbp_launch(d_kps, d_desc, d_u_right, d_occupied0, nullptr, GRID_CELLS, d_nkp, count_stride, cap_k, d_grid_start, d_grid_idx, d_queries,
                      d_qdesc, d_nq, cap_q, batch, params, d_q_match, d_kp_match, d_nmatches, d_work, stream);
}

extern "C" int orbm_search_by_projection_rig(const orb_keypoint* d_kps, const uint8_t* d_desc, const uint8_t* d_occupied0,
                                             const int32_t* d_kp_link, const int32_t* d_nkp, int count_stride, int cap_k,
                                             const int32_t* d_grid_start, const int32_t* d_grid_idx, const orbm_query* d_queries,
                                             const uint8_t* d_qdesc, const int32_t* d_nq, int cap_q, int batch, const orbm_search_params* params,
                                             int32_t* d_q_match, int32_t* d_kp_match, int32_t* d_nmatches, void* d_work, void* stream) {
    if (params && params->mode == ORBM_MODE_INIT) return ORB_E_INVALID;
    return sbp_launch(d_kps, d_desc, nullptr, d_occupied0, d_kp_link, 2 * GRID_CELLS, d_nkp, count_stride, cap_k, d_grid_start, d_grid_idx, d_queries,
                      d_qdesc, d_nq, cap_q, batch, params, d_q_match, d_kp_match, d_nmatches, d_work, stream);
}

extern "C" int orbm_search_by_bow(const orbm_bow_side* kf, const uint8_t* d_kf_valid, const orbm_bow_side* f, int batch, float nn_ratio,
                                  int check_orientation, int32_t* d_f_match, int32_t* d_nmatches, void* stream) {
    if (!kf || !f || !d_kf_valid || !d_f_match || !d_nmatches || batch < 1 || kf->cap_f < 1 || f->cap_f < 1 || kf->cap_nodes < 1 || f->cap_nodes < 1)
        return ORB_E_INVALID;
    const size_t smem = (32 + 8 + (size_t)kf->cap_nodes) * 4 + (((size_t)f->cap_f + 15) & ~(size_t)15);
    if (smem > 64 * 1024) return ORB_E_INVALID;
    BowArgs A;
    A.kf = *kf; A.f = *f; A.kf_valid = d_kf_valid; A.f_valid = nullptr; A.nn_ratio = nn_ratio; A.check_orientation = check_orientation;
    A.f_match = d_f_match; A.nmatches = d_nmatches;
    hipLaunchKernelGGL(k_bow<false>, dim3(batch), dim3(256), smem, (hipStream_t)stream, A);
    return launch_status();
}

extern "C" int orbm_search_by_bow_kf(const orbm_bow_side* kf1, const uint8_t* d_valid1, const orbm_bow_side* kf2, const uint8_t* d_valid2, int batch,
                                     float nn_ratio, int check_orientation, int32_t* d_match12, int32_t* d_nmatches, void* stream) {
    if (!kf1 || !kf2 || !d_valid1 || !d_valid2 || !d_match12 || !d_nmatches || batch < 1 || kf1->cap_f < 1 || kf2->cap_f < 1 || kf1->cap_nodes < 1 ||
        kf2->cap_nodes < 1)
        return ORB_E_INVALID;
    const size_t smem = (32 + 8 + (size_t)kf1->cap_nodes) * 4 + (((size_t)kf2->cap_f + 15) & ~(size_t)15);
    if (smem > 64 * 1024) return ORB_E_INVALID;
    BowArgs A;
    A.kf = *kf1; A.f = *kf2; A.kf_valid = d_valid1; A.f_valid = d_valid2; A.nn_ratio = nn_ratio; A.check_orientation = check_orientation;
    A.f.n_left = nullptr;
    A.f_match = d_match12; A.nmatches = d_nmatches;
    hipLaunchKernelGGL(k_bow<true>, dim3(batch), dim3(256), smem, (hipStream_t)stream, A);
    return launch_status();
}

extern "C" int orbm_fuse(const orb_keypoint* d_kps, const uint8_t* d_desc, const float* d_u_right, const int32_t* d_nkp, int count_stride,
                         int cap_k, const int32_t* d_grid_start, const int32_t* d_grid_idx, const orbm_query* d_queries,
                         const uint8_t* d_qdesc, const int32_t* d_nq, int cap_q, int batch, const orbm_fuse_params* params,
                         int32_t* d_q_match, int32_t* d_q_dist, int32_t* d_nfused, void* stream) {
    if (!d_kps || !d_desc || !d_nkp || !d_grid_start || !d_grid_idx || !d_queries || !d_qdesc || !d_nq || !params || !d_q_match || !d_q_dist ||
        !d_nfused || cap_k < 1 || cap_k > 65535 || cap_q < 1 || batch < 1 || count_stride < 1)
        return ORB_E_INVALID;
    SbpArgs A;
    A.kps = d_kps; A.desc = d_desc; A.u_right = d_u_right; A.occupied0 = nullptr; A.nkp = d_nkp; A.cstride = count_stride; A.cap_k = cap_k;
    A.grid_start = d_grid_start; A.grid_idx = d_grid_idx; A.queries = d_queries; A.qdesc = d_qdesc; A.nq = d_nq; A.cap_q = cap_q;
    A.prm.mode = ORBM_MODE_BEST_ONLY; A.prm.th_dist = params->th_dist; A.prm.nn_ratio = 1.f; A.prm.check_orientation = 0; A.prm.grid = params->grid;
    A.q_match = d_q_match; A.kp_match = nullptr; A.nmatches = d_nfused; A.work = nullptr; A.q_dist = d_q_dist;
    A.chi2_gate = params->chi2_gate ? 1 : 0; A.cells = GRID_CELLS; A.kp_link = nullptr; A.serial_flag = nullptr;
    for (int i = 0; i < 16; i++) A.inv_sigma2[i] = params->inv_level_sigma2[i];
    if (hipMemsetAsync(d_nfused, 0, (size_t)batch * 4, (hipStream_t)stream) != hipSuccess) return ORB_E_HIP;
    hipLaunchKernelGGL(k_fuse, dim3((cap_q + 3) / 4, batch), dim3(256), 0, (hipStream_t)stream, A);
    return launch_status();
}

extern "C" int orbm_search_for_triangulation(const orbm_tri_side* kf1, const orbm_tri_side* kf2, const orbm_tri_pair* d_pairs, int batch,
                                             int only_stereo, int coarse, int check_orientation, int32_t* d_match12, int32_t* d_nmatches,
                                             void* stream) {
    if (!kf1 || !kf2 || !d_pairs || !d_match12 || !d_nmatches || batch < 1 || kf1->cap_f < 1 || kf2->cap_f < 1 || kf1->cap_nodes < 1 ||
        kf2->cap_nodes < 1 || !kf1->kps || !kf2->kps || !kf1->desc || !kf2->desc || !kf1->has_mp || !kf2->has_mp)
        return ORB_E_INVALID;
    const size_t smem = (32 + 8 + (size_t)kf1->cap_nodes) * 4 + (((size_t)kf1->cap_f + 15) & ~(size_t)15);
    if (smem > 64 * 1024) return ORB_E_INVALID;
    TriArgs A;
    A.k1 = *kf1; A.k2 = *kf2; A.pairs = d_pairs; A.only_stereo = only_stereo; A.coarse = coarse; A.check_orientation = check_orientation;
    A.kb8 = nullptr; A.nleft1 = A.nleft2 = nullptr;
    A.match12 = d_match12; A.nmatches = d_nmatches;
    hipLaunchKernelGGL(k_tri<false>, dim3(batch), dim3(256), smem, (hipStream_t)stream, A);
    return launch_status();
}

extern "C" int orbm_search_for_triangulation_kb8(const orbm_tri_side* kf1, const orbm_tri_side* kf2, const int32_t* d_nleft1, const int32_t* d_nleft2,
                                                 const orbm_tri_kb8_pair* d_pairs, int batch, int only_stereo, int coarse, int check_orientation,
                                                 int32_t* d_match12, int32_t* d_nmatches, void* stream) {
    if (!kf1 || !kf2 || !d_pairs || !d_nleft1 || !d_nleft2 || !d_match12 || !d_nmatches || batch < 1 || kf1->cap_f < 1 || kf2->cap_f < 1 ||
        kf1->cap_nodes < 1 || kf2->cap_nodes < 1 || !kf1->kps || !kf2->kps || !kf1->desc || !kf2->desc || !kf1->has_mp || !kf2->has_mp)
        return ORB_E_INVALID;
    const size_t smem = (32 + 8 + (size_t)kf1->cap_nodes) * 4 + (((size_t)kf1->cap_f + 15) & ~(size_t)15);
    if (smem > 64 * 1024) return ORB_E_INVALID;
    TriArgs A;
    A.k1 = *kf1; A.k2 = *kf2; A.pairs = nullptr; A.only_stereo = only_stereo; A.coarse = coarse; A.check_orientation = check_orientation;
    A.kb8 = d_pairs; A.nleft1 = d_nleft1; A.nleft2 = d_nleft2;
    A.match12 = d_match12; A.nmatches = d_nmatches;
    hipLaunchKernelGGL(k_tri<true>, dim3(batch), dim3(256), smem, (hipStream_t)stream, A);
    return launch_status();
}

// ---- SearchBySim3's agreement pass (ORBmatcher.cc:2203-2219): vpMatches12[i1] = idx2 iff vnMatch1[i1] == idx2 and vnMatch2[idx2] == i1
static __global__ __launch_bounds__(256) void k_mutual(const int32_t* m12, const int32_t* m21, const int32_t* n1p, const int32_t* n2p, const int cap1,
                                                        const int cap2, int32_t* out12, int32_t* nfound) {
    const int b = blockIdx.y, i1 = blockIdx.x * 256 + threadIdx.x;
    const int n1 = min(n1p[b], cap1), n2 = min(n2p[b], cap2);
    bool ok = false;
    if (i1 < cap1) {
        int idx2 = -1;
        if (i1 < n1) { idx2 = m12[(size_t)b * cap1 + i1]; ok = idx2 >= 0 && idx2 < n2 && m21[(size_t)b * cap2 + idx2] == i1; }
        out12[(size_t)b * cap1 + i1] = ok ? idx2 : -1;
    }
    const int c = __popcll(__ballot(ok));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&nfound[b], c);
}
extern "C" int orbm_mutual_matches(const int32_t* d_match12, const int32_t* d_match21, const int32_t* d_n1, const int32_t* d_n2, int cap1, int cap2,
                                   int batch, int32_t* d_out12, int32_t* d_nfound, void* stream) {
    if (!d_match12 || !d_match21 || !d_n1 || !d_n2 || !d_out12 || !d_nfound || cap1 <= 0 || cap2 <= 0 || batch < 0) return ORB_E_INVALID;
    if (batch == 0) return ORB_OK;
    if (hipMemsetAsync(d_nfound, 0, (size_t)batch * 4, (hipStream_t)stream) != hipSuccess) return ORB_E_HIP;
    hipLaunchKernelGGL(k_mutual, dim3((cap1 + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, d_match12, d_match21, d_n1, d_n2, cap1, cap2, d_out12, d_nfound);
    return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP;
}
