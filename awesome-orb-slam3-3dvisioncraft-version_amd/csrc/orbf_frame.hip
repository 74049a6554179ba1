// orbf_frame.hip — the Frame-constructor steps between the extractor and the matcher ("callers either side
// of the path"): Frame::UndistortKeyPoints (Frame.cc:874-925), Frame::ComputeImageBounds (:926-953) with the grid scalars of
// Frame.cc:388-399, Frame::ComputeStereoFromRGBD (:1136-1157).  All three are per-keypoint maps: one thread per keypoint, HBM bound
// (28 B in + 28 B out per keypoint), nothing to tile.
//
// cv::undistortPoints (OpenCV 3.x, un-vendored: restated from the published algorithm; parity unpinned against a
// real OpenCV build): normalise with the camera matrix in double, 5 fixed-point iterations of the Brown-Conrady inverse, re-project
// with the new camera matrix P = K, round to float.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/orbhip.h"

#include "frame_undistort.inc"

static __global__ __launch_bounds__(256) void k_undistort(const orb_keypoint* kps, const int32_t* nkp, const int countStride, const int capK,
                                                          const orbf_camera cam, orb_keypoint* out) {
    const int b = blockIdx.y;
    const int n = min(nkp[(size_t)b * countStride], capK);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    orb_keypoint kp = kps[(size_t)b * capK + i];
    if (cam.dist[0] != 0.0f) undistort_point(cam, kp.x, kp.y, &kp.x, &kp.y);   // Frame.cc:879-883: k1 == 0 -> mvKeysUn = mvKeys
    out[(size_t)b * capK + i] = kp;
}

static __global__ void k_bounds(const orbf_camera cam, const float w, const float h, float* out) {
    const int t = threadIdx.x;
    if (t >= 4) return;
    const float px = (t & 1) ? w : 0.0f, py = (t & 2) ? h : 0.0f;
    undistort_point(cam, px, py, &out[2 * t], &out[2 * t + 1]);
}

static __global__ __launch_bounds__(256) void k_rgbd(const orb_keypoint* kps, const orb_keypoint* kpsUn, const int32_t* nkp, const int countStride,
                                                     const int capK, const float* depth, const size_t frameStride, const int rowStride,
                                                     const float mbf, float* uRight, float* depthOut) {
    const int b = blockIdx.y;
    const int n = min(nkp[(size_t)b * countStride], capK);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= capK) return;
    float ur = -1.0f, dz = -1.0f;
    if (i < n) {
        const orb_keypoint kp = kps[(size_t)b * capK + i];
        // imDepth.at<float>(v, u) with float arguments: implicit float -> int conversion (truncation), Frame.cc:1146-1149
        const int v = (int)kp.y, u = (int)kp.x;
        const float d = depth[(size_t)b * frameStride + (size_t)v * rowStride + u];
        if (d > 0) { dz = d; ur = kpsUn[(size_t)b * capK + i].x - mbf / d; }
    }
    uRight[(size_t)b * capK + i] = ur;
    depthOut[(size_t)b * capK + i] = dz;
}

extern "C" int orbf_undistort_keypoints(const orb_keypoint* d_kps, const int32_t* d_nkp, int count_stride, int cap_k, int batch,
                                        const orbf_camera* cam, orb_keypoint* d_kps_un, void* stream) {
    if (!d_kps || !d_nkp || !cam || !d_kps_un || cap_k <= 0 || batch < 0 || count_stride <= 0) return ORB_E_INVALID;
    if (batch == 0) return ORB_OK;
    hipLaunchKernelGGL(k_undistort, dim3((cap_k + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, d_kps, d_nkp, count_stride, cap_k, *cam, d_kps_un);
    return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP;
}

extern "C" int orbf_image_bounds(const orbf_camera* cam, int width, int height, float bounds[4], orbm_grid_params* gp) {
    if (!cam || !bounds || width <= 0 || height <= 0) return ORB_E_INVALID;
    float c[8];
    if (cam->dist[0] != 0.0f) {
        float* d = nullptr;
        if (hipMalloc(&d, sizeof(c)) != hipSuccess) return ORB_E_NOMEM;
        hipLaunchKernelGGL(k_bounds, dim3(1), dim3(64), 0, 0, *cam, (float)width, (float)height, d);
        const hipError_t e = hipMemcpy(c, d, sizeof(c), hipMemcpyDeviceToHost);
        (void)hipFree(d);
        if (e != hipSuccess) return ORB_E_HIP;
        // corners: 0 = (0,0), 1 = (w,0), 2 = (0,h), 3 = (w,h)   (Frame.cc:941-944)
        bounds[0] = fminf(c[0], c[4]); bounds[1] = fmaxf(c[2], c[6]);
        bounds[2] = fminf(c[1], c[3]); bounds[3] = fmaxf(c[5], c[7]);
    } else {
        bounds[0] = 0.0f; bounds[1] = (float)width; bounds[2] = 0.0f; bounds[3] = (float)height;
    }
    if (gp) {
        gp->min_x = bounds[0]; gp->min_y = bounds[2];
        gp->grid_w_inv = (float)ORBM_GRID_COLS / (float)(bounds[1] - bounds[0]);   // Frame.cc:394-397
        gp->grid_h_inv = (float)ORBM_GRID_ROWS / (float)(bounds[3] - bounds[2]);
    }
    return ORB_OK;
}

extern "C" int orbf_stereo_from_rgbd(const orb_keypoint* d_kps, const orb_keypoint* d_kps_un, const int32_t* d_nkp, int count_stride, int cap_k,
                                     int batch, const float* d_depth, size_t frame_stride, int row_stride, int width, int height, float mbf,
                                     float* d_u_right, float* d_depth_out, void* stream) {
    if (!d_kps || !d_kps_un || !d_nkp || !d_depth || !d_u_right || !d_depth_out || cap_k <= 0 || batch < 0 || count_stride <= 0 ||
        width <= 0 || height <= 0 || row_stride < width) return ORB_E_INVALID;
    if (batch == 0) return ORB_OK;
    hipLaunchKernelGGL(k_rgbd, dim3((cap_k + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, d_kps, d_kps_un, d_nkp, count_stride, cap_k,
                       d_depth, frame_stride, row_stride, mbf, d_u_right, d_depth_out);
    return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP;
}

// ---- Frame::ComputeStereoFishEyeMatches (Frame.cc:1281-1325) + KannalaBrandt8::TriangulateMatches (KannalaBrandt8.cpp:334-400) -----------------
// One thread per left-camera keypoint of the lapping area: brute-force 2-NN over the right camera's lapping keypoints (all lanes read the same
// train descriptor: broadcast loads), ratio test, triangulation, gates.  Float32 cv::Mat arithmetic / libm float calls / cv::SVD of the reference
// follow rule R4 (the same rule oracle/frame_oracle.cpp states): transcendental functions in double on the float argument,
// rounded to float; float products as double-accumulated sums rounded once; the null vector of the 4x4 system by cyclic Jacobi on A^T A in double.
#include "kb8_geom.inc"
static __device__ __forceinline__ float triangulate_matches(const orbf_fisheye_rig& G, const orb_keypoint& kp1, const orb_keypoint& kp2, const float sigmaLevel,
                                                            const float unc, float* x3D) {
    return kb8_triangulate_matches(G.k_left, G.k_right, G.R_lr, G.t_lr, kp1, kp2, sigmaLevel, unc, x3D);
}

struct FishArgs {
    const orb_keypoint* kl; const uint8_t* dl; const int32_t* nl; const int32_t* ml;
    const orb_keypoint* kr; const uint8_t* dr; const int32_t* nr; const int32_t* mr;
    int capL, capR, cstride;
    orbf_fisheye_rig rig;
    int32_t* l2r; int32_t* r2l; float* depth; float* p3d; int32_t* nmatches;
};
static __global__ __launch_bounds__(256) void k_fisheye_init(FishArgs A) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i < A.capL) { A.l2r[(size_t)b * A.capL + i] = -1; A.depth[(size_t)b * A.capL + i] = -1.0f; for (int c = 0; c < 3; c++) A.p3d[((size_t)b * A.capL + i) * 3 + c] = 0.f; }
    if (i < A.capR) A.r2l[(size_t)b * A.capR + i] = -1;
    if (i == 0) A.nmatches[b] = 0;
}
static __global__ __launch_bounds__(256) void k_fisheye_match(FishArgs A) {
    const int b = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    const int nleft = min(A.nl[(size_t)b * A.cstride], A.capL), nright = min(A.nr[(size_t)b * A.cstride], A.capR);
    const int monoL = min(max(A.ml[(size_t)b * A.cstride], 0), nleft), monoR = min(max(A.mr[(size_t)b * A.cstride], 0), nright);
    const int nq = nleft - monoL, nt = nright - monoR;
    const bool act = q < nq;
    const int iq = min(monoL + q, max(nleft - 1, 0));
    const uint32_t* dq = (const uint32_t*)(A.dl + ((size_t)b * A.capL + iq) * 32);
    uint32_t Q[8];
#pragma unroll
    for (int k = 0; k < 8; k++) Q[k] = (act || nleft > 0) ? dq[k] : 0u;
    // BFMatcher(NORM_HAMMING).knnMatch(k = 2): ascending train scan, strict '<' (the rule of orbm_knn2)
    int d0 = 256, d1 = 256, i0 = -1, i1 = -1;
    const uint32_t* dt = (const uint32_t*)(A.dr + ((size_t)b * A.capR + monoR) * 32);
    for (int j = 0; j < nt; j++) {
        int d = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) d += __popc(Q[k] ^ dt[(size_t)j * 8 + k]);
        if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
        else if (d < d1) { d1 = d; i1 = j; }
    }
    if (!act || i1 < 0) return;
    if (!((double)(float)d0 < (double)(float)d1 * 0.7)) return;
    const orb_keypoint k1 = A.kl[(size_t)b * A.capL + monoL + q];
    const orb_keypoint k2 = A.kr[(size_t)b * A.capR + monoR + i0];
    float x3D[3];
    const float dep = triangulate_matches(A.rig, k1, k2, A.rig.level_sigma2[min(max(k1.octave, 0), 15)], A.rig.level_sigma2[min(max(k2.octave, 0), 15)], x3D);
    if (dep > 0.0001f) {
        const int gl = monoL + q, gr = monoR + i0;
        A.l2r[(size_t)b * A.capL + gl] = gr;
        atomicMax(&A.r2l[(size_t)b * A.capR + gr], gl);     // the serial loop's last writer = the largest left index
        A.depth[(size_t)b * A.capL + gl] = dep;
        for (int c = 0; c < 3; c++) A.p3d[((size_t)b * A.capL + gl) * 3 + c] = x3D[c];
        atomicAdd(&A.nmatches[b], 1);
    }
}

extern "C" int orbf_stereo_fisheye_matches(const orb_keypoint* d_kps_l, const uint8_t* d_desc_l, const int32_t* d_n_l, const int32_t* d_mono_l,
                                           const orb_keypoint* d_kps_r, const uint8_t* d_desc_r, const int32_t* d_n_r, const int32_t* d_mono_r, int cap_l,
                                           int cap_r, int count_stride, int batch, const orbf_fisheye_rig* rig, int32_t* d_left_to_right,
                                           int32_t* d_right_to_left, float* d_depth, float* d_p3d, int32_t* d_nmatches, void* stream) {
    if (!d_kps_l || !d_desc_l || !d_n_l || !d_mono_l || !d_kps_r || !d_desc_r || !d_n_r || !d_mono_r || !rig || !d_left_to_right || !d_right_to_left ||
        !d_depth || !d_p3d || !d_nmatches || cap_l <= 0 || cap_r <= 0 || count_stride <= 0 || batch < 0) return ORB_E_INVALID;
    if (batch == 0) return ORB_OK;
    FishArgs A{d_kps_l, d_desc_l, d_n_l, d_mono_l, d_kps_r, d_desc_r, d_n_r, d_mono_r, cap_l, cap_r, count_stride, *rig,
               d_left_to_right, d_right_to_left, d_depth, d_p3d, d_nmatches};
    hipLaunchKernelGGL(k_fisheye_init, dim3((std::max(cap_l, cap_r) + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, A);
    hipLaunchKernelGGL(k_fisheye_match, dim3((cap_l + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, A);
    return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP;
}
