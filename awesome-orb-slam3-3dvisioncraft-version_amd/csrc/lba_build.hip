// lba_build.hip — stage 3 of the hot path on gfx950: the linearisation of Optimizer::LocalBundleAdjustment
// (reference src/Optimizer.cc:1957-2344 graph; g2o BlockSolver::buildSystem block_solver.hpp:502-560).
//
//   k_lba_landmarks  one thread per EDGE (workgroup = 32 consecutive landmarks = a contiguous run of the landmark-major edges): computeError +
//                    linearizeOplus + the landmark half of constructQuadraticForm (H_ll, b_l accumulated in registers in
//                    the reference's edge order) and the pose-landmark block H_pl of every edge.
//   k_lba_poses      one wave per free pose walks the pose's edge list (CSR built once per optimize(), like
//                    BlockSolver::buildStructure) and reduces H_pp / b_p with a fixed butterfly -> deterministic.
//   k_lba_errors     one thread per edge: computeActiveErrors / chi2 / Huber rho (LM trial evaluation).
// All arithmetic is FP64 VALU (map state is float32 widened by the adapter).  The per-edge Jacobians are recomputed in
// the pose pass instead of being stored: ~350 flop/edge is cheaper than 200+ B/edge of extra HBM traffic (H8).
// No f64 atomics on H: every block has exactly one writer.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../../include/orbhip.h"
#include "lds_optin.inc"

struct Quat { double x, y, z, w; };
struct SE3 { Quat r; double t[3]; };
struct Lin { int D; double e[3], A[9], B[18], chi2, rho0, rho1, depth; };  // A: D x 3, B: D x 6, row-major

static __device__ __forceinline__ void quat_normalize(Quat& q) {  // SE3Quat::normalizeRotation, se3quat.h:283-288
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}
static __device__ __forceinline__ void quat_to_R(const Quat& q, double R[9]) {  // Eigen toRotationMatrix, row-major
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static __device__ __forceinline__ void quat_rotate(const Quat& q, const double v[3], double out[3]) {  // Eigen _transformVector
    double ux = q.y * v[2] - q.z * v[1], uy = q.z * v[0] - q.x * v[2], uz = q.x * v[1] - q.y * v[0];
    ux += ux; uy += uy; uz += uz;
    out[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
    out[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
    out[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}
static __device__ __forceinline__ void se3_map(const SE3& T, const double x[3], double out[3]) {  // se3quat.h:217
    quat_rotate(T.r, x, out);
    out[0] += T.t[0]; out[1] += T.t[1]; out[2] += T.t[2];
}
static __device__ __forceinline__ SE3 se3_mul(const SE3& a, const SE3& b) {  // se3quat.h:103-109
    SE3 r = a;
    double rt[3];
    quat_rotate(a.r, b.t, rt);
    r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
    Quat q;
    q.w = a.r.w * b.r.w - a.r.x * b.r.x - a.r.y * b.r.y - a.r.z * b.r.z;
    q.x = a.r.w * b.r.x + a.r.x * b.r.w + a.r.y * b.r.z - a.r.z * b.r.y;
    q.y = a.r.w * b.r.y + a.r.y * b.r.w + a.r.z * b.r.x - a.r.x * b.r.z;
    q.z = a.r.w * b.r.z + a.r.z * b.r.w + a.r.x * b.r.y - a.r.y * b.r.x;
    r.r = q;
    quat_normalize(r.r);
    return r;
}

static __device__ __forceinline__ void cam_project(const lba_camera& c, const double v[3], double res[2]) {
    if (c.model == LBA_CAM_PINHOLE) {  // Pinhole.cpp:43-49
        res[0] = c.p[0] * v[0] / v[2] + c.p[2];
        res[1] = c.p[1] * v[1] / v[2] + c.p[3];
    } else {
        // KannalaBrandt8.cpp:52-66 rounds theta and psi through atan2f/sqrtf; reproduced as float(atan2(double)) — a
        // correctly rounded float result, which is what glibc's atan2f returns in all but rare double-rounding cases.
        const double x2_plus_y2 = v[0] * v[0] + v[1] * v[1];
        const float rf = sqrtf((float)x2_plus_y2);
        const double theta = (double)(float)atan2((double)rf, (double)(float)v[2]);
        const double psi = (double)(float)atan2((double)(float)v[1], (double)(float)v[0]);
        const double theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
                     theta9 = theta7 * theta2;
        const double r = theta + c.p[4] * theta3 + c.p[5] * theta5 + c.p[6] * theta7 + c.p[7] * theta9;
        res[0] = c.p[0] * r * cos(psi) + c.p[2];
        res[1] = c.p[1] * r * sin(psi) + c.p[3];
    }
}
static __device__ __forceinline__ void cam_project_jac(const lba_camera& c, const double v[3], double J[6]) {
    if (c.model == LBA_CAM_PINHOLE) {  // Pinhole.cpp:89-100
        J[0] = c.p[0] / v[2]; J[1] = 0; J[2] = -c.p[0] * v[0] / (v[2] * v[2]);
        J[3] = 0; J[4] = c.p[1] / v[2]; J[5] = -c.p[1] * v[1] / (v[2] * v[2]);
    } else {  // KannalaBrandt8.cpp:166-196
        const double x2 = v[0] * v[0], y2 = v[1] * v[1], z2 = v[2] * v[2];
        const double r2 = x2 + y2, r = sqrt(r2), r3 = r2 * r;
        const double theta = atan2(r, v[2]);
        const double theta2 = theta * theta, theta3 = theta2 * theta, theta4 = theta2 * theta2, theta5 = theta4 * theta,
                     theta6 = theta2 * theta4, theta7 = theta6 * theta, theta8 = theta4 * theta4, theta9 = theta8 * theta;
        const double f = theta + theta3 * c.p[4] + theta5 * c.p[5] + theta7 * c.p[6] + theta9 * c.p[7];
        const double fd = 1 + 3 * c.p[4] * theta2 + 5 * c.p[5] * theta4 + 7 * c.p[6] * theta6 + 9 * c.p[7] * theta8;
        J[0] = c.p[0] * (fd * v[2] * x2 / (r2 * (r2 + z2)) + f * y2 / r3);
        J[3] = c.p[1] * (fd * v[2] * v[1] * v[0] / (r2 * (r2 + z2)) - f * v[1] * v[0] / r3);
        J[1] = c.p[0] * (fd * v[2] * v[1] * v[0] / (r2 * (r2 + z2)) - f * v[1] * v[0] / r3);
        J[4] = c.p[1] * (fd * v[2] * y2 / (r2 * (r2 + z2)) + f * x2 / r3);
        J[2] = -c.p[0] * fd * v[0] / (r2 + z2);
        J[5] = -c.p[1] * fd * v[1] / (r2 + z2);
    }
}

// computeError (+ linearizeOplus when WITH_JAC) + chi2 + Huber for one edge
// MP: every edge of the batch is an EdgeSE3ProjectXYZ on a pinhole camera (the monocular / pinhole configurations; lba_optimize checks the
// batch once per call).  Same expressions in the same order as the generic path takes for such an edge — identical results — but the
// third residual row, the fisheye model and the right-camera transform are gone at compile time: the linearisation kernels need a third
// fewer registers and multiply no structural zeros.
// KS = 2: every edge is an EdgeSE3ProjectXYZ or an EdgeStereoSE3ProjectXYZ on a pinhole camera (stereo / RGB-D pinhole maps): the stereo
// branch below (pinhole by construction) or the monocular-pinhole one; the fisheye model and the right-camera edge are gone.
template <bool WITH_JAC, int KS = 0>
static __device__ __forceinline__ void edge_linearize(const lba_edge& E, const SE3& T, const double X[3], const lba_camera& cam,
                                                     double huberMono, double huberStereo, Lin& L) {
    constexpr bool MP = KS == 1;
    L.e[2] = 0;
    if (MP || (KS == 2 && E.kind != LBA_EDGE_STEREO)) {
        L.D = 2;
        double proj[2], xl[3];
        se3_map(T, X, xl);
        proj[0] = cam.p[0] * xl[0] / xl[2] + cam.p[2];      // Pinhole.cpp:43-49
        proj[1] = cam.p[1] * xl[1] / xl[2] + cam.p[3];
        L.depth = xl[2];
        L.e[0] = (double)E.obs[0] - proj[0];
        L.e[1] = (double)E.obs[1] - proj[1];
        if (WITH_JAC) {
#pragma unroll
            for (int i = 6; i < 9; i++) L.A[i] = 0.0;
#pragma unroll
            for (int i = 12; i < 18; i++) L.B[i] = 0.0;
            double Rm[9], Jp[6];
            quat_to_R(T.r, Rm);
            Jp[0] = cam.p[0] / xl[2]; Jp[1] = 0; Jp[2] = -cam.p[0] * xl[0] / (xl[2] * xl[2]);   // Pinhole.cpp:89-100
            Jp[3] = 0; Jp[4] = cam.p[1] / xl[2]; Jp[5] = -cam.p[1] * xl[1] / (xl[2] * xl[2]);
#pragma unroll
            for (int i = 0; i < 6; i++) Jp[i] = -Jp[i];
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) L.A[r * 3 + c] = Jp[r * 3] * Rm[c] + Jp[r * 3 + 1] * Rm[3 + c] + Jp[r * 3 + 2] * Rm[6 + c];
            const double x = xl[0], y = xl[1], z = xl[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const double m0 = Jp[r * 3], m1 = Jp[r * 3 + 1], m2 = Jp[r * 3 + 2];
                L.B[r * 6 + 0] = m0 * 0 + m1 * (-z) + m2 * y;
                L.B[r * 6 + 1] = m0 * z + m1 * 0 + m2 * (-x);
                L.B[r * 6 + 2] = m0 * (-y) + m1 * x + m2 * 0;
                L.B[r * 6 + 3] = m0 * 1 + m1 * 0 + m2 * 0;
                L.B[r * 6 + 4] = m0 * 0 + m1 * 1 + m2 * 0;
                L.B[r * 6 + 5] = m0 * 0 + m1 * 0 + m2 * 1;
            }
        }
    } else if (E.kind == LBA_EDGE_STEREO) {
        L.D = 3;
        double xt[3];
        se3_map(T, X, xt);
        const double fx = cam.p[0], fy = cam.p[1], cx = cam.p[2], cy = cam.p[3];
        const float bf = (float)cam.bf;          // cam_project(const Vector3d&, const float& bf)
        const float invz = 1.0f / xt[2];          // float invz: types_six_dof_expmap.cpp:191 (double divide, rounded to float)
        double proj[3];
        proj[0] = xt[0] * invz * fx + cx;
        proj[1] = xt[1] * invz * fy + cy;
        proj[2] = proj[0] - bf * invz;
        L.e[0] = (double)E.obs[0] - proj[0]; L.e[1] = (double)E.obs[1] - proj[1]; L.e[2] = (double)E.obs[2] - proj[2];
        L.depth = xt[2];
        if (WITH_JAC) {
            double R[9];
            quat_to_R(T.r, R);
            const double x = xt[0], y = xt[1], z = xt[2], z_2 = z * z, bfd = cam.bf;
            double* A = L.A; double* B = L.B;
            A[0] = -fx * R[0] / z + fx * x * R[6] / z_2; A[1] = -fx * R[1] / z + fx * x * R[7] / z_2; A[2] = -fx * R[2] / z + fx * x * R[8] / z_2;
            A[3] = -fy * R[3] / z + fy * y * R[6] / z_2; A[4] = -fy * R[4] / z + fy * y * R[7] / z_2; A[5] = -fy * R[5] / z + fy * y * R[8] / z_2;
            A[6] = A[0] - bfd * R[6] / z_2; A[7] = A[1] - bfd * R[7] / z_2; A[8] = A[2] - bfd * R[8] / z_2;
            B[0] = x * y / z_2 * fx; B[1] = -(1 + (x * x / z_2)) * fx; B[2] = y / z * fx; B[3] = -1. / z * fx; B[4] = 0; B[5] = x / z_2 * fx;
            B[6] = (1 + y * y / z_2) * fy; B[7] = -x * y / z_2 * fy; B[8] = -x / z * fy; B[9] = 0; B[10] = -1. / z * fy; B[11] = y / z_2 * fy;
            B[12] = B[0] - bfd * y / z_2; B[13] = B[1] + bfd * x / z_2; B[14] = B[2]; B[15] = B[3]; B[16] = 0; B[17] = B[5] - bfd / z_2;
        }
    } else if constexpr (KS == 0) {
        L.D = 2;
        if (WITH_JAC) {
#pragma unroll
            for (int i = 6; i < 9; i++) L.A[i] = 0.0;
#pragma unroll
            for (int i = 12; i < 18; i++) L.B[i] = 0.0;
        }
        double proj[2], xl[3], xp[3], Rm[9];
        se3_map(T, X, xl);
        Quat ql = {cam.trl_q[0], cam.trl_q[1], cam.trl_q[2], cam.trl_q[3]};
        if (E.kind == LBA_EDGE_MONO) {
            xp[0] = xl[0]; xp[1] = xl[1]; xp[2] = xl[2];
            cam_project(cam, xp, proj);
            L.depth = xl[2];
            if (WITH_JAC) quat_to_R(T.r, Rm);
        } else {
            SE3 Trl;
            Trl.r = ql;
            Trl.t[0] = cam.trl_t[0]; Trl.t[1] = cam.trl_t[1]; Trl.t[2] = cam.trl_t[2];
            const SE3 Trw = se3_mul(Trl, T);
            double xe[3];
            se3_map(Trw, X, xe);      // computeError: (mTrl * v1->estimate()).map(X)       OptimizableTypes.h:146
            cam_project(cam, xe, proj);
            se3_map(Trl, xl, xp);     // linearizeOplus: X_r = mTrl.map(T_lw.map(X_w))      OptimizableTypes.cpp:211
            L.depth = xe[2];
            if (WITH_JAC) quat_to_R(Trw.r, Rm);
        }
        L.e[0] = (double)E.obs[0] - proj[0];
        L.e[1] = (double)E.obs[1] - proj[1];
        if (WITH_JAC) {
            double Jp[6], M[6];
            cam_project_jac(cam, xp, Jp);
#pragma unroll
            for (int i = 0; i < 6; i++) Jp[i] = -Jp[i];
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) L.A[r * 3 + c] = Jp[r * 3] * Rm[c] + Jp[r * 3 + 1] * Rm[3 + c] + Jp[r * 3 + 2] * Rm[6 + c];
            if (E.kind == LBA_EDGE_MONO) {
#pragma unroll
                for (int i = 0; i < 6; i++) M[i] = Jp[i];
            } else {
                double Rl[9];
                quat_to_R(ql, Rl);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) M[r * 3 + c] = Jp[r * 3] * Rl[c] + Jp[r * 3 + 1] * Rl[3 + c] + Jp[r * 3 + 2] * Rl[6 + c];
            }
            const double x = xl[0], y = xl[1], z = xl[2];
            // SE3deriv = [0 z -y 1 0 0; -z 0 x 0 1 0; y -x 0 0 0 1]   (OptimizableTypes.cpp:165-168); B = M * SE3deriv, written out
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const double m0 = M[r * 3], m1 = M[r * 3 + 1], m2 = M[r * 3 + 2];
                L.B[r * 6 + 0] = m0 * 0 + m1 * (-z) + m2 * y;
                L.B[r * 6 + 1] = m0 * z + m1 * 0 + m2 * (-x);
                L.B[r * 6 + 2] = m0 * (-y) + m1 * x + m2 * 0;
                L.B[r * 6 + 3] = m0 * 1 + m1 * 0 + m2 * 0;
                L.B[r * 6 + 4] = m0 * 0 + m1 * 1 + m2 * 0;
                L.B[r * 6 + 5] = m0 * 0 + m1 * 0 + m2 * 1;
            }
        }
    }
    const double s = (double)E.inv_sigma2;
    double chi2 = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) chi2 += L.e[i] * s * L.e[i];   // e[2] == 0 for 2-D edges
    L.chi2 = chi2;
    const double delta = (!MP && E.kind == LBA_EDGE_STEREO) ? huberStereo : huberMono;   // KS == 2: runtime kind
    L.rho0 = chi2; L.rho1 = 1.;
    if (delta > 0) {   // RobustKernelHuber::robustify, robust_kernel_impl.cpp:78-91
        const double dsqr = delta * delta;
        if (!(chi2 <= dsqr)) { const double sq = sqrt(chi2); L.rho0 = 2 * sq * delta - dsqr; L.rho1 = delta / sq; }
    }
}

static __device__ __forceinline__ SE3 load_pose(const double* p) {
    SE3 T;
    T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2];
    T.r.x = p[3]; T.r.y = p[4]; T.r.z = p[5]; T.r.w = p[6];
    return T;
}

struct LbaArgs { lba_problem P; lba_system S; };

// Landmark half of buildSystem, one THREAD PER EDGE.  A workgroup owns LBA_LB consecutive landmarks and therefore a contiguous run of
// the landmark-major edge array; every thread linearises one edge (error, Huber, Jacobians), writes the per-edge outputs (H_pl block,
// err, chi2, rho, depth) and parks its H_ll / b_l contribution in LDS; one thread per landmark then adds its edges' contributions in
// edge order — the same summation order as a serial walk, so H_ll / b_l are bit-identical to the thread-per-landmark formulation, but
// the expensive part runs with 8x more parallelism (a landmark has ~8 observations) and coalesced edge loads.
#ifndef LBA_CT
#define LBA_CT 128     // edges per chunk = threads per workgroup (32 KB of LDS: 5 workgroups per CU)
#endif
#define LBA_LB (LBA_CT / 8)   // landmarks per workgroup
#ifndef LBA_MINW
#define LBA_MINW 1     // minimum waves per SIMD the register allocation must allow
#endif
template <int KS>
static __global__ __launch_bounds__(LBA_CT, LBA_MINW) void k_lba_landmarks(LbaArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    // ONE LDS region, used twice per chunk: first the chunk's H_pl blocks ([LBA_CT][18]: written per lane, stored to global memory coalesced), then
    // the per-edge H_ll / b_l contributions ([LBA_CT][13]: H_ll 9 column-major | b_l 3, padded against bank conflicts) — 144 instead of 248 bytes
    // of LDS per thread: the workgroups per CU are then limited by registers, not by LDS
    double (*contrib)[13] = (double (*)[13])orb_smem;
    double* stage = (double*)orb_smem;
    const lba_problem& P = A.P;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int nl = min(P.n_points[b], P.cap_l);
    const int l0 = blockIdx.x * LBA_LB;
    if (l0 >= nl) return;
    const int l1 = min(l0 + LBA_LB, nl);
    const int ne = min(P.n_edges[b], P.cap_e);
    const lba_edge* edges = P.edges + (size_t)b * P.cap_e;
    const double* poses = P.poses + (size_t)b * P.cap_p * 7;
    const int32_t* hidx = P.pose_hidx + (size_t)b * P.cap_p;
    const double* points = P.points + (size_t)b * P.cap_l * 3;
    const int32_t* lms = P.lm_start + (size_t)b * (P.cap_l + 1);
    const int eBegin = min(lms[l0], ne), eEnd = min(lms[l1], ne);
    // landmark thread j: landmark l0 + j, its edge range, its accumulators
    const int myL = l0 + tid;
    int ls = 0, le = 0;
    if (tid < LBA_LB && myL < l1) { ls = min(lms[myL], ne); le = min(lms[myL + 1], ne); }
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
    for (int c0 = eBegin; c0 < eEnd; c0 += LBA_CT) {
        const int ei = c0 + tid;
        double cv[12];   // this edge's H_ll / b_l contribution, parked in LDS after the H_pl blocks have left it
        if (ei < eEnd) {
            const lba_edge E = edges[ei];
            const SE3 T = load_pose(poses + (size_t)E.pose * 7);
            const double* Xp = points + (size_t)E.point * 3;
            const double X[3] = {Xp[0], Xp[1], Xp[2]};
            Lin L;
            edge_linearize<true, KS>(E, T, X, P.cameras[E.cam], P.huber_mono, P.huber_stereo, L);
            const size_t eo = (size_t)b * P.cap_e + ei;
            if (A.S.err) { A.S.err[eo * 3] = L.e[0]; A.S.err[eo * 3 + 1] = L.e[1]; A.S.err[eo * 3 + 2] = L.e[2]; }
            if (A.S.chi2) A.S.chi2[eo] = L.chi2;
            if (A.S.rho) { A.S.rho[eo * 2] = L.rho0; A.S.rho[eo * 2 + 1] = L.rho1; }
            if (A.S.depth) A.S.depth[eo] = L.depth;
            // constructQuadraticForm, robust branch (base_binary_edge.hpp:91-113): omega_r = -Omega e rho1; wOmega = rho1 Omega
            const double s = (double)E.inv_sigma2, w = L.rho1 * s;
            double om[3];
#pragma unroll
            for (int i = 0; i < 3; i++) om[i] = -s * L.e[i] * L.rho1;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double acc = 0;
#pragma unroll
                for (int r = 0; r < 3; r++) acc += L.A[r * 3 + c] * om[r];
                cv[9 + c] = acc;
#pragma unroll
                for (int c2 = 0; c2 < 3; c2++) {
                    double h = 0;
#pragma unroll
                    for (int r = 0; r < 3; r++) h += L.A[r * 3 + c] * w * L.A[r * 3 + c2];
                    cv[c2 * 3 + c] = h;
                }
            }
            if (A.S.Hpl) {
                double* hp = stage + tid * 18;
                const bool freePose = hidx[E.pose] >= 0;
#pragma unroll
                for (int c2 = 0; c2 < 3; c2++)
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        double h = 0;
                        if (freePose) {
#pragma unroll
                            for (int r = 0; r < 3; r++) h += L.B[r * 6 + c] * w * L.A[r * 3 + c2];
                        }
                        hp[c2 * 6 + c] = h;   // logical 6x3 block (pose, landmark), column-major (_hessianTransposed)
                    }
            }
        }
        if (A.S.Hpl) {        // the chunk's H_pl blocks are contiguous in global memory: 16 bytes per lane, consecutive lanes consecutive addresses
            __syncthreads();
            const int n2 = (min(eEnd, c0 + LBA_CT) - c0) * 9;
            double2* dst = (double2*)(A.S.Hpl + ((size_t)b * P.cap_e + c0) * 18);
            const double2* src = (const double2*)stage;
            for (int i = tid; i < n2; i += LBA_CT) dst[i] = src[i];
            __syncthreads();
        }
        if (ei < eEnd) {
#pragma unroll
            for (int i = 0; i < 12; i++) contrib[tid][i] = cv[i];
        }
        __syncthreads();
        if (tid < LBA_LB) {   // H_ll += A^T wOmega A, b_l += A^T omega_r in edge order
            const int a0 = max(ls, c0), a1 = min(le, c0 + LBA_CT);
            for (int e = a0; e < a1; e++) {
                const double* cb = contrib[e - c0];
#pragma unroll
                for (int i = 0; i < 9; i++) H[i] += cb[i];
#pragma unroll
                for (int i = 0; i < 3; i++) bl[i] += cb[9 + i];
            }
        }
        __syncthreads();
    }
    if (tid < LBA_LB && myL < l1) {
        if (A.S.Hll) { double* o = A.S.Hll + ((size_t)b * P.cap_l + myL) * 9; for (int i = 0; i < 9; i++) o[i] = H[i]; }
        if (A.S.bl) { double* o = A.S.bl + ((size_t)b * P.cap_l + myL) * 3; for (int i = 0; i < 3; i++) o[i] = bl[i]; }
    }
}

#ifndef POSES_NT
#define POSES_NT 256   // threads per pose: a pose of a 100-KF window has ~2 000 edges; one wave per pose leaves a 16-window batch at 1 280 waves of 31 serial steps
#endif
template <int KS>
static __global__ __launch_bounds__(POSES_NT) void k_lba_poses(LbaArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    double (*wsum)[27] = (double (*)[27])orb_smem;   // [POSES_NT / 64][27] per-wave sums
    const lba_problem& P = A.P;
    const int b = blockIdx.y, pi = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int np = min(P.n_poses[b], P.cap_p);
    if (pi >= np) return;
    const int h = P.pose_hidx[(size_t)b * P.cap_p + pi];
    if (h < 0) return;  // fixed vertex: no block (base_binary_edge.hpp:65-68)
    const int ne = min(P.n_edges[b], P.cap_e);
    const lba_edge* edges = P.edges + (size_t)b * P.cap_e;
    const double* points = P.points + (size_t)b * P.cap_l * 3;
    const int32_t* pe = P.pose_edges + (size_t)b * P.cap_e;
    const int s0 = P.pose_start[(size_t)b * (P.cap_p + 1) + pi], s1 = min(P.pose_start[(size_t)b * (P.cap_p + 1) + pi + 1], ne);
    const SE3 T = load_pose(P.poses + ((size_t)b * P.cap_p + pi) * 7);
    double acc[27];  // 21 upper-triangle entries of H_pp (column-major order c2 >= c) + 6 of b_p
#pragma unroll
    for (int i = 0; i < 27; i++) acc[i] = 0;
    for (int k = s0 + (int)threadIdx.x; k < s1; k += POSES_NT) {
        const lba_edge E = edges[pe[k]];
        const double* Xp = points + (size_t)E.point * 3;
        const double X[3] = {Xp[0], Xp[1], Xp[2]};
        Lin L;
        edge_linearize<true, KS>(E, T, X, P.cameras[E.cam], P.huber_mono, P.huber_stereo, L);
        const double s = (double)E.inv_sigma2, w = L.rho1 * s;
        double om[3];
        for (int i = 0; i < 3; i++) om[i] = -s * L.e[i] * L.rho1;
        int t = 0;
#pragma unroll
        for (int c2 = 0; c2 < 6; c2++)
#pragma unroll
            for (int c = 0; c <= c2; c++) {
                double hh = 0;
                for (int r = 0; r < 3; r++) hh += L.B[r * 6 + c] * w * L.B[r * 6 + c2];
                acc[t++] += hh;
            }
#pragma unroll
        for (int c = 0; c < 6; c++) {
            double a = 0;
            for (int r = 0; r < 3; r++) a += L.B[r * 6 + c] * om[r];
            acc[21 + c] += a;
        }
    }
#pragma unroll
    for (int i = 0; i < 27; i++)
        for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 27; i++) wsum[wave][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < 27) {   // waves added in fixed order; entry t of the packed upper triangle is (c, c2) with c <= c2, column-major
        const int t = threadIdx.x;
        double v = wsum[0][t];
        for (int w = 1; w < POSES_NT / 64; w++) v += wsum[w][t];
        if (t < 21) {
            int c2 = 0, base = 0;
            while (base + c2 + 1 <= t) { base += c2 + 1; c2++; }
            const int c = t - base;
            if (A.S.Hpp) { double* o = A.S.Hpp + ((size_t)b * P.cap_p + h) * 36; o[c2 * 6 + c] = v; o[c * 6 + c2] = v; }
        } else if (A.S.bp) A.S.bp[((size_t)b * P.cap_p + h) * 6 + (t - 21)] = v;
    }
}

static __global__ __launch_bounds__(256) void k_lba_errors(LbaArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    double* red = (double*)orb_smem;  // [256]
    const lba_problem& P = A.P;
    const int b = blockIdx.y, ei = blockIdx.x * 256 + threadIdx.x;
    const int ne = min(P.n_edges[b], P.cap_e);
    double r0 = 0;
    if (ei < ne) {
        const lba_edge E = P.edges[(size_t)b * P.cap_e + ei];
        const SE3 T = load_pose(P.poses + ((size_t)b * P.cap_p + E.pose) * 7);
        const double* Xp = P.points + ((size_t)b * P.cap_l + E.point) * 3;
        const double X[3] = {Xp[0], Xp[1], Xp[2]};
        Lin L;
        edge_linearize<false>(E, T, X, P.cameras[E.cam], P.huber_mono, P.huber_stereo, L);
        const size_t eo = (size_t)b * P.cap_e + ei;
        if (A.S.err) { A.S.err[eo * 3] = L.e[0]; A.S.err[eo * 3 + 1] = L.e[1]; A.S.err[eo * 3 + 2] = L.e[2]; }
        if (A.S.chi2) A.S.chi2[eo] = L.chi2;
        if (A.S.rho) { A.S.rho[eo * 2] = L.rho0; A.S.rho[eo * 2 + 1] = L.rho1; }
        if (A.S.depth) A.S.depth[eo] = L.depth;
        r0 = L.rho0;
    }
    red[threadIdx.x] = r0;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0 && A.S.robust_chi2_sum && red[0] != 0.0) atomicAdd(A.S.robust_chi2_sum + b, red[0]);
}

static int lba_check(const lba_problem* p, int batch, const lba_system* out) {
    if (!p || !out || batch < 1 || !p->poses || !p->pose_hidx || !p->points || !p->edges || !p->lm_start || !p->pose_start ||
        !p->pose_edges || !p->cameras || !p->n_poses || !p->n_points || !p->n_edges || p->cap_p < 1 || p->cap_l < 1 || p->cap_e < 1 ||
        p->n_cameras < 1)
        return ORB_E_INVALID;
    return ORB_OK;
}

static int lba_build_system_impl(const lba_problem* prob, int batch, const lba_system* out, int ks, void* stream) {
    int rc = lba_check(prob, batch, out);
    if (rc != ORB_OK) return rc;
    LbaArgs A;
    A.P = *prob; A.S = *out;
    hipStream_t st = (hipStream_t)stream;
    // blocks of fixed poses / rows >= n are defined as zero
    if (out->Hpp && hipMemsetAsync(out->Hpp, 0, (size_t)batch * prob->cap_p * 36 * 8, st) != hipSuccess) return ORB_E_HIP;
    if (out->bp && hipMemsetAsync(out->bp, 0, (size_t)batch * prob->cap_p * 6 * 8, st) != hipSuccess) return ORB_E_HIP;
    const dim3 gL((prob->cap_l + LBA_LB - 1) / LBA_LB, batch), gP(prob->cap_p, batch);
#define LBA_LAUNCH_KS(kern, ...) do { if (ks == 1) hipLaunchKernelGGL(kern<1>, __VA_ARGS__); else if (ks == 2) hipLaunchKernelGGL(kern<2>, __VA_ARGS__); \
                                      else hipLaunchKernelGGL(kern<0>, __VA_ARGS__); } while (0)
    LBA_LAUNCH_KS(k_lba_landmarks, gL, dim3(LBA_CT), LBA_CT * 18 * 8, st, A);
    if (out->Hpp || out->bp) LBA_LAUNCH_KS(k_lba_poses, gP, dim3(POSES_NT), (POSES_NT / 64) * 27 * 8, st, A);
    return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP;
}

extern "C" int lba_build_system(const lba_problem* prob, int batch, const lba_system* out, void* stream) {
    return lba_build_system_impl(prob, batch, out, 0, stream);
}

extern "C" int lba_build_system_hint(const lba_problem* prob, int batch, const lba_system* out, unsigned hints, void* stream) {
    if (hints & ~(unsigned)(LBA_HINT_MONO_PINHOLE | LBA_HINT_PINHOLE)) return ORB_E_INVALID;
    return lba_build_system_impl(prob, batch, out, (hints & LBA_HINT_MONO_PINHOLE) ? 1 : (hints & LBA_HINT_PINHOLE) ? 2 : 0, stream);
}

extern "C" int lba_compute_errors(const lba_problem* prob, int batch, const lba_system* out, void* stream) {
    int rc = lba_check(prob, batch, out);
    if (rc != ORB_OK) return rc;
    LbaArgs A;
    A.P = *prob; A.S = *out;
    hipStream_t st = (hipStream_t)stream;
    if (out->robust_chi2_sum && hipMemsetAsync(out->robust_chi2_sum, 0, (size_t)batch * 8, st) != hipSuccess) return ORB_E_HIP;
    hipLaunchKernelGGL(k_lba_errors, dim3((prob->cap_e + 255) / 256, batch), dim3(256), 256 * 8, st, A);
    return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP;
}

// ============================================================================================================
// Levenberg-Marquardt on the GPU (one SparseOptimizer::optimize(iterations) call per window, batched).
// The LM control state lives in device memory per window; the host only launches the fixed kernel sequence of a trial and
// polls one "any window needs another trial" word, exactly where g2o polls terminate().
//   k_lm_maxdiag     computeLambdaInit                       optimization_algorithm_levenberg.cpp:171-186
//   k_lm_begin       per-iteration bookkeeping               :85-105
//   k_lm_dinv        (Hll + lambda I)^-1, Dinv*b_l           block_solver.hpp:381-397 (+ setLambda :564-589)
//   k_lm_schur_rows  Schur complement, one workgroup per block row of Hschur (+ the coefficient rows)   :398-432
//   k_lm_chol        dense in-place Cholesky + two triangular solves, one workgroup per window   (linear_solver_eigen.h:94-123)
//   k_lm_backsub     x_l = Dinv (b_l - Hpl^T x_p), X += x_l, scale partials            block_solver.hpp:461-481, levenberg.cpp:188-195
//   k_lm_update_pose T <- exp(x_p) * T                       types_six_dof_expmap.h:73-76, se3quat.h:223-256
//   k_lm_sum_decide  chi2 sum + rho test, lambda update, push/pop   optimization_algorithm_levenberg.cpp:126-149
//   k_lm_restore     pop (restore the backup state of rejected windows)
//   k_lm_end         "Raul" stop rule                        :151-165
// ============================================================================================================
struct LmState {
    double lambda, ni, currentChi, iniChi, tempChi, rho, scale, maxDiag;
    int32_t qmax, nBad, iter, active, needTrial, ok, trials, accepted;
};
struct LmArgs {
    lba_problem P; lba_system S;
    double* poses; double* points;          // == P.poses / P.points (mutable)
    double* posesBak; double* pointsBak;
    double* Dinv; double* db;               // [cap_l][9], [cap_l][3]
    double* Hs; double* xp; double* xl;     // [np6][np6], [np6], [cap_l*3]
    double* panExt;                         // [np6][CH_LD] per window: Cholesky panel of systems too large for LDS, else nullptr
    double* cholL; double* cholY;           // [np6][np6], [np6] per window: L and y = L^-1 b of the one-launch-per-panel factorisation (k_lm_chol_step), else nullptr
    double* schurPart; int schurG;          // [batch][np6 / 6][schurG][(np6 / 12 + 1) * 36 + 6]: partial rows of the reduced system when a row is split over schurG workgroups (few windows per call), else nullptr / 1
    double* cholX;                          // [ceil(np6 / 32)][32][32] per window: the inverses of L's diagonal blocks (row-major), for its backward substitution
    double* part;                           // [batch][nPart] partial sums (chi2 / scale)
    LmState* st; int* flag; int nPart, np6;
    int shard;                              // lba_optimize_sharded: 0 = the whole window is here; 1 = this rank holds a landmark shard and OWNS the pose-side terms
                                            // (Hpp + lambda I, b_p, the pose part of computeScale enter the all-reduced sums once); 2 = a shard, not the owner
    double* red;                            // [batch][4] packed per-window scalars for the cross-rank reductions of the sharded form
    int4* rowMeta;                          // [batch][cap_e] per entry of the pose-major edge lists: (edge, landmark, first / end edge of the landmark's run)
    int32_t* edgeH;                         // [batch][cap_e + 8] Hessian index of each edge's pose (-1: fixed), landmark-major like the edges
};

// *flag: bit 0 = some edge is neither monocular nor stereo, or its camera is not a pinhole (generic kernels); bit 1 = some edge is stereo
// (else: the monocular-pinhole kernels, see edge_linearize)
static __global__ __launch_bounds__(256) void k_lm_kinds(LmArgs A, int* flag) {
    const lba_problem& P = A.P;
    const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= min(P.n_edges[b], P.cap_e)) return;
    const lba_edge E = P.edges[(size_t)b * P.cap_e + e];
    const int f = ((E.kind != LBA_EDGE_MONO && E.kind != LBA_EDGE_STEREO) || P.cameras[E.cam].model != LBA_CAM_PINHOLE ? 1 : 0) | (E.kind == LBA_EDGE_STEREO ? 2 : 0);
    if (f) atomicOr(flag, f);
}

static __global__ void k_lm_init(LmArgs A, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    LmState s;
    s.lambda = -1; s.ni = 2; s.currentChi = 0; s.iniChi = 0; s.tempChi = 0; s.rho = 0; s.scale = 0; s.maxDiag = 0;
    s.qmax = 0; s.nBad = 0; s.iter = 0; s.active = 1; s.needTrial = 0; s.ok = 1; s.trials = 0; s.accepted = 0;
    A.st[b] = s;
}

// deterministic second stage of the chi2 / scale reductions: one wave per window, lanes stride over the block partials,
// fixed butterfly
static __global__ __launch_bounds__(64) void k_lm_sum_partials(LmArgs A, int batch, int n, int what, int extra) {
    const int b = blockIdx.x, lane = threadIdx.x;
    double s = 0;
    for (int i = lane; i < n; i += 64) s += A.part[(size_t)b * A.nPart + i];
    if (lane == 0 && extra >= 0) s += A.part[(size_t)b * A.nPart + extra];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) { if (what == 0) A.st[b].tempChi = s; else A.st[b].scale = s; }
}

// computeActiveErrors + per-block partial sums of rho[0] (blocks of 256 edges)
template <int KS>
static __global__ __launch_bounds__(256) void k_lm_errors(LmArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    double* red = (double*)orb_smem;
    const lba_problem& P = A.P;
    const int b = blockIdx.y, ei = blockIdx.x * 256 + threadIdx.x;
    const int ne = min(P.n_edges[b], P.cap_e);
    double r0 = 0;
    if (ei < ne && A.st[b].active) {
        const lba_edge E = P.edges[(size_t)b * P.cap_e + ei];
        const SE3 T = load_pose(A.poses + ((size_t)b * P.cap_p + E.pose) * 7);
        const double* Xp = A.points + ((size_t)b * P.cap_l + E.point) * 3;
        const double X[3] = {Xp[0], Xp[1], Xp[2]};
        Lin L;
        edge_linearize<false, KS>(E, T, X, P.cameras[E.cam], P.huber_mono, P.huber_stereo, L);
        r0 = L.rho0;
    }
    red[threadIdx.x] = r0;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) A.part[(size_t)b * A.nPart + blockIdx.x] = red[0];
}

// grid (MAXDIAG_G, batch): the diagonal scan of a window spread over MAXDIAG_G workgroups (one workgroup walked 60 000 entries of a C5 window in
// 70 us — 3 % of a lone window's optimize(5)); the maximum of non-negative doubles is the maximum of their bit patterns, so the slices meet in one
// integer atomicMax on st.maxDiag (k_lm_init zeroes it): exact and order-independent
#define MAXDIAG_G 16
static __global__ __launch_bounds__(256) void k_lm_maxdiag(LmArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    double* red = (double*)orb_smem;
    const lba_problem& P = A.P;
    const int b = blockIdx.y, tid = threadIdx.x, t0 = (int)blockIdx.x * 256 + tid;
    const int np = min(P.n_poses[b], P.cap_p), nl = min(P.n_points[b], P.cap_l);
    double m = 0;
    for (int i = t0; i < np * 6; i += 256 * MAXDIAG_G) m = fmax(m, fabs(A.S.Hpp[((size_t)b * P.cap_p + i / 6) * 36 + (i % 6) * 7]));   // fixed blocks are zero
    for (int i = t0; i < nl * 3; i += 256 * MAXDIAG_G) m = fmax(m, fabs(A.S.Hll[((size_t)b * P.cap_l + i / 3) * 9 + (i % 3) * 4]));
    red[tid] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] = fmax(red[tid], red[tid + off]);
        __syncthreads();
    }
    if (tid == 0 && red[0] > 0) atomicMax((unsigned long long*)&A.st[b].maxDiag, (unsigned long long)__double_as_longlong(red[0]));
}

static __global__ void k_lm_begin(LmArgs A, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    LmState& s = A.st[b];
    if (!s.active) { s.needTrial = 0; return; }
    if (s.iter == 0) s.currentChi = s.tempChi;      // activeRobustChi2 of the state the system was built at (later iterations: see the caller)
    s.iniChi = s.currentChi;
    if (s.iter == 0) { s.lambda = 1e-50 * s.maxDiag; s.ni = 2; s.nBad = 0; }
    s.rho = 0; s.qmax = 0; s.needTrial = 1;
}

static __device__ __forceinline__ bool inv3_sym(const double* D, double* o) {   // column-major 3x3, cofactor inverse
    const double a = D[0], b = D[3], c = D[6], d = D[1], e = D[4], f = D[7], g = D[2], h = D[5], i = D[8];
    const double Ac = e * i - f * h, Bc = -(d * i - f * g), Cc = d * h - e * g;
    const double det = a * Ac + b * Bc + c * Cc;
    const double id = 1.0 / det;
    o[0] = Ac * id; o[3] = -(b * i - c * h) * id; o[6] = (b * f - c * e) * id;
    o[1] = Bc * id; o[4] = (a * i - c * g) * id; o[7] = -(a * f - c * d) * id;
    o[2] = Cc * id; o[5] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
    return det != 0.0 && det == det && fabs(det) < 1.7e308;
}

static __global__ __launch_bounds__(256) void k_lm_dinv(LmArgs A) {
    const lba_problem& P = A.P;
    const int b = blockIdx.y, l = blockIdx.x * 256 + threadIdx.x;
    // the "another trial is needed" word of this trial (k_lm_sum_decide adds to it, the host reads it after k_lm_restore): cleared by the trial's first
    // kernel instead of a memset launch of its own — a lone window's trial is ~30 launches of a few microseconds, every one of them counts
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *A.flag = 0;
    if (!A.st[b].needTrial) return;
    const int nl = min(P.n_points[b], P.cap_l);
    if (l >= nl) return;
    const double lam = A.st[b].lambda;
    const double* H = A.S.Hll + ((size_t)b * P.cap_l + l) * 9;
    double D[9], o[9];
#pragma unroll
    for (int k = 0; k < 9; k++) D[k] = H[k] + ((k % 4 == 0) ? lam : 0.0);
    if (!inv3_sym(D, o)) A.st[b].ok = 0;
    double* out = A.Dinv + ((size_t)b * P.cap_l + l) * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) out[k] = o[k];
    const double* bl = A.S.bl + ((size_t)b * P.cap_l + l) * 3;
    double* db = A.db + ((size_t)b * P.cap_l + l) * 3;
#pragma unroll
    for (int r = 0; r < 3; r++) db[r] = o[r] * bl[0] + o[3 + r] * bl[1] + o[6 + r] * bl[2];
}

// Structure the Schur rows walk, built once per lba_optimize call (constant over iterations and lambda trials): the walk then needs no
// dependent index loads (pose list -> edge -> landmark -> run bounds; run entry -> edge -> pose -> Hessian index).
static __global__ __launch_bounds__(256) void k_lm_rowmeta(LmArgs A) {
    const lba_problem& P = A.P;
    const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    const int ne = min(P.n_edges[b], P.cap_e), np = min(P.n_poses[b], P.cap_p);
    if (k >= P.cap_e + 8) return;
    int h = -1;
    if (k < ne) { const int p = P.edges[(size_t)b * P.cap_e + k].pose; if (p >= 0 && p < np) h = P.pose_hidx[(size_t)b * P.cap_p + p]; }
    A.edgeH[(size_t)b * (P.cap_e + 8) + k] = h;
    if (k < ne) {
        const int e = P.pose_edges[(size_t)b * P.cap_e + k];
        const int l = P.edges[(size_t)b * P.cap_e + e].point;
        const int32_t* lms = P.lm_start + (size_t)b * (P.cap_l + 1);
        A.rowMeta[(size_t)b * P.cap_e + k] = make_int4(e, l, lms[l], min(lms[l + 1], ne));
    }
}

// Schur complement, one workgroup per free pose h1.  Edges are stored landmark-major, so the edges a pose's edge e1 pairs with are the
// contiguous run of its landmark: a lane takes one edge of the pose, forms B_i Dinv once (block_solver.hpp:404) and walks the run; every
// partner edge of a free pose h2 contributes  -B_i Dinv B_j^T  to block (h1, h2) (block_solver.hpp:398-432).  Of the symmetric matrix
// workgroup h1 produces the blocks at cyclic distance d = (h1 - h2) mod n <= n/2 — the lower-triangle ones directly, the others transposed —
// so every workgroup of a window has the same amount of work (a triangular split would give the last pose n times the work of the first).
// The workgroup's blocks live in LDS ([d][36], padded to 37) and take the products as ds_add_f64 — no co-visibility lists, no landmark x pose
// table, each B_i Dinv formed once per edge instead of once per block.  No two waves add into the same LDS copy: a block receives its
// products in a wave's program order, and runs are bit-identical (tools/dbg_lm_determinism.py: 8 windows x 4 runs on MI355X; with several
// waves sharing one copy the cross-wave order of the atomics, and with it the last bits of the sums, varied from run to run).  The diagonal block
// also takes Hpp + lambda I (_Hpp->add(_Hschur) + setLambda) and the row's _bschur entries  b_p - sum_e B_i (Dinv b_l).
// Rows longer than `rowCap` blocks (LDS) are produced in column chunks, walking the pose's edges once per chunk.
#define SCH_LD 37
#ifndef LM_SCHUR_ROWCAP
#define LM_SCHUR_ROWCAP 384   // blocks of a row held in LDS at once (384 x 37 doubles = 111 KiB); tests build with a tiny value to cover the chunking
#endif
// NW waves per row, each with its OWN copy of the row's blocks in LDS (so a block still receives a wave's products in program order, and the
// copies are added in wave order at write-out: bit-reproducible); the same LDS per wave whatever NW is, but a row is finished NW times
// sooner — what a single LocalMapping call (20 rows in flight on a machine that holds 3 000 waves) needs.
static inline size_t lm_schur_smem_bytes(int rowCap, int nw) { return ((size_t)nw * rowCap * SCH_LD + nw * 6) * sizeof(double); }
template <int NW>
static __global__ __launch_bounds__(64 * NW) void k_lm_schur_rows(LmArgs A, int rowCap, int batch, const int32_t* nfreeArr) {
    constexpr int SCH_NT = 64 * NW;
    // A.schurG > 1 (few windows per call: fewer rows than compute units): blockIdx.y = g takes every schurG-th slice of the row's edges and leaves
    // its sums — blocks and right-hand-side terms — in A.schurPart; k_lm_schur_combine adds the slices in g order (a fixed order: reproducible)
    // and does the write-out.  The walk is bound by one compute unit's LDS atomics and L1: a lone window's 80 rows then use 240 units instead of 80.
    const int G = A.schurG, g = (int)blockIdx.y;
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* Sall = (double*)orb_smem;                 // [NW][rowCap][SCH_LD]
    double* Srow = Sall + (size_t)wave * rowCap * SCH_LD;   // this wave's copy
    double* coefw = Sall + (size_t)NW * rowCap * SCH_LD;    // [NW][6]
    const lba_problem& P = A.P;
    // Eight windows or more: all rows of a window run on ONE XCD, back to back (workgroup w is dispatched to XCD w % 8) — the rows walk the same
    // landmark-major edge array in the same direction, so a run of B_j blocks fetched for one row is still in that XCD's L2 when the window's
    // other rows want it.  Fewer windows (the grid is then batch x cap_p wide): rows go round the XCDs — a lone window pinned to one XCD would
    // run its 80 rows on 32 of the 256 compute units.
    int b, i1;
    if (batch >= 8) {
        const int slot = (int)blockIdx.x >> 3;
        b = (slot / P.cap_p) * 8 + ((int)blockIdx.x & 7);
        i1 = P.cap_p - 1 - slot % P.cap_p;            // late (long) rows first
    } else {
        b = (int)blockIdx.x / P.cap_p;
        i1 = P.cap_p - 1 - (int)blockIdx.x % P.cap_p;
    }
    if (b >= batch) return;
    if (!A.st[b].needTrial) return;
    const int np = min(P.n_poses[b], P.cap_p), ne = min(P.n_edges[b], P.cap_e);
    if (i1 >= np) return;
    const int32_t* hidx = P.pose_hidx + (size_t)b * P.cap_p;
    const int h1 = hidx[i1];
    if (h1 < 0) return;
    const int4* meta = A.rowMeta + (size_t)b * P.cap_e;
    const int32_t* eh = A.edgeH + (size_t)b * (P.cap_e + 8);
    const double* Hpl = A.S.Hpl + (size_t)b * P.cap_e * 18;
    const double* Dinv = A.Dinv + (size_t)b * P.cap_l * 9;
    const int s0 = P.pose_start[(size_t)b * (P.cap_p + 1) + i1], s1 = min(P.pose_start[(size_t)b * (P.cap_p + 1) + i1 + 1], ne);
    const int np6 = A.np6;
    double* Hs = A.Hs + (size_t)b * np6 * np6;
    double coef[6];
#pragma unroll
    for (int k = 0; k < 6; k++) coef[k] = 0;
    // every edge pairs with itself (a third of all pairs): those products — the bulk of the diagonal block, symmetric in sum — stay in
    // registers (lower triangle) and are reduced once per workgroup; LDS atomics take the pairs of different edges only
    double dacc[21];
#pragma unroll
    for (int k = 0; k < 21; k++) dacc[k] = 0;
    // partner h2 belongs to this workgroup iff d = (h1 - h2) mod n <= (n - 1) / 2, or (n even) d == n / 2 and h1 < h2
    const int n = nfreeArr[b], dlo = (n - 1) >> 1, dtie = (n & 1) ? -1 : (n >> 1);
    const int nslot = (n >> 1) + 1;                   // slots d = 0 .. n / 2
    for (int c0 = 0; c0 < nslot; c0 += rowCap) {
        const int nblk = min(rowCap, nslot - c0);
        for (int t = lane; t < nblk * SCH_LD; t += 64) Srow[t] = 0.0;
        __syncthreads();
        for (int k = s0 + g * SCH_NT + tid; k < s1; k += G * SCH_NT) {
            const int4 mt = meta[k];
            const int e1 = mt.x, l = mt.y;
            double Bi[18];
            {   // a 6 x 3 block is 144 bytes on a 16-byte boundary: nine 16-byte loads (every lane reads its own block: a load instruction touches 64
                // cache lines whatever its width, and the walk is bound by exactly that)
                const double2* Bg = (const double2*)(Hpl + (size_t)e1 * 18);
#pragma unroll
                for (int q = 0; q < 9; q++) { const double2 t2 = Bg[q]; Bi[2 * q] = t2.x; Bi[2 * q + 1] = t2.y; }
            }
            const double* Di = Dinv + (size_t)l * 9;
            double BDi[18];     // B_i * Dinv, 6x3 column-major
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int r = 0; r < 6; r++) BDi[c * 6 + r] = Bi[r] * Di[c * 3] + Bi[6 + r] * Di[c * 3 + 1] + Bi[12 + r] * Di[c * 3 + 2];
            if (c0 == 0) {
                const double* db = A.db + ((size_t)b * P.cap_l + l) * 3;
#pragma unroll
                for (int r = 0; r < 6; r++) coef[r] += Bi[r] * db[0] + Bi[6 + r] * db[1] + Bi[12 + r] * db[2];
                int t = 0;
#pragma unroll
                for (int c = 0; c < 6; c++)
#pragma unroll
                    for (int r = c; r < 6; r++) dacc[t++] -= BDi[r] * Bi[c] + BDi[6 + r] * Bi[6 + c] + BDi[12 + r] * Bi[12 + c];
            }
            for (int e4 = mt.z; e4 < mt.w; e4 += 4) {
                int hh[4];
                __builtin_memcpy(hh, eh + e4, 16);    // four run entries per load (the array is padded past its end)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                const int e2 = e4 + j, h2 = hh[j];
                if (e2 >= mt.w || h2 < 0 || e2 == e1) continue;   // end of the run, fixed pose, the edge itself (registers, above)
                int d = h1 - h2;
                d += d < 0 ? n : 0;
                if (d > dlo && !(d == dtie && h1 < h2)) continue;   // the block belongs to workgroup h2
                d -= c0;
                if (d < 0 || d >= nblk) continue;     // another column chunk
                double Bj[18];
                {
                    const double2* Bg = (const double2*)(Hpl + (size_t)e2 * 18);
#pragma unroll
                    for (int q = 0; q < 9; q++) { const double2 t2 = Bg[q]; Bj[2 * q] = t2.x; Bj[2 * q + 1] = t2.y; }
                }
                double* blk = Srow + d * SCH_LD;
#pragma unroll
                for (int c = 0; c < 6; c++)
#pragma unroll
                    for (int r = 0; r < 6; r++) atomicAdd(blk + c * 6 + r, -(BDi[r] * Bj[c] + BDi[6 + r] * Bj[6 + c] + BDi[12 + r] * Bj[12 + c]));
                }
            }
        }
        if (c0 == 0) {   // the self-pair sums: wave butterfly, then one lane per wave adds them to slot 0 (both triangles)
#pragma unroll
            for (int t = 0; t < 21; t++)
                for (int off = 32; off > 0; off >>= 1) dacc[t] += __shfl_xor(dacc[t], off);
            if (lane == 0) {
                int t = 0;
#pragma unroll
                for (int c = 0; c < 6; c++)
#pragma unroll
                    for (int r = c; r < 6; r++) { atomicAdd(Srow + c * 6 + r, dacc[t]); if (r != c) atomicAdd(Srow + r * 6 + c, dacc[t]); t++; }
            }
        }
        __syncthreads();
        if (G > 1) {                                              // (one chunk: the host splits rows only when a row fits LDS whole)
            double* part = A.schurPart + (((size_t)b * (np6 / 6) + h1) * G + g) * ((size_t)(np6 / 12 + 1) * 36 + 6);
            for (int t = tid; t < nblk * 36; t += SCH_NT) {
                const int q = t / 36, k = t - q * 36;
                double v = Sall[q * SCH_LD + k];
#pragma unroll
                for (int w = 1; w < NW; w++) v += Sall[(size_t)w * rowCap * SCH_LD + q * SCH_LD + k];
                part[t] = v;
            }
            __syncthreads();
            continue;
        }
        for (int t = tid; t < nblk * 36; t += SCH_NT) {
            const int q = t / 36, k = t - q * 36, d = c0 + q;
            if (d == dtie && 2 * h1 >= n) continue;               // the tie slot belongs to the lower pose of the antipodal pair
            const int h2 = h1 - d + (h1 < d ? n : 0);
            double v = Sall[q * SCH_LD + k];
#pragma unroll
            for (int w = 1; w < NW; w++) v += Sall[(size_t)w * rowCap * SCH_LD + q * SCH_LD + k];   // the waves' copies in wave order
            if (d == 0 && A.shard != 2) v += A.S.Hpp[((size_t)b * P.cap_p + h1) * 36 + k] + ((k % 7 == 0) ? A.st[b].lambda : 0.0);
            const int c = k / 6, r = k - c * 6;                   // entry (r, c) of block (h1, h2)
            if (h2 <= h1) Hs[(size_t)(h2 * 6 + c) * np6 + h1 * 6 + r] = v;     // lower triangle: row block h1, column block h2
            else Hs[(size_t)(h1 * 6 + r) * np6 + h2 * 6 + c] = v;              // transposed into block (h2, h1)
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 6; k++)
        for (int off = 32; off > 0; off >>= 1) coef[k] += __shfl_xor(coef[k], off);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; k++) coefw[wave * 6 + k] = coef[k];
    }
    __syncthreads();
    if (tid < 6) {
        double c = 0;
        for (int w = 0; w < SCH_NT / 64; w++) c += coefw[w * 6 + tid];
        if (G > 1) A.schurPart[(((size_t)b * (np6 / 6) + h1) * G + g) * ((size_t)(np6 / 12 + 1) * 36 + 6) + (size_t)(np6 / 12 + 1) * 36 + tid] = c;
        else A.xp[(size_t)b * np6 + h1 * 6 + tid] = (A.shard != 2 ? A.S.bp[((size_t)b * P.cap_p + h1) * 6 + tid] : 0.0) - c;
    }
}

// the write-out of k_lm_schur_rows for rows that were split over A.schurG workgroups: slices added in g order, then Hpp + lambda I on the diagonal
// block, the lower / transposed placement and the right-hand side exactly as there.  One workgroup per (free pose, window).
static __global__ __launch_bounds__(256) void k_lm_schur_combine(LmArgs A, const int32_t* nfreeArr) {
    const lba_problem& P = A.P;
    const int b = blockIdx.y, h1 = blockIdx.x, tid = threadIdx.x;
    if (!A.st[b].needTrial) return;
    const int n = nfreeArr[b], np6 = A.np6, G = A.schurG;
    if (h1 >= n) return;
    const int dtie = (n & 1) ? -1 : (n >> 1), nslot = (n >> 1) + 1;
    const size_t rowD = (size_t)(np6 / 12 + 1) * 36 + 6;
    const double* part = A.schurPart + ((size_t)b * (np6 / 6) + h1) * G * rowD;
    double* Hs = A.Hs + (size_t)b * np6 * np6;
    for (int t = tid; t < nslot * 36; t += 256) {
        const int d = t / 36, k = t - d * 36;
        if (d == dtie && 2 * h1 >= n) continue;
        const int h2 = h1 - d + (h1 < d ? n : 0);
        double v = part[t];
        for (int gg = 1; gg < G; gg++) v += part[(size_t)gg * rowD + t];
        if (d == 0 && A.shard != 2) v += A.S.Hpp[((size_t)b * P.cap_p + h1) * 36 + k] + ((k % 7 == 0) ? A.st[b].lambda : 0.0);
        const int c = k / 6, r = k - c * 6;
        if (h2 <= h1) Hs[(size_t)(h2 * 6 + c) * np6 + h1 * 6 + r] = v;
        else Hs[(size_t)(h1 * 6 + r) * np6 + h2 * 6 + c] = v;
    }
    if (tid < 6) {
        double c = 0;
        for (int gg = 0; gg < G; gg++) c += part[(size_t)gg * rowD + (size_t)(np6 / 12 + 1) * 36 + tid];
        A.xp[(size_t)b * np6 + h1 * 6 + tid] = (A.shard != 2 ? A.S.bp[((size_t)b * P.cap_p + h1) * 6 + tid] : 0.0) - c;
    }
}

#include "dense_chol.inc"
#ifndef LM_CHOL_SPLIT_MAX_BATCH
#define LM_CHOL_SPLIT_MAX_BATCH 48    // windows per call up to which one launch per panel (k_lm_chol_step) beats one workgroup per window (MI355X, ms per
                                      // optimize(5) of 100-key-frame windows, per panel / per window: 1 window 2.2 / 6.8, 8: 4.0 / 7.5, 32: 10.6 / 12.7, 64: 18.8 / 19.1,
                                      // 128: 35.6 / 31.7, 256: 67.7 / 55.8; 40-key-frame windows: 8: 1.68 / 2.17, 64: 5.4 / 5.7, 256: 16.8 / 16.4 — a tie at 64, where
                                      // stereo / fisheye windows came out 2-4 % behind: the switch sits below it)
#endif
#ifndef LM_CHOL_NT
#define LM_CHOL_NT 512    // 8 waves per window: the factorisation is one workgroup per window, its trailing update a global-memory latency problem
#endif
template <int NB>
static __global__ __launch_bounds__(LM_CHOL_NT) void k_lm_chol(LmArgs A, const int32_t* nfreeArr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int b = blockIdx.x;
    if (!A.st[b].needTrial) return;
    const int n = nfreeArr[b] * 6, ld = A.np6;
    if (!wg_chol_solve<LM_CHOL_NT, NB>(A.Hs + (size_t)b * ld * ld, n, ld, A.xp + (size_t)b * ld, orb_smem, A.panExt ? A.panExt + (size_t)b * ld * (NB + 1) : nullptr) &&
        threadIdx.x == 0)
        A.st[b].ok = 0;
}

#define LM_CHOLS_NB 32
// Few windows per call: the factorisation as one launch per 32-column panel (round 5; round 4 ran a panel launch and an update launch per panel).
// A WAVE per 16 x 16 tile of the trailing triangle does everything its tile needs by itself, so that nothing of a panel step waits on another launch:
//   1. the 32 x 32 diagonal block A11 = L11 L11^T in registers (lane r < 32 = row r) and, in the SAME instruction stream on lanes 32..63, the
//      inverse X = L11^-1 by forward substitution of the identity (lane 32 + j = column j): step c scales entry c by 1 / L(c, c) and takes
//      L(c2, c) * entry c off every later entry c2 — for a row of A that is the right-looking Cholesky update, for a column of X the substitution.
//      Column c of L reaches all lanes through LDS (one ds_write, broadcast ds_reads) instead of two v_readlane per entry;
//   2. L21 rows of the tile's two row blocks as PRODUCTS on the fp64 matrix core, L21 = A21 X^T (v_mfma_f64_16x16x4_f64; the 496 dependent
//      steps per row of a triangular solve are gone), re-laid out through LDS;
//   3. the tile's rank-32 update C -= L21_i L21_j^T (8 more matrix instructions).
// Every wave factors the same block (identical values).  L and y = L^-1 b go to cholL / cholY, not back into Hs / xp: other waves of the launch
// still read the unfactored panel.  Waves of tile column 0 write their rows of L21 (row-major: what k_lm_chol_back_x reads) and take those rows'
// share of the forward substitution; wave 0 writes X and the block's y (L11 itself is not stored: the backward substitution uses X).  The explicit
// inverse costs accuracy cond(L11) * eps on the panel rows instead of eps — a 32 x 32 block of a damped reduced camera system; measured against the
// one-workgroup kernel in test_hip_cholesky_per_phase_launches_agree_with_one_workgroup and over 1 200 random windows by tools/fuzz_lm.py.
#define LM_CHOLF_SMEM ((LM_CHOLS_NB * (LM_CHOLS_NB + 1) + 2 * 16 * (LM_CHOLS_NB + 1) + 2 * LM_CHOLS_NB + 2 * LM_CHOLS_NB) * 8)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIP_EMULATED)
#define LM_KEEP_LOADED(x) asm volatile("" : "+v"(x))     // the value exists in a (vector) register HERE: its load cannot be sunk below this point
#ifdef CHOL_PROF
#define LM_KEEP_LOADED_S(x) do {} while (0)              // (the timers' clock reads move the window state into vector registers: nothing to pin)
#else
#define LM_KEEP_LOADED_S(x) asm volatile("" : "+s"(x))   // the same for a wave-uniform value (scalar register)
#endif
#else
#define LM_KEEP_LOADED(x) do {} while (0)
#define LM_KEEP_LOADED_S(x) do {} while (0)
#endif
#define LM_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
static __global__ __launch_bounds__(64) void k_lm_chol_step(LmArgs A, const int32_t* nfreeArr, int kb) {
    constexpr int NB = LM_CHOLS_NB, LD = NB + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    double* Xs = (double*)orb_smem;        // [NB][LD] X = L11^-1, row-major (zero above the diagonal)
    double* Li = Xs + NB * LD;             // [16][LD] L21 rows of the tile's row block
    double* Lj = Li + 16 * LD;             // [16][LD] L21 rows of the tile's column block
    double* col = Lj + 16 * LD;            // [2][NB]  column c of L11 (double buffered over the steps), later y of the block
    double* bs = col + 2 * NB;             // [NB]     right-hand side of the block
    const int b = blockIdx.y, lane = threadIdx.x, t = blockIdx.x;
    const int ld = A.np6;                  // (kb < ld: the host launches no panel beyond the largest system of the batch)
    double* S = A.Hs + (size_t)b * ld * ld;
    double* x = A.xp + (size_t)b * ld;
    const int ml = lane & 15, kq = lane >> 4, j32 = lane & 31;
    CHP_DECL;
    // ---- every operand this wave will need, requested before anything is waited for.  The launch is a chain of memory round trips otherwise
    // (kernel arguments -> window state -> operands): the addresses below depend on the launch geometry only — rows and columns clamped to the
    // ALLOCATED matrix (ld), not to the window's own size n, which is still in flight — and the guards select afterwards.  (A guarded load
    // compiles to a branch with its own s_waitcnt vmcnt(0): the 32 loads of the block alone were 32 serialised round trips, half of the kernel.)
    int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);                      // tile t of the trailing triangle (row-major over its lower half)
    while (ti * (ti + 1) / 2 > t) ti--;
    while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
    const int tj = t - ti * (ti + 1) / 2;
    const int i0 = NB + ti * 16, j0 = NB + tj * 16;   // panel-relative first row of the tile's two row blocks
    const int rmax = ld - kb - 1;                     // last panel-relative row / column that exists in memory
    double sv[NB], ai[NB / 4], aj[NB / 4], cv[4];
    {
        const double* Sb = S + (size_t)kb * ld + kb + min(j32, rmax);      // the diagonal block: row j32 (lanes 32..63 load it too and do not use it)
#pragma unroll
        for (int c = 0; c < NB; c++) sv[c] = Sb[(size_t)min(c, rmax) * ld];
        const int ri = min(i0 + ml, rmax), rj = min(j0 + ml, rmax);
#pragma unroll
        for (int kk = 0; kk < NB / 4; kk++) {
            const double* Sk = S + (size_t)(kb + min(4 * kk + kq, rmax)) * ld + kb;
            ai[kk] = Sk[ri]; aj[kk] = Sk[rj];
        }
        const double* Sc0 = S + (size_t)(kb + rj) * ld + kb;
#pragma unroll
        for (int q = 0; q < 4; q++) cv[q] = Sc0[min(i0 + kq + 4 * q, rmax)];
    }
    const double xb = x[kb + min(j32, rmax)];
    int needTrial = A.st[b].needTrial, okWin = A.st[b].ok, nfreeB = nfreeArr[b];         // three more loads in flight, one wait
    LM_KEEP_LOADED_S(needTrial); LM_KEEP_LOADED_S(okWin); LM_KEEP_LOADED_S(nfreeB);
#pragma unroll
    for (int c = 0; c < NB; c++) LM_KEEP_LOADED(sv[c]);   // (or the compiler sinks every load below the early exits and into the branch that uses it: one wait per load again)
#pragma unroll
    for (int kk = 0; kk < NB / 4; kk++) { LM_KEEP_LOADED(ai[kk]); LM_KEEP_LOADED(aj[kk]); }
    if (!needTrial || !okWin) return;                 // a window whose earlier panel was not positive definite stays failed
    const int n = nfreeB * 6;
    if (kb >= n) return;
    const int nb = min(NB, n - kb), m = n - kb, mt = m - NB;
    const int T = mt > 0 ? (mt + 15) >> 4 : 0, ntiles = T * (T + 1) / 2;
    if (t >= max(1, ntiles)) return;
    double* LR = A.cholL + (size_t)b * ld * ld;       // L's rows under the diagonal blocks, ROW-major (what the backward substitution reads)
    double* yv = A.cholY + (size_t)b * ld;
    wg_chol_v4 ct = {0.0, 0.0, 0.0, 0.0};
    bool okc[4] = {false, false, false, false};
    double* Sc = S + (size_t)(kb + min(j0 + ml, rmax)) * ld + kb;
    if (ntiles) {
        const bool iok = i0 + ml < m, jok = j0 + ml < m;
#pragma unroll
        for (int kk = 0; kk < NB / 4; kk++) { ai[kk] = iok ? ai[kk] : 0.0; aj[kk] = jok ? aj[kk] : 0.0; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = i0 + kq + 4 * q;
            okc[q] = r < m && j0 + ml < m && r >= j0 + ml;
            ct[q] = okc[q] ? cv[q] : 0.0;
        }
    }
    // ---- 1. diagonal block and its inverse
    const bool isA = lane < NB;
    double v[NB];
#pragma unroll
    for (int c = 0; c < NB; c++) v[c] = isA ? ((lane < nb && c <= lane) ? sv[c] : (c == lane ? 1.0 : 0.0)) : (c == j32 ? 1.0 : 0.0);
    CHP_MARK(0);
    // The pivot of step c + 1 is taken from lane c + 1's OWN registers before step c's column has made its round trip through LDS: for that lane the
    // update v[c + 1] -= L(c + 1, c) * v[c] has both factors in its own v[c], so the value is bit for bit what the update below leaves there — and the
    // next pivot's rsqrt (a chain of eight dependent fp64 operations) runs while the column is written, read back and taken off the other entries.
    bool ok = true;
    double dkk = wg_chol_bcast(v[0], 0);
#ifdef HIP_EMULATED
#define LM_RSQRT(x) (1.0 / sqrt(x))
#else
#define LM_RSQRT(x) rsqrt(x)
#endif
    double ri = LM_RSQRT(dkk);
#pragma unroll
    for (int c = 0; c < NB; c++) {
        ok = ok && (dkk > 0) && (dkk < 1.7e308);
        v[c] *= ri;                                   // lane c: sqrt(dkk); row r > c: L(r, c); column j of X: X(c, j)
        if (c + 1 < NB) {
            double* cb = col + (c & 1) * NB;
            if (isA) cb[lane] = v[c];
            dkk = wg_chol_bcast(__builtin_fma(-v[c], v[c], v[c + 1]), c + 1);
            ri = LM_RSQRT(dkk);
            LM_WAVE_SYNC();
#pragma unroll
            for (int c2 = c + 1; c2 < NB; c2++) v[c2] = __builtin_fma(-cb[c2], v[c], v[c2]);   // (entries above the diagonal of A carry unused values)
        }
    }
#undef LM_RSQRT
    CHP_MARK(1);
    if (!ok) { if (t == 0 && lane == 0) A.st[b].ok = 0; return; }   // uniform over the window's waves: the same values everywhere
    if (!isA) {
#pragma unroll
        for (int i = 0; i < NB; i++) Xs[i * LD + j32] = v[i];
        if (t == 0) {                                 // X of this block, for the backward substitution (k_lm_chol_back_x)
            double* Xg = A.cholX + ((size_t)b * ((ld + NB - 1) / NB) + kb / NB) * (NB * NB);
#pragma unroll
            for (int i = 0; i < NB; i++) Xg[i * NB + j32] = v[i];
        }
    } else bs[lane] = lane < nb ? xb : 0.0;
    LM_WAVE_SYNC();
    if (tj == 0 || !ntiles) {                         // y of the block = X b (also the single wave of a panel without rows below it)
        if (isA) {
            double y = 0.0;
#pragma unroll
            for (int j = 0; j < NB; j++) y = __builtin_fma(Xs[lane * LD + j], bs[j], y);
            col[lane] = y;
            if (t == 0 && lane < nb) yv[kb + lane] = y;
        }
    }
    CHP_MARK(2);
    if (!ntiles) return;
    // ---- 2. L21 = A21 X^T for the two row blocks (nb == NB here: rows below a short last block do not exist)
#pragma unroll
    for (int side = 0; side < 2; side++) {
        if (side == 1 && ti == tj) break;             // a diagonal tile: one row block
        double* Lo = side ? Lj : Li;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            wg_chol_v4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < (h ? NB / 4 : NB / 8); kk++) {   // X(c, k) = 0 for k > c: columns c < 16 need k < 16 only
                const double bv = Xs[(16 * h + ml) * LD + 4 * kk + kq];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(side ? aj[kk] : ai[kk], bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) Lo[(kq + 4 * q) * LD + 16 * h + ml] = acc[q];
        }
    }
    LM_WAVE_SYNC();
    CHP_MARK(3);
    const double* Lb = ti == tj ? Li : Lj;
    // ---- 3. the tile: C -= L21_i L21_j^T
#pragma unroll
    for (int kk = 0; kk < NB / 4; kk++)
        ct = __builtin_amdgcn_mfma_f64_16x16x4f64(-Li[ml * LD + 4 * kk + kq], Lb[ml * LD + 4 * kk + kq], ct, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (okc[q]) Sc[i0 + kq + 4 * q] = ct[q];
    if (tj == 0) {                                    // this row block's rows of L (row-major: two rows of 32 per store) and their share of the forward substitution
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int rr = (lane >> 5) + 2 * q;
            if (i0 + rr < m) LR[(size_t)(kb + i0 + rr) * ld + kb + j32] = Li[rr * LD + j32];
        }
        const int r = i0 + lane;
        if (lane < 16 && r < m) {
            double acc = x[kb + r];
#pragma unroll
            for (int c = 0; c < NB; c++) acc = __builtin_fma(-Li[lane * LD + c], col[c], acc);
            x[kb + r] = acc;
        }
    }
    CHP_MARK(4);
#ifdef CHOL_PROF
    if (t == 0) CHP_FLUSH();
#endif
}

// L^T x = y behind k_lm_chol_step, one workgroup per window, without a dependent chain inside a block: with X = L11^-1 of every diagonal block at
// hand, x_blk = X^T w_blk is a 32 x 32 product (one wave, coalesced rows of X), and the finished block is taken off every earlier entry at once —
// thread c owns w_c and reads the 32 entries L(kb .. kb + 31, c) of its column from the ROW-major copy k_lm_chol_step leaves (consecutive threads,
// consecutive addresses; loaded before the block's x exists: the loads do not depend on it).  15 blocks of a 480-unknown system: two barriers and one memory round trip each.  LDS: 2 x ceil(n / 32) x 32 doubles.
static inline size_t lm_chol_back_x_smem(int np6) { return (size_t)(2 * ((np6 + 31) / 32 * 32) + 2 * 1024) * 8; }
static __global__ __launch_bounds__(LM_CHOL_NT) void k_lm_chol_back_x(LmArgs A, const int32_t* nfreeArr) {
    constexpr int NB = LM_CHOLS_NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (!A.st[b].needTrial || !A.st[b].ok) return;
    const int n = nfreeArr[b] * 6, ld = A.np6, n32 = (n + NB - 1) / NB * NB;
    double* w = (double*)orb_smem;         // [n32] y, then y minus the finished blocks' products
    double* xs = w + n32;                  // [n32] the solution
    double* Xl = xs + n32;                 // [2][NB * NB] X of the current block and (arriving) of the next one
    const double* LR = A.cholL + (size_t)b * ld * ld;   // rows of L under the diagonal blocks, row-major (k_lm_chol_step)
    const double* yv = A.cholY + (size_t)b * ld;
    const double* Xw = A.cholX + (size_t)b * ((ld + NB - 1) / NB) * (NB * NB);
    // Software pipeline over the blocks: while block kb is solved and taken off w, the rows of L and the X of block kb - 32 are already on their way
    // (neither depends on x) — a block costs its two barriers and 64 multiply-adds per thread, not a memory round trip.
    double lr[NB], ln[NB];
    int kb = n32 - NB;
    {
        const int nb = min(NB, n - kb);
        const double* Lc = LR + (size_t)kb * ld + min(tid, max(kb - 1, 0));   // L(kb + r, tid), r = 0 .. 31: a row of L is contiguous over the threads
#pragma unroll
        for (int r = 0; r < NB; r++) lr[r] = Lc[(size_t)min(r, nb - 1) * ld];
        const double* Xg = Xw + (size_t)(kb / NB) * (NB * NB);
        for (int i = tid; i < NB * NB; i += LM_CHOL_NT) Xl[i] = Xg[i];
        if (nb < NB) {
#pragma unroll
            for (int r = 0; r < NB; r++) lr[r] = r < nb ? lr[r] : 0.0;        // a short last block: its missing rows do not exist
        }
    }
    for (int i = tid; i < n32; i += LM_CHOL_NT) w[i] = i < n ? yv[i] : 0.0;
    __syncthreads();
    int buf = 0;
    for (; kb >= 0; kb -= NB) {
        const int kn = kb - NB;
        double xq[(NB * NB + LM_CHOL_NT - 1) / LM_CHOL_NT];
        if (kn >= 0) {                     // (full blocks from here on)
            const double* Lc = LR + (size_t)kn * ld + min(tid, max(kn - 1, 0));
#pragma unroll
            for (int r = 0; r < NB; r++) ln[r] = Lc[(size_t)r * ld];
            const double* Xg = Xw + (size_t)(kn / NB) * (NB * NB);
#pragma unroll
            for (int q = 0; q < (NB * NB + LM_CHOL_NT - 1) / LM_CHOL_NT; q++) xq[q] = Xg[min(tid + q * LM_CHOL_NT, NB * NB - 1)];
        }
        if (tid < NB) {                    // x_j = sum_{i >= j} X(i, j) w_i  (a short last block is padded with the identity, its w with zeros)
            const double* Xc = Xl + buf * (NB * NB);
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < NB; i++) acc = __builtin_fma(Xc[i * NB + tid], w[kb + i], acc);
            xs[kb + tid] = acc;
        }
        __syncthreads();
        if (tid < kb) {
            double acc = w[tid];
#pragma unroll
            for (int r = 0; r < NB; r++) acc = __builtin_fma(-lr[r], xs[kb + r], acc);
            w[tid] = acc;
        }
        if (kn >= 0) {
#pragma unroll
            for (int q = 0; q < (NB * NB + LM_CHOL_NT - 1) / LM_CHOL_NT; q++)
                if (tid + q * LM_CHOL_NT < NB * NB) Xl[(buf ^ 1) * (NB * NB) + tid + q * LM_CHOL_NT] = xq[q];
#pragma unroll
            for (int r = 0; r < NB; r++) lr[r] = ln[r];
        }
        __syncthreads();
        buf ^= 1;
    }
    double* x = A.xp + (size_t)b * ld;
    for (int i = tid; i < n; i += LM_CHOL_NT) x[i] = xs[i];
}
#ifdef CHOL_PROF
extern "C" int lba_debug_chol_prof(unsigned long long* out8, int clear) {
    static unsigned long long z[8];
    if (hipDeviceSynchronize() != hipSuccess) return ORB_E_HIP;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_cholprof), sizeof(z)) != hipSuccess) return ORB_E_HIP;
    if (clear && hipMemcpyToSymbol(HIP_SYMBOL(g_cholprof), z, sizeof(z)) != hipSuccess) return ORB_E_HIP;
    return ORB_OK;
}
#endif

static __global__ void k_lm_backup(LmArgs A, size_t nPose, size_t nPoint, int capP7, int capL3) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nPose) { if (A.st[i / capP7].needTrial) A.posesBak[i] = A.poses[i]; }
    else if (i < nPose + nPoint) { const size_t j = i - nPose; if (A.st[j / capL3].needTrial) A.pointsBak[j] = A.points[j]; }
}

// Workgroup = LBA_LB landmarks = one contiguous chunk of the landmark-major edge array: the chunk's H_pl blocks are loaded coalesced into LDS
// (16 bytes per lane, consecutive lanes consecutive addresses — a lane reading its own 144-byte block touches 64 cache lines per
// instruction), every edge thread forms B_i^T x_p from its LDS copy, the landmark threads subtract them in edge order.
#ifndef BS_CT
#define BS_CT 128     // edges per chunk = threads per workgroup of the back-substitution
#endif
#define BS_LB (BS_CT / 8)   // landmarks per workgroup
static __global__ __launch_bounds__(BS_CT) void k_lm_backsub(LmArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    double* stage = (double*)orb_smem;                // [BS_CT][18]
    double* tc = stage + BS_CT * 18;                 // [BS_CT][3]  B_i^T x_p per edge of the chunk
    double* red = tc + BS_CT * 3;                    // [BS_LB]
    const lba_problem& P = A.P;
    const int b = blockIdx.y, tid = threadIdx.x;
    const LmState st = A.st[b];
    double sc = 0;
    const int nl = min(P.n_points[b], P.cap_l), ne = min(P.n_edges[b], P.cap_e);
    const int l0 = blockIdx.x * BS_LB;
    if (st.needTrial && st.ok && l0 < nl) {           // uniform
        const int l1 = min(l0 + BS_LB, nl);
        const int32_t* lms = P.lm_start + (size_t)b * (P.cap_l + 1);
        const int eBegin = min(lms[l0], ne), eEnd = min(lms[l1], ne);
        const int myL = l0 + tid;
        int ls = 0, le = 0;
        double cl[3] = {0, 0, 0};
        const bool mine = tid < BS_LB && myL < l1;
        // everything a landmark thread needs at the end is loaded up front: a workgroup's life is a chain of dependent global loads
        double blv[3] = {0, 0, 0}, Dv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Xv[3] = {0, 0, 0};
        double* X = A.points + ((size_t)b * P.cap_l + myL) * 3;
        if (mine) {
            ls = min(lms[myL], ne); le = min(lms[myL + 1], ne);
            const double* bl = A.S.bl + ((size_t)b * P.cap_l + myL) * 3;
            const double* Di = A.Dinv + ((size_t)b * P.cap_l + myL) * 9;
#pragma unroll
            for (int r = 0; r < 3; r++) { blv[r] = bl[r]; Xv[r] = X[r]; cl[r] = blv[r]; }
#pragma unroll
            for (int r = 0; r < 9; r++) Dv[r] = Di[r];
        }
        for (int c0 = eBegin; c0 < eEnd; c0 += BS_CT) {
            const int cnt = min(eEnd, c0 + BS_CT) - c0;
            const double2* src = (const double2*)(A.S.Hpl + ((size_t)b * P.cap_e + c0) * 18);
            double2* dst = (double2*)stage;
            int h = -1;
            if (tid < cnt) h = A.edgeH[(size_t)b * (P.cap_e + 8) + c0 + tid];   // Hessian index of the edge's pose (k_lm_rowmeta); index -> x_p runs under the copy
            for (int i = tid; i < cnt * 9; i += BS_CT) dst[i] = src[i];
            double xp[6] = {0, 0, 0, 0, 0, 0};
            if (h >= 0) {
                const double* x = A.xp + (size_t)b * A.np6 + h * 6;
#pragma unroll
                for (int r = 0; r < 6; r++) xp[r] = x[r];
            }
            __syncthreads();
            if (tid < cnt) {
                const double* Bi = stage + tid * 18;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    double t = 0;
#pragma unroll
                    for (int r = 0; r < 6; r++) t += Bi[c * 6 + r] * xp[r];
                    tc[tid * 3 + c] = t;      // a fixed pose's block is zero (k_lba_landmarks) and its x_p is taken as zero
                }
            }
            __syncthreads();
            if (mine) {
                const int a0 = max(ls, c0), a1 = min(le, c0 + BS_CT);
                for (int e = a0; e < a1; e++) {
#pragma unroll
                    for (int c = 0; c < 3; c++) cl[c] -= tc[(e - c0) * 3 + c];
                }
            }
            __syncthreads();
        }
        if (mine) {
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const double xl = Dv[r] * cl[0] + Dv[3 + r] * cl[1] + Dv[6 + r] * cl[2];
                sc += xl * (st.lambda * xl + blv[r]);     // computeScale, landmark part
                X[r] = Xv[r] + xl;                        // VertexSBAPointXYZ::oplusImpl
            }
        }
    }
    if (tid < BS_LB) red[tid] = sc;
    __syncthreads();
    if (tid == 0) {
        double t = 0;
        for (int i = 0; i < BS_LB; i++) t += red[i];
        A.part[(size_t)b * A.nPart + blockIdx.x] = t;
    }
}

static __device__ __forceinline__ Quat quat_from_R(const double m[9]) {   // Eigen Quaterniond(Matrix3d), row-major in
    Quat q;
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t; t = 0.5 / t;
        q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        double c[3];
        c[i] = 0.5 * t; t = 0.5 / t;
        q.w = (m[k * 3 + j] - m[j * 3 + k]) * t; c[j] = (m[j * 3 + i] + m[i * 3 + j]) * t; c[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        q.x = c[0]; q.y = c[1]; q.z = c[2];
    }
    return q;
}

// pose update + the pose part of computeScale (one thread per pose; the partial goes to slot nPart-1 via a block of its own)
// (+ the scale sum of the rho test — the back-substitution's per-block partials and this kernel's own pose part — in k_lm_sum_partials' order: lanes of
// wave 0 stride over the nBack partials, lane 0 adds the pose part, fixed butterfly; one launch fewer per lambda trial)
static __global__ __launch_bounds__(256) void k_lm_update_pose(LmArgs A, int nBack) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    double* red = (double*)orb_smem;
    const lba_problem& P = A.P;
    const int b = blockIdx.x, tid = threadIdx.x;
    const LmState st = A.st[b];
    double sc = 0;
    if (st.needTrial && st.ok) {
        const int np = min(P.n_poses[b], P.cap_p);
        for (int i = tid; i < np; i += 256) {
            const int h = P.pose_hidx[(size_t)b * P.cap_p + i];
            if (h < 0) continue;
            const double* u = A.xp + (size_t)b * A.np6 + h * 6;
            const double* bp = A.S.bp + ((size_t)b * P.cap_p + h) * 6;
            for (int r = 0; r < 6; r++) sc += u[r] * (st.lambda * u[r] + bp[r]);
            // SE3Quat::exp(update) (se3quat.h:223-256)
            const double om0 = u[0], om1 = u[1], om2 = u[2];
            const double theta = sqrt(om0 * om0 + om1 * om1 + om2 * om2);
            const double O[9] = {0, -om2, om1, om2, 0, -om0, -om1, om0, 0};
            double O2[9], Rm[9], V[9];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) O2[r * 3 + c] = O[r * 3] * O[c] + O[r * 3 + 1] * O[3 + c] + O[r * 3 + 2] * O[6 + c];
            for (int k = 0; k < 9; k++) {
                const double I = (k % 4 == 0) ? 1.0 : 0.0;
                if (theta < 0.00001) { Rm[k] = I + O[k] + O2[k]; V[k] = Rm[k]; }
                else {
                    Rm[k] = I + sin(theta) / theta * O[k] + (1 - cos(theta)) / (theta * theta) * O2[k];
                    V[k] = I + (1 - cos(theta)) / (theta * theta) * O[k] + (theta - sin(theta)) / (theta * theta * theta) * O2[k];
                }
            }
            SE3 D;
            D.r = quat_from_R(Rm);
            quat_normalize(D.r);
            for (int r = 0; r < 3; r++) D.t[r] = V[r * 3] * u[3] + V[r * 3 + 1] * u[4] + V[r * 3 + 2] * u[5];
            double* p = A.poses + ((size_t)b * P.cap_p + i) * 7;
            const SE3 Tn = se3_mul(D, load_pose(p));
            p[0] = Tn.t[0]; p[1] = Tn.t[1]; p[2] = Tn.t[2]; p[3] = Tn.r.x; p[4] = Tn.r.y; p[5] = Tn.r.z; p[6] = Tn.r.w;
        }
    }
    red[tid] = sc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
    if (tid == 0) A.part[(size_t)b * A.nPart + A.nPart - 1] = red[0];
    if (tid < 64) {
        double s = 0;
        for (int i = tid; i < nBack; i += 64) s += A.part[(size_t)b * A.nPart + i];
        if (tid == 0 && A.shard != 2) s += red[0];   // (every rank of the sharded form computes the same pose part: it enters the all-reduced sum once)
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (tid == 0) A.st[b].scale = s;
    }
}

// computeActiveErrors' sum (k_lm_sum_partials with what = 0: the same lane-strided partial sums and butterfly) and the rho test of the trial in one launch:
// one wave per window, lane 0 decides
static __device__ __forceinline__ void lm_decide(LmState& s, int* flag) {
    if (!s.needTrial) return;
    double tempChi = s.tempChi;
    if (!s.ok) tempChi = 1.7976931348623157e308;
    double rho = s.currentChi - tempChi;
    double scale = s.ok ? s.scale : 0.0;
    scale += 1e-3;
    rho /= scale;
    s.accepted = 0;
    if (rho > 0 && tempChi < 1.7976931348623157e308 && tempChi == tempChi) {   // good step
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        const double scaleFactor = fmax(1. / 3., alpha);
        s.lambda *= scaleFactor; s.ni = 2; s.currentChi = tempChi; s.accepted = 1;
    } else {
        s.lambda *= s.ni; s.ni *= 2;
    }
    s.rho = rho; s.qmax++; s.trials++; s.ok = 1;
    s.needTrial = (rho < 0 && s.qmax < 100) ? 1 : 0;
    if (s.needTrial) atomicAdd(flag, 1);
}
static __global__ __launch_bounds__(64) void k_lm_sum_decide(LmArgs A, int n) {
    const int b = blockIdx.x, lane = threadIdx.x;
    double sum = 0;
    for (int i = lane; i < n; i += 64) sum += A.part[(size_t)b * A.nPart + i];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane != 0) return;
    LmState& s = A.st[b];
    s.tempChi = sum;
    lm_decide(s, A.flag);
}
// the sharded form (lba_optimize_sharded): the per-window scalars a rank can only know for ITS landmarks go through one all-reduce between the
// kernels that produce them and the ones that read them.  what 0: tempChi (sum); 1: tempChi, scale, failures (sums) of a lambda trial; 2: maxDiag (max)
static __global__ void k_lm_pack(LmArgs A, int batch, int what) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const LmState& s = A.st[b];
    double* r = A.red + (size_t)b * 4;
    if (what == 2) { r[0] = s.maxDiag; r[1] = r[2] = r[3] = 0; return; }
    r[0] = s.tempChi; r[1] = what == 1 ? s.scale : 0.0; r[2] = (what == 1 && !s.ok) ? 1.0 : 0.0; r[3] = 0;
}
static __global__ void k_lm_unpack(LmArgs A, int batch, int what) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    LmState& s = A.st[b];
    const double* r = A.red + (size_t)b * 4;
    if (what == 2) { s.maxDiag = r[0]; return; }
    s.tempChi = r[0];
    if (what == 1) { s.scale = r[1]; s.ok = r[2] == 0.0 ? 1 : 0; lm_decide(s, A.flag); }
}

// pop: windows whose last trial was rejected get their state back (also those that will retry)
static __global__ void k_lm_restore(LmArgs A, size_t nPose, size_t nPoint, int capP7, int capL3) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nPose) { const LmState& s = A.st[i / capP7]; if (s.active && s.qmax > 0 && !s.accepted) A.poses[i] = A.posesBak[i]; }
    else if (i < nPose + nPoint) { const size_t j = i - nPose; const LmState& s = A.st[j / capL3]; if (s.active && s.qmax > 0 && !s.accepted) A.points[j] = A.pointsBak[j]; }
}

static __global__ void k_lm_end(LmArgs A, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    LmState& s = A.st[b];
    if (!s.active) return;
    s.iter++;
    if (s.qmax == 100 || s.rho == 0) { s.active = 0; return; }      // Terminate
    if ((s.iniChi - s.currentChi) * 1e3 < s.iniChi) s.nBad++; else s.nBad = 0;
    if (s.nBad >= 3) s.active = 0;
}

static size_t lm_align(size_t v) { return (v + 255) & ~(size_t)255; }
// rows of the reduced camera system split over G workgroups each (k_lm_schur_rows / k_lm_schur_combine): only while the call has fewer rows than the
// machine has compute units, at least 256 edges per slice, and the partial rows fit 64 MB of workspace
#ifndef LM_SCHUR_SPLIT_MAX
#define LM_SCHUR_SPLIT_MAX 8
#endif
#ifndef LM_SCHUR_SPLIT_MIN_EDGES
#define LM_SCHUR_SPLIT_MIN_EDGES 256   // edges of a row per slice, at least (tests build with a tiny value to cover the split on their small windows)
#endif
static size_t lm_schur_part_doubles(size_t rows6) { return (rows6 / 12 + 1) * 36 + 6; }     // per (row, slice) of a system of rows6 unknowns
// compute units of the calling thread's current device (256 on MI355X), asked once per device
static int lm_device_cus() {
    static std::mutex mu;
    static int cached[64] = {0};
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lock(mu);
    if (!cached[dev]) cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    return cached[dev];
}
static int lm_schur_groups(int batch, int rows, int cap_e, size_t np6alloc) {
    if (rows < 1 || batch < 1) return 1;
    if (rows / 2 + 1 > LM_SCHUR_ROWCAP) return 1;   // a split row leaves its slice in ONE chunk of schurPart: rows longer than the LDS chunk are never split
    int G = std::min(LM_SCHUR_SPLIT_MAX, lm_device_cus() / std::max(1, batch * rows));
    G = std::min(G, std::max(1, cap_e / rows / LM_SCHUR_SPLIT_MIN_EDGES));
    const size_t perG = (size_t)batch * (np6alloc / 6) * lm_schur_part_doubles(np6alloc) * 8;
    G = (int)std::min<size_t>((size_t)G, ((size_t)64 << 20) / std::max<size_t>(perG, 1));
    return std::max(G, 1);
}
// the one-launch-per-panel factorisation (k_lm_chol_step) needs L and y next to Hs / xp: only calls that can take it reserve them
// (reserved by the shape's capacity np6cap — the call's own system may be smaller — while that stays under 1 GB)
static bool lm_chol_step_possible(int batch, size_t np6cap) { return batch <= LM_CHOL_SPLIT_MAX_BATCH && (size_t)batch * np6cap * np6cap * 8 <= ((size_t)1 << 30); }

extern "C" size_t lba_lm_workspace_bytes(const lba_problem* p, int batch) {
    if (!p || batch < 1) return 0;
    const size_t B = (size_t)batch, np6 = (size_t)p->cap_p * 6;
    const size_t nPart = (size_t)std::max((p->cap_e + 255) / 256, (p->cap_l + BS_LB - 1) / BS_LB) + 1;
    size_t s = 0;
    s += lm_align(B * p->cap_p * 36 * 8) + lm_align(B * p->cap_p * 6 * 8);            // Hpp, bp
    s += lm_align(B * p->cap_l * 9 * 8) + lm_align(B * p->cap_l * 3 * 8);            // Hll, bl
    s += lm_align(B * p->cap_e * 18 * 8);                                             // Hpl
    s += lm_align(B * p->cap_p * 7 * 8) + lm_align(B * p->cap_l * 3 * 8);            // backups
    s += lm_align(B * p->cap_l * 9 * 8) + lm_align(B * p->cap_l * 3 * 8);            // Dinv, db
    s += lm_align(B * np6 * np6 * 8) + lm_align(B * np6 * 8) + lm_align(B * p->cap_l * 3 * 8);   // Hs, xp, xl
    s += lm_align(B * nPart * 8) + lm_align(B * sizeof(LmState)) + lm_align(B * 4) + 256;
    s += lm_align(B * 4 * 8);                                                         // packed scalars of the sharded form
    s += lm_align(B * p->cap_e * 16) + lm_align(B * ((size_t)p->cap_e + 8) * 4);     // Schur row metadata
    {   // partial Schur rows (few windows per call): sized for the most slices any free-pose count of this shape could take
        int gmax = 1;
        for (int rows = 1; rows <= p->cap_p; rows++) gmax = std::max(gmax, lm_schur_groups(batch, rows, p->cap_e, np6));
        if (gmax > 1) s += lm_align(B * (np6 / 6) * gmax * lm_schur_part_doubles(np6) * 8);
    }
    if (lm_chol_step_possible(batch, np6)) s += lm_align(B * np6 * np6 * 8) + lm_align(B * np6 * 8) + lm_align(B * ((np6 + 31) / 32) * 1024 * 8);   // cholL, cholY, cholX (few windows per call only)
    if (np6 > WG_CHOL_LDS_MAX_LD) s += lm_align(B * np6 * CH_LD * 8);                 // out-of-LDS Cholesky panel (only if ALL poses could be free)
    return s;
}

// shard / reduce / user: lba_optimize_sharded (0 / nullptr: the whole window lives here)
static int lba_optimize_impl(const lba_problem* prob, int batch, int iterations, void* d_workspace, double* h_stats,
                             const volatile int* abort_flag, const volatile unsigned char* abort_byte, void* stream,
                             int shard = 0, lba_allreduce_fn reduce = nullptr, void* user = nullptr) {
    lba_system dummy;
    memset(&dummy, 0, sizeof(dummy));
    int rc = lba_check(prob, batch, &dummy);
    if (rc != ORB_OK || !d_workspace || iterations < 0) return ORB_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const lba_problem& P = *prob;
    // The reduced camera system holds the FREE poses only (fixed key frames — lFixedCameras, Optimizer.cc:2011-2039 — feed the Schur terms
    // through their edges but own no Hessian block): its dimension is 6 x the largest free-pose count of the batch, not 6 x cap_p.
    std::vector<int32_t> nf((size_t)batch);
    {
        std::vector<int32_t> hid((size_t)batch * P.cap_p), npz((size_t)batch);
        if (hipMemcpyAsync(hid.data(), P.pose_hidx, hid.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return ORB_E_HIP;
        if (hipMemcpyAsync(npz.data(), P.n_poses, (size_t)batch * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return ORB_E_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return ORB_E_HIP;
        for (int b = 0; b < batch; b++) {
            int m = 0;
            for (int i = 0; i < std::min(npz[b], P.cap_p); i++) m = std::max(m, hid[(size_t)b * P.cap_p + i] + 1);
            nf[b] = m;
        }
    }
    const int maxFree = std::max(1, *std::max_element(nf.begin(), nf.end()));
    const size_t B = (size_t)batch, np6 = (size_t)maxFree * 6, np6cap = (size_t)P.cap_p * 6;
    const int nPart = std::max((P.cap_e + 255) / 256, (P.cap_l + BS_LB - 1) / BS_LB) + 1;
    char* w = (char*)d_workspace;
    auto take = [&](size_t bytes) { char* p = w; w += lm_align(bytes); return p; };
    LmArgs A;
    memset(&A, 0, sizeof(A));
    A.P = P;
    A.S.Hpp = (double*)take(B * P.cap_p * 36 * 8); A.S.bp = (double*)take(B * P.cap_p * 6 * 8);
    A.S.Hll = (double*)take(B * P.cap_l * 9 * 8); A.S.bl = (double*)take(B * P.cap_l * 3 * 8);
    A.S.Hpl = (double*)take(B * P.cap_e * 18 * 8);
    A.posesBak = (double*)take(B * P.cap_p * 7 * 8); A.pointsBak = (double*)take(B * P.cap_l * 3 * 8);
    A.Dinv = (double*)take(B * P.cap_l * 9 * 8); A.db = (double*)take(B * P.cap_l * 3 * 8);
    A.Hs = (double*)take(B * np6cap * np6cap * 8); A.xp = (double*)take(B * np6cap * 8); A.xl = (double*)take(B * P.cap_l * 3 * 8);   // used with ld = np6
    A.part = (double*)take(B * nPart * 8); A.st = (LmState*)take(B * sizeof(LmState));
    int32_t* nfree = (int32_t*)take(B * 4);
    A.flag = (int*)take(4);
    A.rowMeta = (int4*)take(B * P.cap_e * 16); A.edgeH = (int32_t*)take(B * ((size_t)P.cap_e + 8) * 4);
    A.red = (double*)take(B * 4 * 8);
    A.shard = shard;
    // the cross-rank sums of the sharded form: `what` as in k_lm_pack / k_lm_unpack
#define LM_REDUCE(ptr, n, op) do { if (shard && reduce(user, (ptr), (n), (op), stream) != 0) return ORB_E_HIP; } while (0)
#define LM_REDUCE_STATE(what) do { if (shard) {                                                            \
        hipLaunchKernelGGL(k_lm_pack, dim3((batch + 63) / 64), dim3(64), 0, st, A, batch, (what));             \
        LM_REDUCE(A.red, B * 4, (what) == 2 ? 1 : 0);                                                        \
        hipLaunchKernelGGL(k_lm_unpack, dim3((batch + 63) / 64), dim3(64), 0, st, A, batch, (what)); } } while (0)
    A.schurG = lm_schur_groups(batch, maxFree, P.cap_e, np6cap);
    if (A.schurG > 1) A.schurPart = (double*)take(B * (np6cap / 6) * A.schurG * lm_schur_part_doubles(np6cap) * 8);   // (used with the strides of np6)
    if (lm_chol_step_possible(batch, np6cap)) {       // used with ld = np6
        A.cholL = (double*)take(B * np6cap * np6cap * 8); A.cholY = (double*)take(B * np6cap * 8); A.cholX = (double*)take(B * ((np6cap + 31) / 32) * 1024 * 8);
    }
    A.poses = (double*)P.poses; A.points = (double*)P.points; A.nPart = nPart; A.np6 = (int)np6;

    if (hipMemcpyAsync(nfree, nf.data(), B * 4, hipMemcpyHostToDevice, st) != hipSuccess) return ORB_E_HIP;
    int otherKinds = 0;
    if (hipMemsetAsync(A.flag, 0, 4, st) != hipSuccess) return ORB_E_HIP;
    hipLaunchKernelGGL(k_lm_kinds, dim3((P.cap_e + 255) / 256, batch), dim3(256), 0, st, A, A.flag);
    if (hipMemcpyAsync(&otherKinds, A.flag, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return ORB_E_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return ORB_E_HIP;
    const int ks = (otherKinds & 1) ? 0 : (otherKinds & 2) ? 2 : 1;   // generic / pinhole mono + stereo / monocular pinhole kernels
#define LM_LAUNCH_MP(kern, ...) do { if (ks == 1) hipLaunchKernelGGL(kern<1>, __VA_ARGS__); else if (ks == 2) hipLaunchKernelGGL(kern<2>, __VA_ARGS__); \
                                     else hipLaunchKernelGGL(kern<0>, __VA_ARGS__); } while (0)
    // Cholesky panel: LDS while 6 x maxFree rows x 17 doubles fit (<= 180 free key frames — every LocalBundleAdjustment window); beyond that
    // (GlobalBundleAdjustemnt of a large map) the panel lives in the workspace and LDS holds the right-hand side only
    A.panExt = nullptr;
    if (np6 > WG_CHOL_LDS_MAX_LD) {
        if (np6cap <= WG_CHOL_LDS_MAX_LD) return ORB_E_INVALID;   // cannot happen: np6 <= np6cap
        A.panExt = (double*)take(B * np6cap * CH_LD * 8);
    }
    // panel width 32 for 54 ... 88 free key frames (6 x maxFree x 33 doubles fit LDS; smaller systems are faster with 16: dense_chol.inc), else 16
    const bool nb32 = !A.panExt && np6 <= WG_CHOL_NB32_MAX_LD && np6 > WG_CHOL_NB32_MIN_LD;
    const size_t cholSmem = A.panExt ? wg_chol_smem_bytes_ext_t<LM_CHOL_NT, CH_NB>((int)np6)
                                     : nb32 ? wg_chol_smem_bytes_t<LM_CHOL_NT, 32>((int)np6) : wg_chol_smem_bytes_t<LM_CHOL_NT, CH_NB>((int)np6);
    if (cholSmem > 160 * 1024) return ORB_E_CAPACITY;   // > ~20 000 unknowns: the right-hand side no longer fits LDS (documented in INTEGRATION.md)
    if (orb_lds_optin(nb32 ? (const void*)k_lm_chol<32> : (const void*)k_lm_chol<CH_NB>, cholSmem) != ORB_OK) return ORB_E_HIP;
    // few windows per call: one launch per 32-column panel, its tiles spread over the machine (k_lm_chol_step), then k_lm_chol_back_x — every system
    // size that fits the backward kernel's one-thread-per-unknown layout (<= 90 free key frames), LocalMapping's 20-50 key frame windows included
    // (one window, ms per optimize(5), one workgroup vs this: 20 free key frames 1.35 / 1.14, 30: 1.79 / 1.33, 50: 3.02 / 1.70, 80: 6.8 / 2.2)
    const bool cholStep = A.cholL && batch <= LM_CHOL_SPLIT_MAX_BATCH && np6 <= LM_CHOL_NT + LM_CHOLS_NB;
    if (!cholStep) A.cholL = A.cholY = A.cholX = nullptr;
    const int gB = (batch + 63) / 64;
    const size_t nPose = B * P.cap_p * 7, nPoint = B * P.cap_l * 3;
    const int gCopy = (int)((nPose + nPoint + 255) / 256);
    const dim3 gE((P.cap_e + 255) / 256, batch), gL((P.cap_l + 255) / 256, batch), gLB((P.cap_l + BS_LB - 1) / BS_LB, batch);
    hipLaunchKernelGGL(k_lm_init, dim3(gB), dim3(64), 0, st, A, batch);
    // Schur rows: a row's blocks in LDS (37 doubles each); rows of more than LM_SCHUR_ROWCAP blocks are produced in column chunks
    const int rowCap = std::min(maxFree / 2 + 1, LM_SCHUR_ROWCAP);
    // waves per row: 4 while the four copies of a row fit 48 KB of LDS (rows of up to 81 blocks = 160 free key frames), else 2, else 1
    // — but one wave per row once the batch alone fills the machine (256 CUs x 12 resident waves; measured: 58.5 vs 59.7 ms per optimize(5) of
    // 256 C5-size windows, and 1.62 vs 2.36 ms for one 30-key-frame window)
    const bool fills = (size_t)batch * (size_t)maxFree >= 3072;
    const int schurNW = fills ? 1 : lm_schur_smem_bytes(rowCap, 4) <= 48 * 1024 ? 4 : lm_schur_smem_bytes(rowCap, 2) <= 48 * 1024 ? 2 : 1;
    const size_t schurSmem = lm_schur_smem_bytes(rowCap, schurNW);
    hipLaunchKernelGGL(k_lm_rowmeta, dim3((P.cap_e + 8 + 255) / 256, batch), dim3(256), 0, st, A);
    if (orb_lds_optin((const void*)k_lm_schur_rows<1>, schurSmem) != ORB_OK) return ORB_E_HIP;
    int aborted = 0;
    for (int it = 0; it < iterations && !aborted; it++) {
        // computeActiveErrors + activeRobustChi2, buildSystem.  After the first iteration the state is the one the last lambda trial left —
        // accepted (its chi2 became currentChi) or restored (currentChi unchanged) — so its activeRobustChi2 is currentChi, bit for bit.
        if (it == 0) {
            LM_LAUNCH_MP(k_lm_errors, gE, dim3(256), 256 * 8, st, A);
            hipLaunchKernelGGL(k_lm_sum_partials, dim3(batch), dim3(64), 0, st, A, batch, (int)gE.x, 0, -1);
            LM_REDUCE_STATE(0);
        }
        {
            LbaArgs L;
            L.P = P; L.S = A.S;
            if (hipMemsetAsync(A.S.Hpp, 0, B * P.cap_p * 36 * 8, st) != hipSuccess) return ORB_E_HIP;
            if (hipMemsetAsync(A.S.bp, 0, B * P.cap_p * 6 * 8, st) != hipSuccess) return ORB_E_HIP;
            LM_LAUNCH_MP(k_lba_landmarks, dim3((P.cap_l + LBA_LB - 1) / LBA_LB, batch), dim3(LBA_CT), LBA_CT * 18 * 8, st, L);
            LM_LAUNCH_MP(k_lba_poses, dim3(P.cap_p, batch), dim3(POSES_NT), (POSES_NT / 64) * 27 * 8, st, L);
            // sharded: every rank linearised the edges of ITS landmarks — the pose-side blocks are sums over all of them (the all-reduce of
            // SURVEY 8(e)); H_ll / b_l / H_pl stay local
            LM_REDUCE(A.S.Hpp, B * P.cap_p * 36, 0);
            LM_REDUCE(A.S.bp, B * P.cap_p * 6, 0);
        }
        if (it == 0) { hipLaunchKernelGGL(k_lm_maxdiag, dim3(MAXDIAG_G, batch), dim3(256), 256 * 8, st, A); LM_REDUCE_STATE(2); }
        hipLaunchKernelGGL(k_lm_begin, dim3(gB), dim3(64), 0, st, A, batch);
        for (int trial = 0; trial < 100; trial++) {
            hipLaunchKernelGGL(k_lm_backup, dim3(gCopy), dim3(256), 0, st, A, nPose, nPoint, P.cap_p * 7, P.cap_l * 3);   // push
            hipLaunchKernelGGL(k_lm_dinv, gL, dim3(256), 0, st, A);
            {
                const dim3 gS((unsigned)((batch >= 8 ? ((batch + 7) / 8) * 8 : batch) * P.cap_p), (unsigned)A.schurG);
                if (schurNW == 4) hipLaunchKernelGGL(k_lm_schur_rows<4>, gS, dim3(256), schurSmem, st, A, rowCap, batch, (const int32_t*)nfree);
                else if (schurNW == 2) hipLaunchKernelGGL(k_lm_schur_rows<2>, gS, dim3(128), schurSmem, st, A, rowCap, batch, (const int32_t*)nfree);
                else hipLaunchKernelGGL(k_lm_schur_rows<1>, gS, dim3(64), schurSmem, st, A, rowCap, batch, (const int32_t*)nfree);
                if (A.schurG > 1) hipLaunchKernelGGL(k_lm_schur_combine, dim3(maxFree, batch), dim3(256), 0, st, A, (const int32_t*)nfree);
                // sharded: the reduced camera system is the sum of the ranks' Schur terms (+ H_pp + lambda I and b_p, which the owner rank added):
                // one all-reduce of [np6 x np6 | np6] per window (block_solver.hpp:381-432 builds it in one piece); every rank then factorises it
                LM_REDUCE(A.Hs, B * np6 * np6, 0);
                LM_REDUCE(A.xp, B * np6, 0);
            }
            if (cholStep) {
                for (int kb = 0; kb < (int)np6; kb += LM_CHOLS_NB) {
                    const int T = std::max(0, ((int)np6 - kb - LM_CHOLS_NB + 15) >> 4);
                    hipLaunchKernelGGL(k_lm_chol_step, dim3(std::max(1, T * (T + 1) / 2), batch), dim3(64), LM_CHOLF_SMEM, st, A, (const int32_t*)nfree, kb);
                }
                hipLaunchKernelGGL(k_lm_chol_back_x, dim3(batch), dim3(LM_CHOL_NT), lm_chol_back_x_smem((int)np6), st, A, (const int32_t*)nfree);
            } else if (nb32) hipLaunchKernelGGL(k_lm_chol<32>, dim3(batch), dim3(LM_CHOL_NT), cholSmem, st, A, (const int32_t*)nfree);
            else hipLaunchKernelGGL(k_lm_chol<CH_NB>, dim3(batch), dim3(LM_CHOL_NT), cholSmem, st, A, (const int32_t*)nfree);
            hipLaunchKernelGGL(k_lm_backsub, gLB, dim3(BS_CT), (BS_CT * 21 + BS_LB) * 8, st, A);
            hipLaunchKernelGGL(k_lm_update_pose, dim3(batch), dim3(256), 256 * 8, st, A, (int)gLB.x);       // (+ the scale sum)
            LM_LAUNCH_MP(k_lm_errors, gE, dim3(256), 256 * 8, st, A);
            if (!shard) hipLaunchKernelGGL(k_lm_sum_decide, dim3(batch), dim3(64), 0, st, A, (int)gE.x);               // chi2 sum + rho test
            else {   // chi2 / computeScale / failures summed over the ranks first, then the same rho test on every rank (k_lm_unpack)
                hipLaunchKernelGGL(k_lm_sum_partials, dim3(batch), dim3(64), 0, st, A, batch, (int)gE.x, 0, -1);
                LM_REDUCE_STATE(1);
            }
            hipLaunchKernelGGL(k_lm_restore, dim3(gCopy), dim3(256), 0, st, A, nPose, nPoint, P.cap_p * 7, P.cap_l * 3);       // pop
            int more = 0;
            if (hipMemcpyAsync(&more, A.flag, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return ORB_E_HIP;
            if (hipStreamSynchronize(st) != hipSuccess) return ORB_E_HIP;
            if ((abort_flag && *abort_flag) || (abort_byte && *abort_byte)) { aborted = 1; break; }
            if (!more) break;
        }
        hipLaunchKernelGGL(k_lm_end, dim3(gB), dim3(64), 0, st, A, batch);
    }
    // final activeRobustChi2 + stats
    {
        // evaluate every window (also the terminated ones) at its final state
        std::vector<LmState> hs(B);
        if (hipMemcpyAsync(hs.data(), A.st, B * sizeof(LmState), hipMemcpyDeviceToHost, st) != hipSuccess) return ORB_E_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return ORB_E_HIP;
        std::vector<LmState> on = hs;
        for (auto& s : on) s.active = 1;
        if (hipMemcpyAsync(A.st, on.data(), B * sizeof(LmState), hipMemcpyHostToDevice, st) != hipSuccess) return ORB_E_HIP;
        LM_LAUNCH_MP(k_lm_errors, gE, dim3(256), 256 * 8, st, A);
        hipLaunchKernelGGL(k_lm_sum_partials, dim3(batch), dim3(64), 0, st, A, batch, (int)gE.x, 0, -1);
        LM_REDUCE_STATE(0);
#undef LM_REDUCE_STATE
#undef LM_REDUCE
        std::vector<LmState> fin(B);
#undef LM_LAUNCH_MP
        if (hipMemcpyAsync(fin.data(), A.st, B * sizeof(LmState), hipMemcpyDeviceToHost, st) != hipSuccess) return ORB_E_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return ORB_E_HIP;
        if (h_stats)
            for (size_t b = 0; b < B; b++) { h_stats[4 * b] = hs[b].iter; h_stats[4 * b + 1] = fin[b].tempChi; h_stats[4 * b + 2] = hs[b].lambda; h_stats[4 * b + 3] = hs[b].trials; }
    }
    if (hipGetLastError() != hipSuccess) return ORB_E_HIP;
    return aborted ? ORB_E_ABORTED : ORB_OK;
}

extern "C" int lba_optimize(const lba_problem* prob, int batch, int iterations, void* d_workspace, double* h_stats,
                            const volatile int* abort_flag, void* stream) {
    return lba_optimize_impl(prob, batch, iterations, d_workspace, h_stats, abort_flag, nullptr, stream);
}
// the same with the reference's own stop flag type: Optimizer::LocalBundleAdjustment receives `bool* pbStopFlag` (a byte LocalMapping sets from
// another thread, LocalMapping.cc:1189) and hands it to g2o's setForceStopFlag (Optimizer.cc:1975-1976)
extern "C" int lba_optimize_stopflag(const lba_problem* prob, int batch, int iterations, void* d_workspace, double* h_stats,
                                     const volatile unsigned char* pb_stop_flag, void* stream) {
    return lba_optimize_impl(prob, batch, iterations, d_workspace, h_stats, nullptr, pb_stop_flag, stream);
}

// One window (per batch entry) sharded by LANDMARK over several processes (BASELINE configs[4]): every rank passes the whole pose set and
// the edges / points of ITS landmarks; the pose-side sums (H_pp, b_p), the reduced camera system, chi2, computeScale and the lambda start value go
// through `reduce` (an all-reduce over the ranks: RCCL on MI355X, gloo in the CPU tests), after which every rank takes the same decisions, solves the
// same system and updates all poses; landmarks are back-substituted where they live.  owner: non-zero on EXACTLY ONE rank.  No stop flag: the ranks
// must take the same number of trials.
extern "C" int lba_optimize_sharded(const lba_problem* prob, int batch, int iterations, void* d_workspace, double* h_stats, int owner,
                                    lba_allreduce_fn reduce, void* user, void* stream) {
    if (!reduce) return ORB_E_INVALID;
    return lba_optimize_impl(prob, batch, iterations, d_workspace, h_stats, nullptr, nullptr, stream, owner ? 1 : 2, reduce, user);
}

// ============================================================================================================
// Optimizer::PoseOptimization (reference src/Optimizer.cc:907-1273) — motion-only BA, one workgroup per frame,
// the whole 4-round x 10-iteration Levenberg-Marquardt schedule inside ONE launch (no host round trips): every thread keeps
// the same LM scalars, block reductions (fixed order) give H (6x6), b and chi2, each thread solves the 6x6 system redundantly.
// ============================================================================================================
struct PLin { int D; double e[3], J[18], chi2; };   // J: 3 x 6 row-major, third row zero for 2-D edges

// PH: every edge of the batch is an EdgeSE3ProjectXYZOnlyPose or an EdgeStereoSE3ProjectXYZOnlyPose on a pinhole camera (the monocular, stereo
// and RGB-D pinhole configurations): the fisheye model and the right-camera (ToBody) edge are gone at compile time; same expressions in the
// same order as the generic path takes for such edges.
template <bool WITH_JAC, bool PH = false>
static __device__ __forceinline__ void pose_linearize(const pose_edge& E, const SE3& T, const lba_camera& cam, PLin& L) {
    const double Xw[3] = {(double)E.xw[0], (double)E.xw[1], (double)E.xw[2]};
    L.e[2] = 0;
    if (E.kind == LBA_EDGE_STEREO) {   // g2o::EdgeStereoSE3ProjectXYZOnlyPose, types_six_dof_expmap.cpp:339-346, 375-404
        L.D = 3;
        double xt[3];
        se3_map(T, Xw, xt);
        const double fx = cam.p[0], fy = cam.p[1], cx = cam.p[2], cy = cam.p[3], bf = cam.bf;
        const float invzf = 1.0f / xt[2];   // float invz, double bf member
        double proj[3];
        proj[0] = xt[0] * invzf * fx + cx; proj[1] = xt[1] * invzf * fy + cy; proj[2] = proj[0] - bf * invzf;
        L.e[0] = (double)E.obs[0] - proj[0]; L.e[1] = (double)E.obs[1] - proj[1]; L.e[2] = (double)E.obs[2] - proj[2];
        if (WITH_JAC) {
            const double x = xt[0], y = xt[1], invz = 1.0 / xt[2], invz_2 = invz * invz;
            double* J = L.J;
            J[0] = x * y * invz_2 * fx; J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx; J[3] = -invz * fx; J[4] = 0; J[5] = x * invz_2 * fx;
            J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy; J[8] = -x * invz * fy; J[9] = 0; J[10] = -invz * fy; J[11] = y * invz_2 * fy;
            J[12] = J[0] - bf * y * invz_2; J[13] = J[1] + bf * x * invz_2; J[14] = J[2]; J[15] = J[3]; J[16] = 0; J[17] = J[5] - bf * invz_2;
        }
    } else {   // EdgeSE3ProjectXYZOnlyPose / ...ToBody, OptimizableTypes.h:47-51,75-79, .cpp:50-65,93-109
        L.D = 2;
        double xl[3], xp[3], proj[2];
        se3_map(T, Xw, xl);
        Quat ql = {cam.trl_q[0], cam.trl_q[1], cam.trl_q[2], cam.trl_q[3]};
        if (PH || E.kind == LBA_EDGE_MONO) {
            xp[0] = xl[0]; xp[1] = xl[1]; xp[2] = xl[2];
            if constexpr (PH) {
                proj[0] = cam.p[0] * xp[0] / xp[2] + cam.p[2];   // Pinhole.cpp:43-49
                proj[1] = cam.p[1] * xp[1] / xp[2] + cam.p[3];
            } else cam_project(cam, xp, proj);
        } else {
            SE3 Trl;
            Trl.r = ql; Trl.t[0] = cam.trl_t[0]; Trl.t[1] = cam.trl_t[1]; Trl.t[2] = cam.trl_t[2];
            double xe[3];
            se3_map(se3_mul(Trl, T), Xw, xe);
            cam_project(cam, xe, proj);
            se3_map(Trl, xl, xp);
        }
        L.e[0] = (double)E.obs[0] - proj[0]; L.e[1] = (double)E.obs[1] - proj[1];
        if (WITH_JAC) {
            double Jp[6], M[6];
            if constexpr (PH) {
                Jp[0] = cam.p[0] / xp[2]; Jp[1] = 0; Jp[2] = -cam.p[0] * xp[0] / (xp[2] * xp[2]);   // Pinhole.cpp:89-100
                Jp[3] = 0; Jp[4] = cam.p[1] / xp[2]; Jp[5] = -cam.p[1] * xp[1] / (xp[2] * xp[2]);
            } else cam_project_jac(cam, xp, Jp);
#pragma unroll
            for (int i = 0; i < 6; i++) Jp[i] = -Jp[i];
            if (PH || E.kind == LBA_EDGE_MONO) {
#pragma unroll
                for (int i = 0; i < 6; i++) M[i] = Jp[i];
            } else {
                double Rl[9];
                quat_to_R(ql, Rl);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) M[r * 3 + c] = Jp[r * 3] * Rl[c] + Jp[r * 3 + 1] * Rl[3 + c] + Jp[r * 3 + 2] * Rl[6 + c];
            }
            const double x = xl[0], y = xl[1], z = xl[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const double m0 = M[r * 3], m1 = M[r * 3 + 1], m2 = M[r * 3 + 2];
                L.J[r * 6 + 0] = m0 * 0 + m1 * (-z) + m2 * y;
                L.J[r * 6 + 1] = m0 * z + m1 * 0 + m2 * (-x);
                L.J[r * 6 + 2] = m0 * (-y) + m1 * x + m2 * 0;
                L.J[r * 6 + 3] = m0 * 1 + m1 * 0 + m2 * 0;
                L.J[r * 6 + 4] = m0 * 0 + m1 * 1 + m2 * 0;
                L.J[r * 6 + 5] = m0 * 0 + m1 * 0 + m2 * 1;
            }
#pragma unroll
            for (int i = 12; i < 18; i++) L.J[i] = 0.0;
        }
    }
    const double s = (double)E.inv_sigma2;
    L.chi2 = L.e[0] * s * L.e[0] + L.e[1] * s * L.e[1] + L.e[2] * s * L.e[2];
}

static __device__ __forceinline__ SE3 se3_exp(const double* u) {   // SE3Quat::exp, se3quat.h:223-256
    const double om0 = u[0], om1 = u[1], om2 = u[2];
    const double theta = sqrt(om0 * om0 + om1 * om1 + om2 * om2);
    const double O[9] = {0, -om2, om1, om2, 0, -om0, -om1, om0, 0};
    double O2[9], Rm[9], V[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) O2[r * 3 + c] = O[r * 3] * O[c] + O[r * 3 + 1] * O[3 + c] + O[r * 3 + 2] * O[6 + c];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const double I = (k % 4 == 0) ? 1.0 : 0.0;
        if (theta < 0.00001) { Rm[k] = I + O[k] + O2[k]; V[k] = Rm[k]; }
        else {
            Rm[k] = I + sin(theta) / theta * O[k] + (1 - cos(theta)) / (theta * theta) * O2[k];
            V[k] = I + (1 - cos(theta)) / (theta * theta) * O[k] + (theta - sin(theta)) / (theta * theta * theta) * O2[k];
        }
    }
    SE3 D;
    D.r = quat_from_R(Rm);
    quat_normalize(D.r);
#pragma unroll
    for (int r = 0; r < 3; r++) D.t[r] = V[r * 3] * u[3] + V[r * 3 + 1] * u[4] + V[r * 3 + 2] * u[5];
    return D;
}

// 6x6 SPD solve (column-major, lower Cholesky) in registers; returns false when the matrix is not positive definite
static __device__ __forceinline__ bool chol6(double* S, double* x) {
#pragma unroll
    for (int k = 0; k < 6; k++) {
        double dkk = S[k * 6 + k];
        if (!(dkk > 0) || !(dkk < 1.7e308)) return false;
        dkk = sqrt(dkk);
        S[k * 6 + k] = dkk;
#pragma unroll
        for (int i = k + 1; i < 6; i++) S[k * 6 + i] /= dkk;
#pragma unroll
        for (int j = k + 1; j < 6; j++)
#pragma unroll
            for (int i = j; i < 6; i++) S[j * 6 + i] -= S[k * 6 + i] * S[k * 6 + j];
    }
#pragma unroll
    for (int i = 0; i < 6; i++) { double v = x[i]; for (int k = 0; k < i; k++) v -= S[k * 6 + i] * x[k]; x[i] = v / S[i * 6 + i]; }
#pragma unroll
    for (int i = 5; i >= 0; i--) { double v = x[i]; for (int k = i + 1; k < 6; k++) v -= S[i * 6 + k] * x[k]; x[i] = v / S[i * 6 + i]; }
    return true;
}

struct PoseOptArgs {
    const double* posesIn; const pose_edge* edges; const int32_t* nEdges; int capE;
    const lba_camera* cams; double* posesOut; uint8_t* outlier; int32_t* nGood;
};

// block-wide sum of NV doubles per thread, result broadcast to every thread (fixed order: butterfly inside a wave, waves 0..3)
#ifndef POSE_T_MANY
#define POSE_T_MANY 64    // threads per frame of a batch that fills the machine: a frame has a few hundred edges, and at 256 VGPRs one wave per SIMD is
                     // all that fits — one-wave workgroups put four frames on a CU instead of one and need no cross-wave reduction step
#endif
#ifndef POSE_T_FEW
#define POSE_T_FEW 256   // threads per frame when there are fewer frames than SIMDs to put them on (Tracking optimises ONE frame per call: 1-2
                         // edges per lane instead of 5; MI355X, one 300-edge frame: 0.60 -> 0.41 ms with the redundant error pass below gone too)
#endif
template <int NV, int POSE_T>
static __device__ __forceinline__ void block_sum(double (&v)[NV], double* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++)
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off);
    if (POSE_T > 64) {
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NV; k++) scratch[wave * NV + k] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NV; k++) {
            double t = scratch[k];
#pragma unroll
            for (int w = 1; w < POSE_T / 64; w++) t += scratch[w * NV + k];
            v[k] = t;
        }
    }
}

template <bool PH, int POSE_T>
static __global__ __launch_bounds__(POSE_T) void k_pose_opt(PoseOptArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int ne = min(A.nEdges[b], A.capE);
    double* scratch = (double*)orb_smem;                  // [4 * 28]
    double* chiLast = scratch + 4 * 28;                   // [capE]
    uint8_t* level = (uint8_t*)(chiLast + A.capE);        // [capE]
    uint8_t* outl = level + A.capE;                       // [capE]
    const pose_edge* edges = A.edges + (size_t)b * A.capE;
    const SE3 T0 = load_pose(A.posesIn + (size_t)b * 7);
    for (int e = tid; e < ne; e += POSE_T) { chiLast[e] = 0; level[e] = 0; outl[e] = 0; }
    __syncthreads();
    const double deltaMono = (double)sqrtf(5.991f), deltaStereo = (double)sqrtf(7.815f);
    SE3 T = T0;
    int nBad = 0;
    bool robust = true;
    if (ne >= 3) {
        for (int it = 0; it < 4; it++) {
            T = T0;
            double na[1] = {0};
            for (int e = tid; e < ne; e += POSE_T) na[0] += level[e] == 0 ? 1.0 : 0.0;
            block_sum<1, POSE_T>(na, scratch);
            auto evalErrors = [&](const SE3& Tc) {   // computeActiveErrors + activeRobustChi2
                double s[1] = {0};
                for (int e = tid; e < ne; e += POSE_T) {
                    if (level[e] != 0) continue;
                    const pose_edge E = edges[e];
                    PLin L;
                    pose_linearize<false, PH>(E, Tc, A.cams[E.cam], L);
                    chiLast[e] = L.chi2;
                    double r0 = L.chi2;
                    if (robust) { const double d = E.kind == LBA_EDGE_STEREO ? deltaStereo : deltaMono, dsq = d * d; if (!(L.chi2 <= dsq)) r0 = 2 * sqrt(L.chi2) * d - dsq; }
                    s[0] += r0;
                }
                block_sum<1, POSE_T>(s, scratch);
                return s[0];
            };
            if (na[0] > 0) {
                double lambda = -1, ni = 2;
                int nBadLM = 0;
                double currentChi = 0;
                for (int iter = 0; iter < 10; iter++) {
                    // computeActiveErrors at the head of an iteration: after the first one the state is the one the last (accepted) trial
                    // evaluated — a rejected last trial leaves the loop below — so its errors and their sum are already there, bit for bit
                    if (iter == 0) currentChi = evalErrors(T);
                    double tempChi = currentChi;
                    const double iniChi = currentChi;
                    double acc[27];   // 21 lower-triangle entries of H (column-major order c2 <= c) + 6 of b
#pragma unroll
                    for (int k = 0; k < 27; k++) acc[k] = 0;
                    for (int e = tid; e < ne; e += POSE_T) {
                        if (level[e] != 0) continue;
                        const pose_edge E = edges[e];
                        PLin L;
                        pose_linearize<true, PH>(E, T, A.cams[E.cam], L);
                        double rho1 = 1.0;
                        if (robust) { const double d = E.kind == LBA_EDGE_STEREO ? deltaStereo : deltaMono; if (!(L.chi2 <= d * d)) rho1 = d / sqrt(L.chi2); }
                        const double s = (double)E.inv_sigma2;
                        int t = 0;
#pragma unroll
                        for (int c2 = 0; c2 < 6; c2++)
#pragma unroll
                            for (int c = c2; c < 6; c++) {
                                double h = 0;
#pragma unroll
                                for (int r = 0; r < 3; r++) h += L.J[r * 6 + c] * (rho1 * s) * L.J[r * 6 + c2];
                                acc[t++] += h;
                            }
#pragma unroll
                        for (int c = 0; c < 6; c++) {
                            double a = 0;
#pragma unroll
                            for (int r = 0; r < 3; r++) a += L.J[r * 6 + c] * s * L.e[r];
                            acc[21 + c] -= rho1 * a;
                        }
                    }
                    block_sum<27, POSE_T>(acc, scratch);
                    double H[36], bvec[6];
                    {
                        int t = 0;
#pragma unroll
                        for (int c2 = 0; c2 < 6; c2++)
#pragma unroll
                            for (int c = c2; c < 6; c++) { H[c2 * 6 + c] = acc[t]; H[c * 6 + c2] = acc[t]; t++; }
#pragma unroll
                        for (int c = 0; c < 6; c++) bvec[c] = acc[21 + c];
                    }
                    if (iter == 0) {
                        double md = 0;
#pragma unroll
                        for (int j = 0; j < 6; j++) md = fmax(md, fabs(H[j * 7]));
                        lambda = 1e-50 * md; ni = 2; nBadLM = 0;
                    }
                    double rho = 0;
                    int qmax = 0;
                    do {
                        const SE3 Tbak = T;
                        double S[36], x[6];
#pragma unroll
                        for (int k = 0; k < 36; k++) S[k] = H[k] + ((k % 7 == 0) ? lambda : 0.0);
#pragma unroll
                        for (int k = 0; k < 6; k++) x[k] = bvec[k];
                        const bool ok2 = chol6(S, x);
                        if (ok2) T = se3_mul(se3_exp(x), T);
                        tempChi = evalErrors(T);
                        if (!ok2) tempChi = 1.7976931348623157e308;
                        rho = currentChi - tempChi;
                        double scale = 0;
                        if (ok2) {
#pragma unroll
                            for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + bvec[j]);
                        }
                        scale += 1e-3;
                        rho /= scale;
                        if (rho > 0 && tempChi < 1.7976931348623157e308 && tempChi == tempChi) {
                            double alpha = 1. - pow((2 * rho - 1), 3);
                            alpha = fmin(alpha, 2. / 3.);
                            lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
                        } else { lambda *= ni; ni *= 2; T = Tbak; }
                        qmax++;
                    } while (rho < 0 && qmax < 100);
                    if (qmax == 100 || rho == 0) break;
                    if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
                    if (nBadLM >= 3) break;
                }
            }
            // classification (Optimizer.cc:1142-1246)
            double nb[1] = {0};
            const float th2 = 5.991f, th3 = 7.815f;
            for (int e = tid; e < ne; e += POSE_T) {
                const pose_edge E = edges[e];
                if (outl[e]) { PLin L; pose_linearize<false, PH>(E, T, A.cams[E.cam], L); chiLast[e] = L.chi2; }
                const float chi2 = (float)chiLast[e];
                if (chi2 > (E.kind == LBA_EDGE_STEREO ? th3 : th2)) { outl[e] = 1; level[e] = 1; nb[0] += 1.0; }
                else { outl[e] = 0; level[e] = 0; }
            }
            block_sum<1, POSE_T>(nb, scratch);
            nBad = (int)nb[0];
            if (it == 2) robust = false;
            if (ne < 10) break;
        }
    }
    for (int e = tid; e < A.capE; e += POSE_T) A.outlier[(size_t)b * A.capE + e] = e < ne ? outl[e] : 0;
    if (tid == 0) {
        double* o = A.posesOut + (size_t)b * 7;
        o[0] = T.t[0]; o[1] = T.t[1]; o[2] = T.t[2]; o[3] = T.r.x; o[4] = T.r.y; o[5] = T.r.z; o[6] = T.r.w;
        A.nGood[b] = ne >= 3 ? ne - nBad : 0;
    }
}

static int pose_optimize_impl(const double* d_poses_in, const pose_edge* d_edges, const int32_t* d_n_edges, int cap_e, int batch,
                              const lba_camera* d_cameras, int n_cameras, double* d_poses_out, uint8_t* d_outlier, int32_t* d_n_good,
                              bool pinhole, void* stream) {
    if (!d_poses_in || !d_edges || !d_n_edges || !d_cameras || !d_poses_out || !d_outlier || !d_n_good || cap_e < 1 || batch < 1 || n_cameras < 1)
        return ORB_E_INVALID;
    const size_t smem = (size_t)4 * 28 * 8 + (size_t)cap_e * 8 + 2 * (((size_t)cap_e + 15) & ~(size_t)15);
    if (smem > 64 * 1024) return ORB_E_INVALID;
    PoseOptArgs A{d_poses_in, d_edges, d_n_edges, cap_e, d_cameras, d_poses_out, d_outlier, d_n_good};
    if (batch * (POSE_T_FEW / 64) <= 1024) {   // four waves per frame while every wave still gets a SIMD of its own (256 CUs x 4)
        if (pinhole) hipLaunchKernelGGL((k_pose_opt<true, POSE_T_FEW>), dim3(batch), dim3(POSE_T_FEW), smem, (hipStream_t)stream, A);
        else hipLaunchKernelGGL((k_pose_opt<false, POSE_T_FEW>), dim3(batch), dim3(POSE_T_FEW), smem, (hipStream_t)stream, A);
    } else if (pinhole) hipLaunchKernelGGL((k_pose_opt<true, POSE_T_MANY>), dim3(batch), dim3(POSE_T_MANY), smem, (hipStream_t)stream, A);
    else hipLaunchKernelGGL((k_pose_opt<false, POSE_T_MANY>), dim3(batch), dim3(POSE_T_MANY), smem, (hipStream_t)stream, A);
    return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP;
}

extern "C" int pose_optimize(const double* d_poses_in, const pose_edge* d_edges, const int32_t* d_n_edges, int cap_e, int batch,
                             const lba_camera* d_cameras, int n_cameras, double* d_poses_out, uint8_t* d_outlier, int32_t* d_n_good,
                             void* stream) {
    return pose_optimize_impl(d_poses_in, d_edges, d_n_edges, cap_e, batch, d_cameras, n_cameras, d_poses_out, d_outlier, d_n_good, false, stream);
}

extern "C" int pose_optimize_hint(const double* d_poses_in, const pose_edge* d_edges, const int32_t* d_n_edges, int cap_e, int batch,
                                  const lba_camera* d_cameras, int n_cameras, double* d_poses_out, uint8_t* d_outlier, int32_t* d_n_good,
                                  unsigned hints, void* stream) {
    if (hints & ~(unsigned)LBA_HINT_PINHOLE) return ORB_E_INVALID;
    return pose_optimize_impl(d_poses_in, d_edges, d_n_edges, cap_e, batch, d_cameras, n_cameras, d_poses_out, d_outlier, d_n_good,
                              (hints & LBA_HINT_PINHOLE) != 0, stream);
}
