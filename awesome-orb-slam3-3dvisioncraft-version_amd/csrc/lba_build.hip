// lba_build.hip — stage 3 of the hot path on gfx950: the linearisation of Optimizer::LocalBundleAdjustment
// (reference src/Optimizer.cc:1957-2344 graph; g2o BlockSolver::buildSystem block_solver.hpp:502-560).
//
//   k_lba_landmarks  one thread per landmark walks its (contiguous, landmark-major) edges: computeError +
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

#include "../../include/orbhip.h"

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
template <bool WITH_JAC>
static __device__ __forceinline__ void edge_linearize(const lba_edge& E, const SE3& T, const double X[3], const lba_camera& cam,
                                                     double huberMono, double huberStereo, Lin& L) {
    L.e[2] = 0;
    if (E.kind == LBA_EDGE_STEREO) {
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
    } else {
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
    const double delta = E.kind == LBA_EDGE_STEREO ? huberStereo : huberMono;
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

static __global__ __launch_bounds__(128) void k_lba_landmarks(LbaArgs A) {
    const lba_problem& P = A.P;
    const int b = blockIdx.y;
    const int l = blockIdx.x * 128 + threadIdx.x;
    const int nl = min(P.n_points[b], P.cap_l);
    if (l >= nl) return;
    const int ne = min(P.n_edges[b], P.cap_e);
    const lba_edge* edges = P.edges + (size_t)b * P.cap_e;
    const double* poses = P.poses + (size_t)b * P.cap_p * 7;
    const int32_t* hidx = P.pose_hidx + (size_t)b * P.cap_p;
    const double* Xp = P.points + ((size_t)b * P.cap_l + l) * 3;
    const double X[3] = {Xp[0], Xp[1], Xp[2]};
    const int e0 = P.lm_start[(size_t)b * (P.cap_l + 1) + l], e1 = min(P.lm_start[(size_t)b * (P.cap_l + 1) + l + 1], ne);
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
    for (int ei = e0; ei < e1; ei++) {
        const lba_edge E = edges[ei];
        const SE3 T = load_pose(poses + (size_t)E.pose * 7);
        Lin L;
        edge_linearize<true>(E, T, X, P.cameras[E.cam], P.huber_mono, P.huber_stereo, L);
        const size_t eo = (size_t)b * P.cap_e + ei;
        if (A.S.err) { A.S.err[eo * 3] = L.e[0]; A.S.err[eo * 3 + 1] = L.e[1]; A.S.err[eo * 3 + 2] = L.e[2]; }
        if (A.S.chi2) A.S.chi2[eo] = L.chi2;
        if (A.S.rho) { A.S.rho[eo * 2] = L.rho0; A.S.rho[eo * 2 + 1] = L.rho1; }
        if (A.S.depth) A.S.depth[eo] = L.depth;
        // constructQuadraticForm, robust branch (base_binary_edge.hpp:91-113): omega_r = -Omega e rho1; wOmega = rho1 Omega
        const double s = (double)E.inv_sigma2, w = L.rho1 * s;
        double om[3];
        for (int i = 0; i < 3; i++) om[i] = -s * L.e[i] * L.rho1;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double acc = 0;
            for (int r = 0; r < 3; r++) acc += L.A[r * 3 + c] * om[r];
            bl[c] += acc;
#pragma unroll
            for (int c2 = 0; c2 < 3; c2++) {
                double h = 0;
                for (int r = 0; r < 3; r++) h += L.A[r * 3 + c] * w * L.A[r * 3 + c2];
                H[c2 * 3 + c] += h;
            }
        }
        if (A.S.Hpl) {
            double* hp = A.S.Hpl + eo * 18;
            const bool freePose = hidx[E.pose] >= 0;
#pragma unroll
            for (int c2 = 0; c2 < 3; c2++)
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double h = 0;
                    if (freePose)
                        for (int r = 0; r < 3; r++) h += L.B[r * 6 + c] * w * L.A[r * 3 + c2];
                    hp[c2 * 6 + c] = h;   // logical 6x3 block (pose, landmark), column-major (_hessianTransposed)
                }
        }
    }
    if (A.S.Hll) { double* o = A.S.Hll + ((size_t)b * P.cap_l + l) * 9; for (int i = 0; i < 9; i++) o[i] = H[i]; }
    if (A.S.bl) { double* o = A.S.bl + ((size_t)b * P.cap_l + l) * 3; for (int i = 0; i < 3; i++) o[i] = bl[i]; }
}

static __global__ __launch_bounds__(64) void k_lba_poses(LbaArgs A) {
    const lba_problem& P = A.P;
    const int b = blockIdx.y, pi = blockIdx.x, lane = threadIdx.x;
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
    for (int k = s0 + lane; k < s1; k += 64) {
        const lba_edge E = edges[pe[k]];
        const double* Xp = points + (size_t)E.point * 3;
        const double X[3] = {Xp[0], Xp[1], Xp[2]};
        Lin L;
        edge_linearize<true>(E, T, X, P.cameras[E.cam], P.huber_mono, P.huber_stereo, L);
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
        if (A.S.Hpp) {
            double* o = A.S.Hpp + ((size_t)b * P.cap_p + h) * 36;
            int t = 0;
            for (int c2 = 0; c2 < 6; c2++)
                for (int c = 0; c <= c2; c++) { o[c2 * 6 + c] = acc[t]; o[c * 6 + c2] = acc[t]; t++; }
        }
        if (A.S.bp) { double* o = A.S.bp + ((size_t)b * P.cap_p + h) * 6; for (int c = 0; c < 6; c++) o[c] = acc[21 + c]; }
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

extern "C" int lba_build_system(const lba_problem* prob, int batch, const lba_system* out, void* stream) {
    int rc = lba_check(prob, batch, out);
    if (rc != ORB_OK) return rc;
    LbaArgs A;
    A.P = *prob; A.S = *out;
    hipStream_t st = (hipStream_t)stream;
    // blocks of fixed poses / rows >= n are defined as zero
    if (out->Hpp && hipMemsetAsync(out->Hpp, 0, (size_t)batch * prob->cap_p * 36 * 8, st) != hipSuccess) return ORB_E_HIP;
    if (out->bp && hipMemsetAsync(out->bp, 0, (size_t)batch * prob->cap_p * 6 * 8, st) != hipSuccess) return ORB_E_HIP;
    hipLaunchKernelGGL(k_lba_landmarks, dim3((prob->cap_l + 127) / 128, batch), dim3(128), 0, st, A);
    if (out->Hpp || out->bp) hipLaunchKernelGGL(k_lba_poses, dim3(prob->cap_p, batch), dim3(64), 0, st, A);
    return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP;
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
