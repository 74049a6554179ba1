// ORACLE — TEST INFRASTRUCTURE ONLY (see orb_oracle.cpp header).  Stage 3: LocalBundleAdjustment linearisation.
//
// CPU restatement (no Eigen: the few fixed-size products are written out) of what one
// BlockSolver::buildSystem pass does for the LBA graph (reference src/Optimizer.cc:1957-2193):
//   SE3Quat(R,t), map, operator*, rotation().toRotationMatrix()   g2o/types/se3quat.h:60-110,217 + Eigen Quaternion
//   EdgeSE3ProjectXYZ::computeError / linearizeOplus               include/OptimizableTypes.h:113-119, src/OptimizableTypes.cpp:142-172
//   EdgeSE3ProjectXYZToBody::computeError / linearizeOplus         OptimizableTypes.h:142-147, .cpp:204-225
//   EdgeStereoSE3ProjectXYZ::computeError / cam_project / linearizeOplus   g2o/types/types_six_dof_expmap.h:156-161, .cpp:190-197, 228-275
//   Pinhole::project / projectJac (Eigen overloads)                src/CameraModels/Pinhole.cpp:43-49, 89-100
//   KannalaBrandt8::project / projectJac (Eigen overloads)         src/CameraModels/KannalaBrandt8.cpp:52-66 (atan2f/sqrtf!), 166-196
//   BaseEdge::chi2, RobustKernelHuber::robustify, robustInformation   g2o/core/base_edge.h:58-61,96-102, robust_kernel_impl.cpp:78-91
//   BaseBinaryEdge::constructQuadraticForm (robust branch)          g2o/core/base_binary_edge.hpp:55-120
// Eigen itself is not vendored/installed: Quaterniond(Matrix3d), toRotationMatrix and q*v are restated from Eigen 3's
// published algorithms [recalled]; any correct FP64 implementation is inside the 1e-4 budget.  PARITY UNPINNED vs a
// reference binary (unbuildable here); pinned to the math by finite-difference Jacobian tests (tests/test_lba_parity.py).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Camera {  // == lba_camera
    int32_t model, reserved;
    double p[8];
    double bf;
    double trl_q[4];
    double trl_t[3];
};
struct Edge {  // == lba_edge
    int32_t pose, point;
    int16_t kind, cam;
    float obs[3];
    float inv_sigma2;
};
struct Quat { double x, y, z, w; };
struct SE3 { Quat r; double t[3]; };

void normalizeRotation(Quat& q) {  // se3quat.h:283-288
    if (q.w < 0) { q.x *= -1; q.y *= -1; q.z *= -1; q.w *= -1; }
    const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}
Quat quatFromMatrix(const double m[9]) {  // Eigen quaternionbase_assign_impl<Other,3,3>; m row-major
    Quat q;
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        double c[3];
        c[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (m[k * 3 + j] - m[j * 3 + k]) * t;
        c[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        c[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        q.x = c[0]; q.y = c[1]; q.z = c[2];
    }
    return q;
}
void toRotationMatrix(const Quat& q, double R[9]) {  // Eigen QuaternionBase::toRotationMatrix; row-major out
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
void rotate(const Quat& q, const double v[3], double out[3]) {  // Eigen QuaternionBase::_transformVector
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
    out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
    out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
void se3map(const SE3& T, const double x[3], double out[3]) {  // se3quat.h:217
    rotate(T.r, x, out);
    out[0] += T.t[0]; out[1] += T.t[1]; out[2] += T.t[2];
}
SE3 se3mul(const SE3& a, const SE3& b) {  // se3quat.h:103-109
    SE3 r = a;
    double rt[3];
    rotate(a.r, b.t, rt);
    r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
    Quat q;
    q.w = a.r.w * b.r.w - a.r.x * b.r.x - a.r.y * b.r.y - a.r.z * b.r.z;
    q.x = a.r.w * b.r.x + a.r.x * b.r.w + a.r.y * b.r.z - a.r.z * b.r.y;
    q.y = a.r.w * b.r.y + a.r.y * b.r.w + a.r.z * b.r.x - a.r.x * b.r.z;
    q.z = a.r.w * b.r.z + a.r.z * b.r.w + a.r.x * b.r.y - a.r.y * b.r.x;
    r.r = q;
    normalizeRotation(r.r);
    return r;
}

void camProject(const Camera& c, const double v[3], double res[2]) {
    if (c.model == 0) {  // Pinhole.cpp:43-49
        res[0] = c.p[0] * v[0] / v[2] + c.p[2];
        res[1] = c.p[1] * v[1] / v[2] + c.p[3];
    } else {  // KannalaBrandt8.cpp:52-66 — note atan2f / sqrtf on doubles
        const double x2_plus_y2 = v[0] * v[0] + v[1] * v[1];
        const double theta = atan2f(sqrtf((float)x2_plus_y2), (float)v[2]);
        const double psi = atan2f((float)v[1], (float)v[0]);
        const double theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
                     theta9 = theta7 * theta2;
        const double r = theta + c.p[4] * theta3 + c.p[5] * theta5 + c.p[6] * theta7 + c.p[7] * theta9;
        res[0] = c.p[0] * r * std::cos(psi) + c.p[2];
        res[1] = c.p[1] * r * std::sin(psi) + c.p[3];
    }
}
void camProjectJac(const Camera& c, const double v[3], double J[6]) {  // row-major 2x3
    if (c.model == 0) {  // Pinhole.cpp:89-100
        J[0] = c.p[0] / v[2]; J[1] = 0; J[2] = -c.p[0] * v[0] / (v[2] * v[2]);
        J[3] = 0; J[4] = c.p[1] / v[2]; J[5] = -c.p[1] * v[1] / (v[2] * v[2]);
    } else {  // KannalaBrandt8.cpp:166-196
        const double x2 = v[0] * v[0], y2 = v[1] * v[1], z2 = v[2] * v[2];
        const double r2 = x2 + y2, r = std::sqrt(r2), r3 = r2 * r;
        const double theta = std::atan2(r, v[2]);
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

struct Lin { int D; double e[3], A[9], B[18], chi2, rho0, rho1, depth; };  // A: D x 3, B: D x 6, row-major

void linearize(const Edge& E, const SE3& T, const double X[3], const Camera& cam, double huberMono, double huberStereo, Lin& L) {
    double R[9];
    memset(&L, 0, sizeof(L));
    if (E.kind == 1) {  // stereo
        L.D = 3;
        double xt[3];
        se3map(T, X, xt);
        const double fx = cam.p[0], fy = cam.p[1], cx = cam.p[2], cy = cam.p[3];
        const float bf = (float)cam.bf;  // cam_project takes `const float& bf` (types_six_dof_expmap.cpp:190)
        const float invz = 1.0f / xt[2];
        double proj[3];
        proj[0] = xt[0] * invz * fx + cx;
        proj[1] = xt[1] * invz * fy + cy;
        proj[2] = proj[0] - bf * invz;
        for (int i = 0; i < 3; i++) L.e[i] = (double)E.obs[i] - proj[i];
        toRotationMatrix(T.r, R);
        const double x = xt[0], y = xt[1], z = xt[2], z_2 = z * z, bfd = cam.bf;
        double* A = L.A; double* B = L.B;
        A[0] = -fx * R[0] / z + fx * x * R[6] / z_2; A[1] = -fx * R[1] / z + fx * x * R[7] / z_2; A[2] = -fx * R[2] / z + fx * x * R[8] / z_2;
        A[3] = -fy * R[3] / z + fy * y * R[6] / z_2; A[4] = -fy * R[4] / z + fy * y * R[7] / z_2; A[5] = -fy * R[5] / z + fy * y * R[8] / z_2;
        A[6] = A[0] - bfd * R[6] / z_2; A[7] = A[1] - bfd * R[7] / z_2; A[8] = A[2] - bfd * R[8] / z_2;
        B[0] = x * y / z_2 * fx; B[1] = -(1 + (x * x / z_2)) * fx; B[2] = y / z * fx; B[3] = -1. / z * fx; B[4] = 0; B[5] = x / z_2 * fx;
        B[6] = (1 + y * y / z_2) * fy; B[7] = -x * y / z_2 * fy; B[8] = -x / z * fy; B[9] = 0; B[10] = -1. / z * fy; B[11] = y / z_2 * fy;
        B[12] = B[0] - bfd * y / z_2; B[13] = B[1] + bfd * x / z_2; B[14] = B[2]; B[15] = B[3]; B[16] = 0; B[17] = B[5] - bfd / z_2;
        L.depth = xt[2];
    } else {
        L.D = 2;
        double proj[2], Jp[6], xl[3], xp[3];
        se3map(T, X, xl);
        double Rm[9];
        if (E.kind == 0) {
            memcpy(xp, xl, sizeof(xp));
            camProject(cam, xp, proj);
            toRotationMatrix(T.r, Rm);
        } else {  // body edge: right camera through mTrl
            SE3 Trl;
            Trl.r = Quat{cam.trl_q[0], cam.trl_q[1], cam.trl_q[2], cam.trl_q[3]};
            memcpy(Trl.t, cam.trl_t, sizeof(Trl.t));
            const SE3 Trw = se3mul(Trl, T);
            double xe[3];
            se3map(Trw, X, xe);          // computeError: (mTrl * v1->estimate()).map(X)
            camProject(cam, xe, proj);
            se3map(Trl, xl, xp);         // linearizeOplus: X_r = mTrl.map(T_lw.map(X_w))
            toRotationMatrix(Trw.r, Rm);
            L.depth = xe[2];
        }
        if (E.kind == 0) L.depth = xl[2];
        L.e[0] = (double)E.obs[0] - proj[0];
        L.e[1] = (double)E.obs[1] - proj[1];
        camProjectJac(cam, xp, Jp);
        for (int i = 0; i < 6; i++) Jp[i] = -Jp[i];
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++) L.A[r * 3 + c] = Jp[r * 3] * Rm[c] + Jp[r * 3 + 1] * Rm[3 + c] + Jp[r * 3 + 2] * Rm[6 + c];
        const double x = xl[0], y = xl[1], z = xl[2];
        const double S[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};  // SE3deriv 3x6
        double M[6];  // 2x3: projectJac (* R(Trl) for the body edge)
        if (E.kind == 0) memcpy(M, Jp, sizeof(M));
        else {
            double Rl[9];
            toRotationMatrix(Quat{cam.trl_q[0], cam.trl_q[1], cam.trl_q[2], cam.trl_q[3]}, Rl);
            for (int r = 0; r < 2; r++)
                for (int c = 0; c < 3; c++) M[r * 3 + c] = Jp[r * 3] * Rl[c] + Jp[r * 3 + 1] * Rl[3 + c] + Jp[r * 3 + 2] * Rl[6 + c];
        }
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 6; c++) L.B[r * 6 + c] = M[r * 3] * S[c] + M[r * 3 + 1] * S[6 + c] + M[r * 3 + 2] * S[12 + c];
    }
    const double s = (double)E.inv_sigma2;  // information = Identity * invSigma2 (Optimizer.cc:2106,2140,2172)
    double chi2 = 0;
    for (int i = 0; i < L.D; i++) chi2 += L.e[i] * s * L.e[i];
    L.chi2 = chi2;
    const double delta = E.kind == 1 ? huberStereo : huberMono;
    if (delta <= 0) { L.rho0 = chi2; L.rho1 = 1; }
    else {
        const double dsqr = delta * delta;
        if (chi2 <= dsqr) { L.rho0 = chi2; L.rho1 = 1.; }
        else { const double sq = std::sqrt(chi2); L.rho0 = 2 * sq * delta - dsqr; L.rho1 = delta / sq; }
    }
}

}  // namespace

extern "C" {

// Quaterniond(R) + normalizeRotation, as SE3Quat(R, t) does (se3quat.h:60-62); R row-major
void olb_quat_from_matrix(const double* R, double* q4) {
    Quat q = quatFromMatrix(R);
    normalizeRotation(q);
    q4[0] = q.x; q4[1] = q.y; q4[2] = q.z; q4[3] = q.w;
}

// One buildSystem pass over a single window.  Edge order = array order (g2o active-edge order = insertion order).
// Outputs may be null.  Hpp [n_free*36] / bp [n_free*6] indexed by pose_hidx; Hll [n_pts*9]; bl [n_pts*3]; Hpl [E*18].
void olb_build_system(const double* poses, const int32_t* pose_hidx, int n_poses, const double* points, int n_points,
                      const void* edges_, int n_edges, const void* cams_, double huberMono, double huberStereo, double* Hpp,
                      double* bp, double* Hll, double* bl, double* Hpl, double* err, double* chi2, double* rho, double* depth,
                      double* robust_sum) {
    const Edge* edges = (const Edge*)edges_;
    const Camera* cams = (const Camera*)cams_;
    int nfree = 0;
    for (int i = 0; i < n_poses; i++) if (pose_hidx[i] >= 0) nfree = std::max(nfree, pose_hidx[i] + 1);
    if (Hpp) memset(Hpp, 0, sizeof(double) * 36 * nfree);
    if (bp) memset(bp, 0, sizeof(double) * 6 * nfree);
    if (Hll) memset(Hll, 0, sizeof(double) * 9 * n_points);
    if (bl) memset(bl, 0, sizeof(double) * 3 * n_points);
    if (Hpl) memset(Hpl, 0, sizeof(double) * 18 * n_edges);
    double rsum = 0;
    for (int ei = 0; ei < n_edges; ei++) {
        const Edge& E = edges[ei];
        SE3 T;
        const double* p = poses + (size_t)E.pose * 7;
        T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2];
        T.r = Quat{p[3], p[4], p[5], p[6]};
        Lin L;
        linearize(E, T, points + (size_t)E.point * 3, cams[E.cam], huberMono, huberStereo, L);
        if (err) for (int i = 0; i < 3; i++) err[(size_t)ei * 3 + i] = L.e[i];
        if (chi2) chi2[ei] = L.chi2;
        if (rho) { rho[2 * ei] = L.rho0; rho[2 * ei + 1] = L.rho1; }
        if (depth) depth[ei] = L.depth;
        rsum += L.rho0;
        const double s = (double)E.inv_sigma2, w = L.rho1 * s;   // weightedOmega = rho[1] * information
        double omega_r[3];
        for (int i = 0; i < L.D; i++) omega_r[i] = -s * L.e[i] * L.rho1;   // omega_r = -Omega*e; omega_r *= rho[1]
        const int hp = pose_hidx[E.pose];
        // landmark side: from->b += A^T omega_r ; from->A += A^T wOmega A      (points are never fixed in LBA)
        for (int c = 0; c < 3; c++) {
            double acc = 0;
            for (int r = 0; r < L.D; r++) acc += L.A[r * 3 + c] * omega_r[r];
            if (bl) bl[(size_t)E.point * 3 + c] += acc;
            for (int c2 = 0; c2 < 3; c2++) {
                double h = 0;
                for (int r = 0; r < L.D; r++) h += L.A[r * 3 + c] * w * L.A[r * 3 + c2];
                if (Hll) Hll[(size_t)E.point * 9 + c2 * 3 + c] += h;
            }
        }
        if (hp >= 0) {
            for (int c = 0; c < 6; c++) {
                double acc = 0;
                for (int r = 0; r < L.D; r++) acc += L.B[r * 6 + c] * omega_r[r];
                if (bp) bp[(size_t)hp * 6 + c] += acc;
                for (int c2 = 0; c2 < 6; c2++) {
                    double h = 0;
                    for (int r = 0; r < L.D; r++) h += L.B[r * 6 + c] * w * L.B[r * 6 + c2];
                    if (Hpp) Hpp[(size_t)hp * 36 + c2 * 6 + c] += h;
                }
                // _hessianTransposed += B^T wOmega A : logical 6x3 block (pose, landmark), column-major
                for (int c2 = 0; c2 < 3; c2++) {
                    double h = 0;
                    for (int r = 0; r < L.D; r++) h += L.B[r * 6 + c] * w * L.A[r * 3 + c2];
                    if (Hpl) Hpl[(size_t)ei * 18 + c2 * 6 + c] += h;
                }
            }
        }
    }
    if (robust_sum) *robust_sum = rsum;
}

// error vector of one edge as a function of a perturbed state — for finite-difference checks of the Jacobians.
// dpose: 6-vector (omega, upsilon) applied as exp(dpose)*T  (VertexSE3Expmap::oplusImpl, types_six_dof_expmap.h:73-76);
// dpoint: 3-vector added to the point (VertexSBAPointXYZ::oplusImpl, types_sba.h:49-53).
void olb_edge_error(const double* pose7, const double* point3, const void* edge_, const void* cam_, const double* dpose,
                    const double* dpoint, double* e3) {
    const Edge& E = *(const Edge*)edge_;
    const Camera& cam = *(const Camera*)cam_;
    SE3 T;
    T.t[0] = pose7[0]; T.t[1] = pose7[1]; T.t[2] = pose7[2];
    T.r = Quat{pose7[3], pose7[4], pose7[5], pose7[6]};
    // SE3Quat::exp (se3quat.h:223-256)
    const double* om = dpose;
    const double* up = dpose + 3;
    const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9], Rm[9], V[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += O[r * 3 + k] * O[k * 3 + c]; O2[r * 3 + c] = a; }
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        if (theta < 0.00001) { Rm[i] = I + O[i] + O2[i]; V[i] = Rm[i]; }
        else {
            Rm[i] = I + std::sin(theta) / theta * O[i] + (1 - std::cos(theta)) / (theta * theta) * O2[i];
            V[i] = I + (1 - std::cos(theta)) / (theta * theta) * O[i] + (theta - std::sin(theta)) / (std::pow(theta, 3)) * O2[i];
        }
    }
    SE3 D;
    D.r = quatFromMatrix(Rm);
    normalizeRotation(D.r);
    for (int r = 0; r < 3; r++) D.t[r] = V[r * 3] * up[0] + V[r * 3 + 1] * up[1] + V[r * 3 + 2] * up[2];
    const SE3 Tn = se3mul(D, T);
    const double X[3] = {point3[0] + dpoint[0], point3[1] + dpoint[1], point3[2] + dpoint[2]};
    Lin L;
    linearize(E, Tn, X, cam, 0, 0, L);
    e3[0] = L.e[0]; e3[1] = L.e[1]; e3[2] = L.e[2];
}

}  // extern "C"

// ---- one SparseOptimizer::optimize(iterations) call on the LBA graph, restated densely -------------------------------
//   OptimizationAlgorithmLevenberg::solve            g2o/core/optimization_algorithm_levenberg.cpp:61-168 (_tau=1e-50, 100 trials, "nBad" stop)
//   BlockSolver::setLambda / solve (Schur) / restoreDiagonal   g2o/core/block_solver.hpp:564-589, 354-486
//   LinearSolverEigen (SimplicialLDLT)                g2o/solvers/linear_solver_eigen.h:94-123  -> dense Cholesky here (same solution of the SPD system)
//   VertexSE3Expmap::oplusImpl / VertexSBAPointXYZ::oplusImpl  types_six_dof_expmap.h:73-76, types_sba.h:49-53
//   SparseOptimizer::optimize / push / pop / activeRobustChi2   sparse_optimizer.cpp:354-410, 100-114
namespace {
SE3 se3exp(const double* upd) {  // se3quat.h:223-256
    const double* om = upd; const double* up = upd + 3;
    const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9], Rm[9], V[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += O[r * 3 + k] * O[k * 3 + c]; O2[r * 3 + c] = a; }
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        if (theta < 0.00001) { Rm[i] = I + O[i] + O2[i]; V[i] = Rm[i]; }
        else {
            Rm[i] = I + std::sin(theta) / theta * O[i] + (1 - std::cos(theta)) / (theta * theta) * O2[i];
            V[i] = I + (1 - std::cos(theta)) / (theta * theta) * O[i] + (theta - std::sin(theta)) / (std::pow(theta, 3)) * O2[i];
        }
    }
    SE3 D;
    D.r = quatFromMatrix(Rm);
    normalizeRotation(D.r);
    for (int r = 0; r < 3; r++) D.t[r] = V[r * 3] * up[0] + V[r * 3 + 1] * up[1] + V[r * 3 + 2] * up[2];
    return D;
}
bool inv3(const double* D, double* out) {  // column-major 3x3 (symmetric): cofactor inverse (Eigen fixed-size inverse)
    const double a = D[0], b = D[3], c = D[6], d = D[1], e = D[4], f = D[7], g = D[2], h = D[5], i = D[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    const double id = 1.0 / det;
    out[0] = A * id; out[3] = -(b * i - c * h) * id; out[6] = (b * f - c * e) * id;
    out[1] = B * id; out[4] = (a * i - c * g) * id; out[7] = -(a * f - c * d) * id;
    out[2] = C * id; out[5] = -(a * h - b * g) * id; out[8] = (a * e - b * d) * id;
    return std::isfinite(det) && det != 0.0;
}
bool cholSolve(std::vector<double>& S, int n, std::vector<double>& rhs) {  // in-place dense Cholesky (lower), column-major
    for (int k = 0; k < n; k++) {
        double dkk = S[(size_t)k * n + k];
        if (!(dkk > 0) || !std::isfinite(dkk)) return false;
        dkk = std::sqrt(dkk);
        S[(size_t)k * n + k] = dkk;
        for (int i = k + 1; i < n; i++) S[(size_t)k * n + i] /= dkk;
        for (int j = k + 1; j < n; j++) {
            const double ljk = S[(size_t)k * n + j];
            for (int i = j; i < n; i++) S[(size_t)j * n + i] -= S[(size_t)k * n + i] * ljk;
        }
    }
    for (int i = 0; i < n; i++) { double v = rhs[i]; for (int k = 0; k < i; k++) v -= S[(size_t)k * n + i] * rhs[k]; rhs[i] = v / S[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double v = rhs[i]; for (int k = i + 1; k < n; k++) v -= S[(size_t)i * n + k] * rhs[k]; rhs[i] = v / S[(size_t)i * n + i]; }
    return true;
}
}  // namespace

extern "C" {
void olb_build_system(const double*, const int32_t*, int, const double*, int, const void*, int, const void*, double, double, double*,
                      double*, double*, double*, double*, double*, double*, double*, double*, double*);

// poses / points are updated in place.  stats[0]=iterations run, [1]=final robust chi2, [2]=final lambda, [3]=total LM trials.
void olb_optimize(double* poses, const int32_t* pose_hidx, int n_poses, double* points, int n_points, const void* edges_, int n_edges,
                  const void* cams_, double huberMono, double huberStereo, int iterations, double* stats) {
    const Edge* edges = (const Edge*)edges_;
    int nf = 0;
    for (int i = 0; i < n_poses; i++) if (pose_hidx[i] >= 0) nf = std::max(nf, pose_hidx[i] + 1);
    std::vector<int> poseOfH(nf, -1);
    for (int i = 0; i < n_poses; i++) if (pose_hidx[i] >= 0) poseOfH[pose_hidx[i]] = i;
    const int np6 = nf * 6;
    std::vector<double> Hpp((size_t)nf * 36), bp(np6), Hll((size_t)n_points * 9), bl((size_t)n_points * 3), Hpl((size_t)n_edges * 18),
        chi2(n_edges), rho((size_t)n_edges * 2);
    std::vector<int> lmStart(n_points + 1, 0);
    for (int e = 0; e < n_edges; e++) lmStart[edges[e].point + 1]++;
    for (int l = 0; l < n_points; l++) lmStart[l + 1] += lmStart[l];
    auto robustChi = [&]() { double r = 0; olb_build_system(poses, pose_hidx, n_poses, points, n_points, edges_, n_edges, cams_, huberMono, huberStereo,
                                                             nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &r); return r; };
    double lambda = -1, ni = 2;
    int nBad = 0, it = 0, trialsTotal = 0;
    double currentChi = 0;
    for (it = 0; it < iterations; it++) {
        currentChi = robustChi();                       // computeActiveErrors + activeRobustChi2
        double tempChi = currentChi;
        const double iniChi = currentChi;
        olb_build_system(poses, pose_hidx, n_poses, points, n_points, edges_, n_edges, cams_, huberMono, huberStereo, Hpp.data(), bp.data(),
                         Hll.data(), bl.data(), Hpl.data(), nullptr, chi2.data(), rho.data(), nullptr, nullptr);
        if (it == 0) {                                  // computeLambdaInit
            double maxDiagonal = 0;
            for (int h = 0; h < nf; h++) for (int j = 0; j < 6; j++) maxDiagonal = std::max(std::fabs(Hpp[(size_t)h * 36 + j * 7]), maxDiagonal);
            for (int l = 0; l < n_points; l++) for (int j = 0; j < 3; j++) maxDiagonal = std::max(std::fabs(Hll[(size_t)l * 9 + j * 4]), maxDiagonal);
            lambda = 1e-50 * maxDiagonal; ni = 2; nBad = 0;
        }
        double rhoLM = 0;
        int qmax = 0;
        std::vector<double> posesBak, pointsBak;
        do {
            posesBak.assign(poses, poses + (size_t)n_poses * 7);       // _optimizer->push()
            pointsBak.assign(points, points + (size_t)n_points * 3);
            // setLambda + Schur complement (block_solver.hpp:381-432)
            std::vector<double> S((size_t)np6 * np6, 0.0), coeff(np6, 0.0), Dinv((size_t)n_points * 9), db((size_t)n_points * 3);
            for (int h = 0; h < nf; h++)
                for (int c = 0; c < 6; c++) for (int r = 0; r < 6; r++) S[(size_t)(h * 6 + c) * np6 + h * 6 + r] = Hpp[(size_t)h * 36 + c * 6 + r] + (r == c ? lambda : 0.0);
            bool ok2 = true;
            for (int l = 0; l < n_points; l++) {
                double D[9];
                for (int k = 0; k < 9; k++) D[k] = Hll[(size_t)l * 9 + k] + ((k % 4 == 0) ? lambda : 0.0);
                if (!inv3(D, &Dinv[(size_t)l * 9])) ok2 = false;
                for (int r = 0; r < 3; r++) db[(size_t)l * 3 + r] = Dinv[(size_t)l * 9 + r] * bl[(size_t)l * 3] + Dinv[(size_t)l * 9 + 3 + r] * bl[(size_t)l * 3 + 1] + Dinv[(size_t)l * 9 + 6 + r] * bl[(size_t)l * 3 + 2];
                for (int e1 = lmStart[l]; e1 < lmStart[l + 1]; e1++) {
                    const int h1 = pose_hidx[edges[e1].pose];
                    if (h1 < 0) continue;
                    const double* Bi = &Hpl[(size_t)e1 * 18];       // 6x3 column-major
                    double BD[18];
                    for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) BD[c * 6 + r] = Bi[r] * Dinv[(size_t)l * 9 + c * 3] + Bi[6 + r] * Dinv[(size_t)l * 9 + c * 3 + 1] + Bi[12 + r] * Dinv[(size_t)l * 9 + c * 3 + 2];
                    for (int r = 0; r < 6; r++) coeff[h1 * 6 + r] += Bi[r] * db[(size_t)l * 3] + Bi[6 + r] * db[(size_t)l * 3 + 1] + Bi[12 + r] * db[(size_t)l * 3 + 2];
                    for (int e2 = lmStart[l]; e2 < lmStart[l + 1]; e2++) {
                        const int h2 = pose_hidx[edges[e2].pose];
                        if (h2 < 0) continue;
                        const double* Bj = &Hpl[(size_t)e2 * 18];
                        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++)
                            S[(size_t)(h2 * 6 + c) * np6 + h1 * 6 + r] -= BD[r] * Bj[c] + BD[6 + r] * Bj[6 + c] + BD[12 + r] * Bj[12 + c];
                    }
                }
            }
            std::vector<double> xp(np6);
            for (int i = 0; i < np6; i++) xp[i] = bp[i] - coeff[i];       // _bschur
            std::vector<double> Sc = S;
            if (ok2) ok2 = cholSolve(Sc, np6, xp);
            // back-substitution (block_solver.hpp:461-481): xl = Dinv * (bl - Hpl^T xp)
            std::vector<double> xl((size_t)n_points * 3, 0.0);
            if (ok2)
                for (int l = 0; l < n_points; l++) {
                    double cl[3] = {bl[(size_t)l * 3], bl[(size_t)l * 3 + 1], bl[(size_t)l * 3 + 2]};
                    for (int e1 = lmStart[l]; e1 < lmStart[l + 1]; e1++) {
                        const int h1 = pose_hidx[edges[e1].pose];
                        if (h1 < 0) continue;
                        const double* Bi = &Hpl[(size_t)e1 * 18];
                        for (int c = 0; c < 3; c++) for (int r = 0; r < 6; r++) cl[c] -= Bi[c * 6 + r] * xp[h1 * 6 + r];
                    }
                    for (int r = 0; r < 3; r++) xl[(size_t)l * 3 + r] = Dinv[(size_t)l * 9 + r] * cl[0] + Dinv[(size_t)l * 9 + 3 + r] * cl[1] + Dinv[(size_t)l * 9 + 6 + r] * cl[2];
                }
            // _optimizer->update(x): pose <- exp(dx) * pose ; point += dx
            if (ok2) {
                for (int h = 0; h < nf; h++) {
                    double* p = poses + (size_t)poseOfH[h] * 7;
                    SE3 T; T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2]; T.r = Quat{p[3], p[4], p[5], p[6]};
                    const SE3 Tn = se3mul(se3exp(&xp[h * 6]), T);
                    p[0] = Tn.t[0]; p[1] = Tn.t[1]; p[2] = Tn.t[2]; p[3] = Tn.r.x; p[4] = Tn.r.y; p[5] = Tn.r.z; p[6] = Tn.r.w;
                }
                for (size_t k = 0; k < (size_t)n_points * 3; k++) points[k] += xl[k];
            }
            tempChi = robustChi();
            if (!ok2) tempChi = 1.7976931348623157e308;
            rhoLM = currentChi - tempChi;
            double scale = 0;                              // computeScale: x^T (lambda x + b)
            if (ok2) {
                for (int i = 0; i < np6; i++) scale += xp[i] * (lambda * xp[i] + bp[i]);
                for (size_t k = 0; k < (size_t)n_points * 3; k++) scale += xl[k] * (lambda * xl[k] + bl[k]);
            }
            scale += 1e-3;
            rhoLM /= scale;
            if (rhoLM > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rhoLM - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                const double scaleFactor = std::max(1. / 3., alpha);
                lambda *= scaleFactor; ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                memcpy(poses, posesBak.data(), posesBak.size() * 8);    // pop
                memcpy(points, pointsBak.data(), pointsBak.size() * 8);
            }
            qmax++; trialsTotal++;
        } while (rhoLM < 0 && qmax < 100);
        if (qmax == 100 || rhoLM == 0) { it++; break; }                 // Terminate (the iteration still counts in cjIterations)
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) { it++; break; }
    }
    if (stats) { stats[0] = it; stats[1] = robustChi(); stats[2] = lambda; stats[3] = trialsTotal; }
}
}  // extern "C"

// ---- Optimizer::PoseOptimization (reference src/Optimizer.cc:907-1273) -------------------------------------------------
//   edges: EdgeSE3ProjectXYZOnlyPose (OptimizableTypes.h:47-51, .cpp:50-65), EdgeSE3ProjectXYZOnlyPoseToBody (.h:75-79, .cpp:93-109),
//          g2o::EdgeStereoSE3ProjectXYZOnlyPose (types_six_dof_expmap.cpp:339-346, 375-404)
//   BaseUnaryEdge::constructQuadraticForm  g2o/core/base_unary_edge.hpp:43-70 ;  LM as in olb_optimize with LinearSolverDense (6x6)
namespace {
struct PEdge { float xw[3]; float obs[3]; float inv_sigma2; int16_t kind, cam; };   // == pose_edge
struct PLin { int D; double e[3], J[18], chi2; };   // J: D x 6 row-major

void poseLinearize(const PEdge& E, const SE3& T, const Camera& cam, bool withJac, PLin& L) {
    memset(&L, 0, sizeof(L));
    const double Xw[3] = {(double)E.xw[0], (double)E.xw[1], (double)E.xw[2]};
    if (E.kind == 1) {
        L.D = 3;
        double xt[3];
        se3map(T, Xw, xt);
        const double fx = cam.p[0], fy = cam.p[1], cx = cam.p[2], cy = cam.p[3], bf = cam.bf;
        const float invzf = 1.0f / xt[2];                 // cam_project: float invz, double bf member
        double proj[3];
        proj[0] = xt[0] * invzf * fx + cx; proj[1] = xt[1] * invzf * fy + cy; proj[2] = proj[0] - bf * invzf;
        for (int i = 0; i < 3; i++) L.e[i] = (double)E.obs[i] - proj[i];
        if (withJac) {
            const double x = xt[0], y = xt[1], invz = 1.0 / xt[2], invz_2 = invz * invz;
            double* J = L.J;
            J[0] = x * y * invz_2 * fx; J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx; J[3] = -invz * fx; J[4] = 0; J[5] = x * invz_2 * fx;
            J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy; J[8] = -x * invz * fy; J[9] = 0; J[10] = -invz * fy; J[11] = y * invz_2 * fy;
            J[12] = J[0] - bf * y * invz_2; J[13] = J[1] + bf * x * invz_2; J[14] = J[2]; J[15] = J[3]; J[16] = 0; J[17] = J[5] - bf * invz_2;
        }
    } else {
        L.D = 2;
        double xl[3], xp[3], proj[2], M[6];
        se3map(T, Xw, xl);
        if (E.kind == 0) {
            memcpy(xp, xl, sizeof(xp));
            camProject(cam, xp, proj);
        } else {
            SE3 Trl;
            Trl.r = Quat{cam.trl_q[0], cam.trl_q[1], cam.trl_q[2], cam.trl_q[3]};
            memcpy(Trl.t, cam.trl_t, sizeof(Trl.t));
            double xe[3];
            se3map(se3mul(Trl, T), Xw, xe);
            camProject(cam, xe, proj);
            se3map(Trl, xl, xp);
        }
        L.e[0] = (double)E.obs[0] - proj[0]; L.e[1] = (double)E.obs[1] - proj[1];
        if (withJac) {
            double Jp[6];
            camProjectJac(cam, xp, Jp);
            for (int i = 0; i < 6; i++) Jp[i] = -Jp[i];
            if (E.kind == 0) memcpy(M, Jp, sizeof(M));
            else {
                double Rl[9];
                toRotationMatrix(Quat{cam.trl_q[0], cam.trl_q[1], cam.trl_q[2], cam.trl_q[3]}, Rl);
                for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) M[r * 3 + c] = Jp[r * 3] * Rl[c] + Jp[r * 3 + 1] * Rl[3 + c] + Jp[r * 3 + 2] * Rl[6 + c];
            }
            const double x = xl[0], y = xl[1], z = xl[2];
            const double S[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
            for (int r = 0; r < 2; r++) for (int c = 0; c < 6; c++) L.J[r * 6 + c] = M[r * 3] * S[c] + M[r * 3 + 1] * S[6 + c] + M[r * 3 + 2] * S[12 + c];
        }
    }
    const double s = (double)E.inv_sigma2;
    for (int i = 0; i < L.D; i++) L.chi2 += L.e[i] * s * L.e[i];
}
}  // namespace

extern "C" {
// One frame.  pose7 in/out; outlier[n_edges] out; returns nInitialCorrespondences - nBad (0 if fewer than 3 edges).
int opo_pose_optimize(const double* pose_in, const void* edges_, int n_edges, const void* cams_, double* pose_out, uint8_t* outlier) {
    const PEdge* edges = (const PEdge*)edges_;
    const Camera* cams = (const Camera*)cams_;
    const double deltaMono = (double)std::sqrt(5.991f), deltaStereo = (double)std::sqrt(7.815f);   // const float deltaMono = sqrt(5.991)
    const float chi2Mono[4] = {5.991f, 5.991f, 5.991f, 5.991f}, chi2Stereo[4] = {7.815f, 7.815f, 7.815f, 7.815f};
    for (int i = 0; i < 7; i++) pose_out[i] = pose_in[i];
    for (int e = 0; e < n_edges; e++) outlier[e] = 0;
    if (n_edges < 3) return 0;
    auto loadT = [](const double* p) { SE3 T; T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2]; T.r = Quat{p[3], p[4], p[5], p[6]}; return T; };
    std::vector<int> level(n_edges, 0);
    std::vector<double> chiLast(n_edges, 0.0);   // chi2 of the edge's _error at its last computeError()
    bool robust = true;
    int nBad = 0;
    SE3 T = loadT(pose_in);
    for (int it = 0; it < 4; it++) {
        T = loadT(pose_in);                                   // vSE3->setEstimate(Converter::toSE3Quat(pFrame->mTcw))
        // ---- optimizer.initializeOptimization(0); optimizer.optimize(10)
        int nActive = 0;
        for (int e = 0; e < n_edges; e++) nActive += level[e] == 0;
        auto evalErrors = [&](const SE3& Tc) {                // computeActiveErrors + activeRobustChi2
            double sum = 0;
            for (int e = 0; e < n_edges; e++) {
                if (level[e] != 0) continue;
                PLin L;
                poseLinearize(edges[e], Tc, cams[edges[e].cam], false, L);
                chiLast[e] = L.chi2;
                double r0 = L.chi2;
                if (robust) { const double d = edges[e].kind == 1 ? deltaStereo : deltaMono, dsq = d * d; if (!(L.chi2 <= dsq)) r0 = 2 * std::sqrt(L.chi2) * d - dsq; }
                sum += r0;
            }
            return sum;
        };
        if (nActive > 0) {
            double lambda = -1, ni = 2;
            int nBadLM = 0;
            for (int iter = 0; iter < 10; iter++) {
                double currentChi = evalErrors(T), tempChi = currentChi;
                const double iniChi = currentChi;
                double H[36] = {0}, bvec[6] = {0};
                for (int e = 0; e < n_edges; e++) {
                    if (level[e] != 0) continue;
                    PLin L;
                    poseLinearize(edges[e], T, cams[edges[e].cam], true, L);
                    double rho1 = 1.0;
                    if (robust) { const double d = edges[e].kind == 1 ? deltaStereo : deltaMono; if (!(L.chi2 <= d * d)) rho1 = d / std::sqrt(L.chi2); }
                    const double s = (double)edges[e].inv_sigma2;
                    for (int c = 0; c < 6; c++) {
                        double a = 0;
                        for (int r = 0; r < L.D; r++) a += L.J[r * 6 + c] * s * L.e[r];
                        bvec[c] -= rho1 * a;
                        for (int c2 = 0; c2 < 6; c2++) { double h = 0; for (int r = 0; r < L.D; r++) h += L.J[r * 6 + c] * (rho1 * s) * L.J[r * 6 + c2]; H[c2 * 6 + c] += h; }
                    }
                }
                if (iter == 0) { double md = 0; for (int j = 0; j < 6; j++) md = std::max(md, std::fabs(H[j * 7])); lambda = 1e-50 * md; ni = 2; nBadLM = 0; }
                double rho = 0;
                int qmax = 0;
                do {
                    const SE3 Tbak = T;
                    std::vector<double> S(H, H + 36), x(bvec, bvec + 6);
                    for (int j = 0; j < 6; j++) S[j * 7] += lambda;
                    const bool ok2 = cholSolve(S, 6, x);
                    if (ok2) T = se3mul(se3exp(x.data()), T);
                    tempChi = evalErrors(T);
                    if (!ok2) tempChi = 1.7976931348623157e308;
                    rho = currentChi - tempChi;
                    double scale = 0;
                    if (ok2) for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + bvec[j]);
                    scale += 1e-3;
                    rho /= scale;
                    if (rho > 0 && std::isfinite(tempChi)) {
                        double alpha = 1. - std::pow((2 * rho - 1), 3);
                        alpha = std::min(alpha, 2. / 3.);
                        lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi;
                    } else { lambda *= ni; ni *= 2; T = Tbak; }
                    qmax++;
                } while (rho < 0 && qmax < 100);
                if (qmax == 100 || rho == 0) break;
                if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
                if (nBadLM >= 3) break;
            }
        }
        // ---- classification (Optimizer.cc:1142-1246): outliers are re-evaluated at the final pose, inliers keep their last error
        nBad = 0;
        for (int e = 0; e < n_edges; e++) {
            if (outlier[e]) { PLin L; poseLinearize(edges[e], T, cams[edges[e].cam], false, L); chiLast[e] = L.chi2; }
            const float chi2 = (float)chiLast[e];
            const float th = edges[e].kind == 1 ? chi2Stereo[it] : chi2Mono[it];
            if (chi2 > th) { outlier[e] = 1; level[e] = 1; nBad++; }
            else { outlier[e] = 0; level[e] = 0; }
        }
        if (it == 2) robust = false;
        if (n_edges < 10) break;
    }
    pose_out[0] = T.t[0]; pose_out[1] = T.t[1]; pose_out[2] = T.t[2]; pose_out[3] = T.r.x; pose_out[4] = T.r.y; pose_out[5] = T.r.z; pose_out[6] = T.r.w;
    return n_edges - nBad;
}
}  // extern "C"
