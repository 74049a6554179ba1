// liba_inertial.hip — Optimizer::LocalInertialBA's optimisation (reference src/Optimizer.cc:4753-5365) on gfx950: one workgroup runs the whole
// optimizer.optimize(iterations) of one window — linearisation of the visual (EdgeMono / EdgeStereo) and inertial (EdgeInertial, EdgeGyroRW,
// EdgeAccRW) edges, Schur complement of the landmarks, dense Cholesky of the reduced system (poses 6 + velocity / gyro bias / acc bias 9 per
// optimisable key frame), g2o's Levenberg-Marquardt control — in a single launch; windows are batched over the grid.
//
// Why one workgroup per window: an inertial window is small (<= 25 optimisable key frames, a few thousand landmarks, 10^4 edges: Optimizer.cc:4758-
// 4765) and LM is a chain of dependent phases; a multi-kernel pipeline like lba_optimize's would be launch-latency bound.  Throughput comes from the
// batch.  All sums run in a fixed order (lane-strided partial sums, butterfly, waves 0..3): results are run-to-run deterministic.
//
// Float32 pieces of the reference (IMU::Preintegrated getters on cv::Mat, IMU::NormalizeRotation inside ExpSO3) follow rule R3 of DESIGN.md section 2,
// the same rule oracle/inertial_oracle.cpp states; everything else is the double arithmetic of G2oTypes.cc, expression by expression.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/orbhip.h"
#include "lds_optin.inc"
#include "dense_chol.inc"

#define LIBA_T 256
#define LIBA_IMULIN (216 + 216 + 9 + 10)   // per inertial edge: J (9x24), rho1*Omega*J (9x24), -rho1*Omega*e (9), e (9), rho1
#define LIBA_KFD (sizeof(liba_keyframe) / 8)

struct LibaArgs {
    liba_problem P;
    double lambdaInit;
    int iterations;
    unsigned char* work;
    size_t workStride;
    double* stats;
    int DRmax;
    size_t cholOff;     // LDS offset of the Cholesky scratch
};

struct Win {
    liba_keyframe* kfs; int nKf; const liba_rig* rig; double* pts; int nPts; const lba_edge* edges; int nE; const liba_imu_edge* imu; int nImu;
    double *Hpl, *Hll, *bl, *Dinv, *xl, *ptBak, *H, *S, *bv, *xp, *imuLin, *kfBak;
    int *lmStart, *kfEdges, *obsTab;
    double huberMono, huberStereo;
};

static inline size_t liba_window_bytes(const liba_problem& P, int DRmax) {
    size_t d = (size_t)P.cap_e * 18 + (size_t)P.cap_l * (9 + 3 + 9 + 3 + 3) + (size_t)DRmax * DRmax * 2 + (size_t)DRmax * 2 + (size_t)P.cap_i * LIBA_IMULIN +
               (size_t)P.cap_kf * LIBA_KFD;
    size_t i = (size_t)P.cap_l + 1 + P.cap_e + 3 + (size_t)P.cap_l * P.max_free;
    return (d * 8 + i * 4 + 255) & ~(size_t)255;
}
static __device__ __host__ inline void liba_carve(const liba_problem& P, int DRmax, unsigned char* base, Win& w) {
    double* d = (double*)base;
    w.Hpl = d; d += (size_t)P.cap_e * 18;
    w.Hll = d; d += (size_t)P.cap_l * 9;
    w.bl = d; d += (size_t)P.cap_l * 3;
    w.Dinv = d; d += (size_t)P.cap_l * 9;
    w.xl = d; d += (size_t)P.cap_l * 3;
    w.ptBak = d; d += (size_t)P.cap_l * 3;
    w.H = d; d += (size_t)DRmax * DRmax;
    w.S = d; d += (size_t)DRmax * DRmax;
    w.bv = d; d += DRmax;
    w.xp = d; d += DRmax;
    w.imuLin = d; d += (size_t)P.cap_i * LIBA_IMULIN;
    w.kfBak = d; d += (size_t)P.cap_kf * LIBA_KFD;
    int* i = (int*)d;
    w.lmStart = i; i += P.cap_l + 1;
    w.kfEdges = i; i += P.cap_e;
    w.obsTab = i;      // [cap_l][max_free]: first edge of landmark l on optimisable pose slot s, as e * 4 + n (n = 1 or 2 consecutive edges;
                       // 3 = more, or not consecutive: scan the landmark's list), -1 = not observed
}

// ---- 3x3 helpers, row-major ----
static __device__ __forceinline__ void mul33(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
static __device__ __forceinline__ void mul31(const double* A, const double* x, double* y) {
#pragma unroll
    for (int i = 0; i < 3; i++) y[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
}
static __device__ __forceinline__ void mulT31(const double* A, const double* x, double* y) {
#pragma unroll
    for (int i = 0; i < 3; i++) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}
static __device__ __forceinline__ void skew3(const double* w, double* W) {
    W[0] = 0; W[1] = -w[2]; W[2] = w[1]; W[3] = w[2]; W[4] = 0; W[5] = -w[0]; W[6] = -w[1]; W[7] = w[0]; W[8] = 0;
}
// rule R3: IMU::NormalizeRotation (cv::SVDecomp of a CV_32F matrix, ImuTypes.cc:31-37) = orthogonal polar factor, Newton's iteration in
// double on the float-rounded input, rounded to float
static __device__ void normalize_rotation_f(const double* Min, double* out) {
    double X[9];
#pragma unroll
    for (int i = 0; i < 9; i++) X[i] = (double)(float)Min[i];
    for (int it = 0; it < 8; it++) {
        const double a = X[0], b = X[1], c = X[2], d = X[3], e = X[4], f = X[5], g = X[6], h = X[7], i = X[8];
        const double C0 = e * i - f * h, C1 = -(d * i - f * g), C2 = d * h - e * g;
        const double det = a * C0 + b * C1 + c * C2, id = 1.0 / det;
        const double T[9] = {C0 * id, C1 * id, C2 * id, -(b * i - c * h) * id, (a * i - c * g) * id, -(a * h - b * g) * id,
                             (b * f - c * e) * id, -(a * f - c * d) * id, (a * e - b * d) * id};
#pragma unroll
        for (int k = 0; k < 9; k++) X[k] = 0.5 * (X[k] + T[k]);
    }
#pragma unroll
    for (int i = 0; i < 9; i++) out[i] = (double)(float)X[i];
}
static __device__ void exp_so3(const double* w, double* R) {   // G2oTypes.cc:995-1018
    const double d2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], d = sqrt(d2);
    double W[9], WW[9], res[9];
    skew3(w, W);
    mul33(W, W, WW);
    if (d < 1e-5) {
#pragma unroll
        for (int i = 0; i < 9; i++) res[i] = (i % 4 == 0 ? 1.0 : 0.0) + W[i] + 0.5 * WW[i];
    } else {
        const double sd = sin(d), cd = cos(d);
#pragma unroll
        for (int i = 0; i < 9; i++) res[i] = (i % 4 == 0 ? 1.0 : 0.0) + W[i] * sd / d + WW[i] * (1.0 - cd) / d2;
    }
    normalize_rotation_f(res, R);
}
static __device__ void log_so3(const double* R, double* w) {   // G2oTypes.cc:1020-1036
    const double tr = R[0] + R[4] + R[8];
    w[0] = (R[7] - R[5]) / 2; w[1] = (R[2] - R[6]) / 2; w[2] = (R[3] - R[1]) / 2;
    const double costheta = (tr - 1.0) * 0.5f;
    if (costheta > 1 || costheta < -1) return;
    const double theta = acos(costheta), s = sin(theta);
    if (fabs(s) < 1e-5) return;
#pragma unroll
    for (int i = 0; i < 3; i++) w[i] = theta * w[i] / s;
}
static __device__ void inv_right_jacobian_so3(const double* v, double* J) {   // G2oTypes.cc:1043-1055
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = sqrt(d2);
    double W[9], WW[9];
    skew3(v, W); mul33(W, W, WW);
    if (d < 1e-5) {
#pragma unroll
        for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double k = 1.0 / d2 - (1.0 + cos(d)) / (2.0 * d * sin(d));
#pragma unroll
    for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0 ? 1.0 : 0.0) + W[i] / 2 + WW[i] * k;
}
static __device__ void right_jacobian_so3(const double* v, double* J) {   // G2oTypes.cc:1062-1078
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = sqrt(d2);
    double W[9], WW[9];
    skew3(v, W); mul33(W, W, WW);
    if (d < 1e-5) {
#pragma unroll
        for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double sd = sin(d), cd = cos(d);
#pragma unroll
    for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0 ? 1.0 : 0.0) - W[i] * (1.0 - cd) / d2 + WW[i] * (d - sd) / (d2 * d);
}

// ---- IMU::Preintegrated getters (ImuTypes.cc:367-394) under rule R3 ----
struct DeltaBias { float g[3], a[3]; };
static __device__ __forceinline__ DeltaBias delta_bias(const liba_imu_edge& E, const double* bg, const double* ba) {
    DeltaBias d;
#pragma unroll
    for (int k = 0; k < 3; k++) { d.a[k] = (float)ba[k] - E.b[k]; d.g[k] = (float)bg[k] - E.b[3 + k]; }
    return d;
}
static __device__ __forceinline__ void fmatvec(const float* M, const float* x, float* y) {
#pragma unroll
    for (int i = 0; i < 3; i++) y[i] = (float)((double)M[i * 3] * x[0] + (double)M[i * 3 + 1] * x[1] + (double)M[i * 3 + 2] * x[2]);
}
static __device__ void exp_so3_f(const float* v, double* R) {   // float ExpSO3, ImuTypes.cc:49-61
    const float x = v[0], y = v[1], z = v[2];
    const float d2 = x * x + y * y + z * z;
    const float d = sqrtf(d2);
    const double W[9] = {0, -(double)z, (double)y, (double)z, 0, -(double)x, -(double)y, (double)x, 0};
    double WW[9];
    mul33(W, W, WW);
#pragma unroll
    for (int i = 0; i < 9; i++) WW[i] = (double)(float)WW[i];
    if (d < 1e-4f) {
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = (double)(float)((i % 4 == 0 ? 1.0 : 0.0) + W[i] + 0.5 * WW[i]);
    } else {
        const double a = sin((double)d) / (double)d, c = (1.0 - cos((double)d)) / (double)d2;
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = (double)(float)((i % 4 == 0 ? 1.0 : 0.0) + W[i] * a + WW[i] * c);
    }
}
static __device__ void get_delta_rotation(const liba_imu_edge& E, const DeltaBias& db, double* dR) {
    float v[3];
    fmatvec(E.JRg, db.g, v);
    double Ex[9], R0[9], M[9];
    exp_so3_f(v, Ex);
#pragma unroll
    for (int i = 0; i < 9; i++) R0[i] = (double)E.dR[i];
    mul33(R0, Ex, M);
    normalize_rotation_f(M, dR);
}
static __device__ __forceinline__ void get_delta_vp(const float* d0, const float* Jg, const float* Ja, const DeltaBias& db, double* out) {
    float t1[3], t2[3];
    fmatvec(Jg, db.g, t1); fmatvec(Ja, db.a, t2);
#pragma unroll
    for (int i = 0; i < 3; i++) out[i] = (double)((d0[i] + t1[i]) + t2[i]);
}

#define LIBA_G 9.81   // IMU::GRAVITY_VALUE, ImuTypes.h:40

// EdgeInertial::computeError (G2oTypes.cc:727-745); eR / dR returned for the Jacobian
static __device__ void inertial_error(const liba_imu_edge& E, const liba_keyframe& k1, const liba_keyframe& k2, double* e, double* eRout) {
    const DeltaBias db = delta_bias(E, k1.bg, k1.ba);
    double dR[9], dV[3], dP[3];
    get_delta_rotation(E, db, dR);
    get_delta_vp(E.dV, E.JVg, E.JVa, db, dV);
    get_delta_vp(E.dP, E.JPg, E.JPa, db, dP);
    const double dt = (double)E.dT, g[3] = {0, 0, -LIBA_G};
    double A[9], eR[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) A[i * 3 + j] = dR[i] * k1.Rwb[j * 3] + dR[3 + i] * k1.Rwb[j * 3 + 1] + dR[6 + i] * k1.Rwb[j * 3 + 2];
    mul33(A, k2.Rwb, eR);
    log_so3(eR, e);
    double t[3], r[3];
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = k2.v[i] - k1.v[i] - g[i] * dt;
    mulT31(k1.Rwb, t, r);
#pragma unroll
    for (int i = 0; i < 3; i++) e[3 + i] = r[i] - dV[i];
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = k2.twb[i] - k1.twb[i] - k1.v[i] * dt - g[i] * dt * dt / 2;
    mulT31(k1.Rwb, t, r);
#pragma unroll
    for (int i = 0; i < 3; i++) e[6 + i] = r[i] - dP[i];
    if (eRout) {
#pragma unroll
        for (int i = 0; i < 9; i++) eRout[i] = eR[i];
    }
}
// EdgeInertial::linearizeOplus (G2oTypes.cc:747-800): J 9 x 24 row-major, columns [VP1 6 | VV1 3 | VG1 3 | VA1 3 | VP2 6 | VV2 3]
static __device__ void inertial_jacobian(const liba_imu_edge& E, const liba_keyframe& k1, const liba_keyframe& k2, const double* er, const double* eR, double* J) {
    for (int i = 0; i < 216; i++) J[i] = 0.0;
    const DeltaBias db = delta_bias(E, k1.bg, k1.ba);
    const double dbg[3] = {(double)db.g[0], (double)db.g[1], (double)db.g[2]};
    const double dt = (double)E.dT, g[3] = {0, 0, -LIBA_G};
    double invJr[9], T2[9], t[3], r[3], S[9];
    inv_right_jacobian_so3(er, invJr);
#define PUT(r0, c0, M, s) for (int i_ = 0; i_ < 3; i_++) for (int j_ = 0; j_ < 3; j_++) J[((r0) + i_) * 24 + (c0) + j_] = (s) * (M)[i_ * 3 + j_]
    {
        double L[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L[i * 3 + j] = -(invJr[i * 3] * k2.Rwb[j * 3] + invJr[i * 3 + 1] * k2.Rwb[j * 3 + 1] + invJr[i * 3 + 2] * k2.Rwb[j * 3 + 2]);
        mul33(L, k1.Rwb, T2);
        PUT(0, 0, T2, 1.0);
    }
    for (int i = 0; i < 3; i++) t[i] = k2.v[i] - k1.v[i] - g[i] * dt;
    mulT31(k1.Rwb, t, r); skew3(r, S); PUT(3, 0, S, 1.0);
    for (int i = 0; i < 3; i++) t[i] = k2.twb[i] - k1.twb[i] - k1.v[i] * dt - 0.5 * g[i] * dt * dt;
    mulT31(k1.Rwb, t, r); skew3(r, S); PUT(6, 0, S, 1.0);
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    PUT(6, 3, I3, -1.0);
    double Rbw1[9], T1[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rbw1[i * 3 + j] = k1.Rwb[j * 3 + i];
    PUT(3, 6, Rbw1, -1.0);
    for (int i = 0; i < 9; i++) T1[i] = -Rbw1[i] * dt;
    PUT(6, 6, T1, 1.0);
    double JRg[9], M9[9], v[3], Jr[9];
    for (int i = 0; i < 9; i++) JRg[i] = (double)E.JRg[i];
    mul31(JRg, dbg, v);
    right_jacobian_so3(v, Jr);
    {
        double L[9], L2[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L[i * 3 + j] = -(invJr[i * 3] * eR[j * 3] + invJr[i * 3 + 1] * eR[j * 3 + 1] + invJr[i * 3 + 2] * eR[j * 3 + 2]);
        mul33(L, Jr, L2); mul33(L2, JRg, T2);
        PUT(0, 9, T2, 1.0);
    }
    for (int i = 0; i < 9; i++) M9[i] = (double)E.JVg[i];
    PUT(3, 9, M9, -1.0);
    for (int i = 0; i < 9; i++) M9[i] = (double)E.JPg[i];
    PUT(6, 9, M9, -1.0);
    for (int i = 0; i < 9; i++) M9[i] = (double)E.JVa[i];
    PUT(3, 12, M9, -1.0);
    for (int i = 0; i < 9; i++) M9[i] = (double)E.JPa[i];
    PUT(6, 12, M9, -1.0);
    PUT(0, 15, invJr, 1.0);
    mul33(Rbw1, k2.Rwb, T1); PUT(6, 18, T1, 1.0);
    PUT(3, 21, Rbw1, 1.0);
#undef PUT
}
static __device__ __forceinline__ void huber(const double chi2, const double delta, double* rho0, double* rho1) {   // robust_kernel_impl.cpp:44-57
    if (delta <= 0) { *rho0 = chi2; *rho1 = 1.0; return; }
    const double dsqr = delta * delta;
    if (chi2 <= dsqr) { *rho0 = chi2; *rho1 = 1.0; }
    else { const double sq = sqrt(chi2); *rho0 = 2 * sq * delta - dsqr; *rho1 = delta / sq; }
}
// chi2 of the three edges of one preintegration: out[0] EdgeInertial, [1] EdgeGyroRW, [2] EdgeAccRW
static __device__ void imu_chi(const liba_imu_edge& E, const liba_keyframe& k1, const liba_keyframe& k2, const double* e, double* out) {
    double c = 0;
    for (int r = 0; r < 9; r++) { double s = 0; for (int q = 0; q < 9; q++) s += E.info[r * 9 + q] * e[q]; c += e[r] * s; }
    out[0] = c;
    double eg[3], ea[3], cg = 0, ca = 0;
    for (int i = 0; i < 3; i++) { eg[i] = k2.bg[i] - k1.bg[i]; ea[i] = k2.ba[i] - k1.ba[i]; }
    for (int r = 0; r < 3; r++) {
        double s = 0, t = 0;
        for (int q = 0; q < 3; q++) { s += E.info_g[r * 3 + q] * eg[q]; t += E.info_a[r * 3 + q] * ea[q]; }
        cg += eg[r] * s; ca += ea[r] * t;
    }
    out[1] = cg; out[2] = ca;
}

// ---- cameras: GeometricCamera::project / projectJac (Pinhole.cpp:43-49,89-100; KannalaBrandt8.cpp:52-66,166-196) ----
static __device__ __forceinline__ void cam_project(const int model, const double* p, const double* v, double* res) {
    if (model == LBA_CAM_PINHOLE) { res[0] = p[0] * v[0] / v[2] + p[2]; res[1] = p[1] * v[1] / v[2] + p[3]; }
    else {
        const double x2_plus_y2 = v[0] * v[0] + v[1] * v[1];
        // the reference rounds through atan2f / sqrtf; float(atan2(double)) is within 1 float ulp of it
        const double theta = (double)(float)atan2((double)sqrtf((float)x2_plus_y2), (double)(float)v[2]);
        const double psi = (double)(float)atan2((double)(float)v[1], (double)(float)v[0]);
        const double theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
        const double r = theta + p[4] * theta3 + p[5] * theta5 + p[6] * theta7 + p[7] * theta9;
        res[0] = p[0] * r * cos(psi) + p[2];
        res[1] = p[1] * r * sin(psi) + p[3];
    }
}
static __device__ __forceinline__ void cam_project_jac(const int model, const double* p, const double* v, double* J) {
    if (model == LBA_CAM_PINHOLE) {
        J[0] = p[0] / v[2]; J[1] = 0; J[2] = -p[0] * v[0] / (v[2] * v[2]);
        J[3] = 0; J[4] = p[1] / v[2]; J[5] = -p[1] * v[1] / (v[2] * v[2]);
    } else {
        const double x2 = v[0] * v[0], y2 = v[1] * v[1], z2 = v[2] * v[2];
        const double r2 = x2 + y2, r = sqrt(r2), r3 = r2 * r;
        const double theta = atan2(r, v[2]);
        const double theta2 = theta * theta, theta3 = theta2 * theta, theta4 = theta2 * theta2, theta5 = theta4 * theta, theta6 = theta2 * theta4,
                     theta7 = theta6 * theta, theta8 = theta4 * theta4, theta9 = theta8 * theta;
        const double f = theta + theta3 * p[4] + theta5 * p[5] + theta7 * p[6] + theta9 * p[7];
        const double fd = 1 + 3 * p[4] * theta2 + 5 * p[5] * theta4 + 7 * p[6] * theta6 + 9 * p[7] * theta8;
        J[0] = p[0] * (fd * v[2] * x2 / (r2 * (r2 + z2)) + f * y2 / r3);
        J[3] = p[1] * (fd * v[2] * v[1] * v[0] / (r2 * (r2 + z2)) - f * v[1] * v[0] / r3);
        J[1] = p[0] * (fd * v[2] * v[1] * v[0] / (r2 * (r2 + z2)) - f * v[1] * v[0] / r3);
        J[4] = p[1] * (fd * v[2] * y2 / (r2 * (r2 + z2)) + f * x2 / r3);
        J[2] = -p[0] * fd * v[0] / (r2 + z2);
        J[5] = -p[1] * fd * v[1] / (r2 + z2);
    }
}

// EdgeMono / EdgeStereo computeError + linearizeOplus (G2oTypes.h:337-437, G2oTypes.cc:352-418)
struct VLin { int D; double e[3], A[9], B[18], chi2, rho0, rho1; bool depthPositive; };
template <bool JAC>
static __device__ void vis_linearize(const lba_edge& E, const liba_keyframe& kf, const liba_rig& rig, const double* X, const double huberMono,
                                     const double huberStereo, VLin& L) {
    const int c = E.cam;
    const double* Rcw = kf.Rcw[c]; const double* tcw = kf.tcw[c];
    double Xc[3], proj[2];
    mul31(Rcw, X, Xc);
#pragma unroll
    for (int i = 0; i < 3; i++) Xc[i] += tcw[i];
    cam_project(rig.model[c], rig.p[c], Xc, proj);
    L.D = E.kind == LBA_EDGE_STEREO ? 3 : 2;
    L.e[0] = (double)E.obs[0] - proj[0];
    L.e[1] = (double)E.obs[1] - proj[1];
    L.e[2] = 0.0;
    if (E.kind == LBA_EDGE_STEREO) { const double invZ = 1 / Xc[2]; L.e[2] = (double)E.obs[2] - (proj[0] - rig.bf * invZ); }
    L.depthPositive = (Rcw[6] * X[0] + Rcw[7] * X[1] + Rcw[8] * X[2] + tcw[2]) > 0.0;
    if (JAC) {
        double Xb[3], pj[9];
        mul31(rig.Rbc[c], Xc, Xb);
#pragma unroll
        for (int i = 0; i < 3; i++) Xb[i] += rig.tbc[c][i];
        cam_project_jac(rig.model[c], rig.p[c], Xc, pj);
        pj[6] = 0; pj[7] = 0; pj[8] = 0;
        if (E.kind == LBA_EDGE_STEREO) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2]; pj[8] += rig.bf * (1.0 / (Xc[2] * Xc[2])); }
        const double S[18] = {0, Xb[2], -Xb[1], 1, 0, 0, -Xb[2], 0, Xb[0], 0, 1, 0, Xb[1], -Xb[0], 0, 0, 0, 1};
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int k = 0; k < 3; k++) L.A[r * 3 + k] = -pj[r * 3] * Rcw[k] - pj[r * 3 + 1] * Rcw[3 + k] - pj[r * 3 + 2] * Rcw[6 + k];
            double M[3];
#pragma unroll
            for (int k = 0; k < 3; k++) M[k] = pj[r * 3] * rig.Rcb[c][k] + pj[r * 3 + 1] * rig.Rcb[c][3 + k] + pj[r * 3 + 2] * rig.Rcb[c][6 + k];
#pragma unroll
            for (int k = 0; k < 6; k++) L.B[r * 6 + k] = M[0] * S[k] + M[1] * S[6 + k] + M[2] * S[12 + k];
        }
    }
    const double s = (double)E.inv_sigma2;
    double chi2 = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) chi2 += L.e[i] * s * L.e[i];
    L.chi2 = chi2;
    huber(chi2, E.kind == LBA_EDGE_STEREO ? huberStereo : huberMono, &L.rho0, &L.rho1);
}

// ImuCamPose::Update (G2oTypes.cc:196-221); the NormalizeRotation(Rwb) of :206 discards its result
static __device__ void pose_update(liba_keyframe& k, const liba_rig& rig, const double* pu) {
    double d[3], Ex[9], Rn[9];
    mul31(k.Rwb, pu + 3, d);
    for (int i = 0; i < 3; i++) k.twb[i] += d[i];
    exp_so3(pu, Ex);
    mul33(k.Rwb, Ex, Rn);
    for (int i = 0; i < 9; i++) k.Rwb[i] = Rn[i];
    double Rbw[9], tbw[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rbw[i * 3 + j] = k.Rwb[j * 3 + i];
    mul31(Rbw, k.twb, tbw);
    for (int i = 0; i < 3; i++) tbw[i] = -tbw[i];
    for (int c = 0; c < rig.n_cams; c++) {
        mul33(rig.Rcb[c], Rbw, k.Rcw[c]);
        double t[3];
        mul31(rig.Rcb[c], tbw, t);
        for (int i = 0; i < 3; i++) k.tcw[c][i] = t[i] + rig.tcb[c][i];
    }
}

static __device__ __forceinline__ bool inv3_sym(const double* D, double* out) {   // column-major 3x3, cofactor inverse (Eigen fixed-size inverse)
    const double a = D[0], b = D[3], c = D[6], d = D[1], e = D[4], f = D[7], g = D[2], h = D[5], i = D[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C, id = 1.0 / det;
    out[0] = A * id; out[3] = -(b * i - c * h) * id; out[6] = (b * f - c * e) * id;
    out[1] = B * id; out[4] = (a * i - c * g) * id; out[7] = -(a * f - c * d) * id;
    out[2] = C * id; out[5] = -(a * h - b * g) * id; out[8] = (a * e - b * d) * id;
    return det != 0.0 && fabs(det) < 1.7e308 && det == det;
}

// workgroup-wide sum of NV doubles per thread, broadcast; fixed order (butterfly inside a wave, then waves 0..3)
template <int NV>
static __device__ __forceinline__ void wg_sum(double (&v)[NV], double* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++)
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) scratch[wave * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = ((scratch[k] + scratch[NV + k]) + scratch[2 * NV + k]) + scratch[3 * NV + k];
}

// an edge is active iff not all its vertices are fixed (sparse_optimizer.cpp:232-235)
static __device__ __forceinline__ bool imu_active(const liba_imu_edge& E, const int* hp, const int* hi) {
    return hp[E.kf1] >= 0 || hi[E.kf1] >= 0 || hp[E.kf2] >= 0 || hi[E.kf2] >= 0;
}

// SparseOptimizer::activeRobustChi2 after computeActiveErrors (sparse_optimizer.cpp:61-75, 100-114)
static __device__ double robust_chi(const Win& w, const int* hp, const int* hi, double* scratch) {
    double v[1] = {0.0};
    VLin L;
    for (int e = threadIdx.x; e < w.nE; e += LIBA_T) {
        const lba_edge E = w.edges[e];
        vis_linearize<false>(E, w.kfs[E.pose], *w.rig, w.pts + (size_t)3 * E.point, w.huberMono, w.huberStereo, L);
        v[0] += L.rho0;
    }
    for (int i = threadIdx.x; i < w.nImu; i += LIBA_T) {
        const liba_imu_edge& E = w.imu[i];
        if (!imu_active(E, hp, hi)) continue;
        double e9[9], c3[3], r0, r1;
        inertial_error(E, w.kfs[E.kf1], w.kfs[E.kf2], e9, nullptr);
        imu_chi(E, w.kfs[E.kf1], w.kfs[E.kf2], e9, c3);
        huber(c3[0], E.huber, &r0, &r1);
        v[0] += r0;
        if (hi[E.kf1] >= 0 || hi[E.kf2] >= 0) v[0] += c3[1] + c3[2];
    }
    wg_sum<1>(v, scratch);
    return v[0];
}

// column (0..23) of the EdgeInertial Jacobian -> row / column offset in the reduced system (-1: fixed vertex)
static __device__ __forceinline__ int imu_col_offset(const liba_imu_edge& E, const int* hp, const int* hi, const int c) {
    if (c < 6) return hp[E.kf1] < 0 ? -1 : hp[E.kf1] + c;
    if (c < 15) return hi[E.kf1] < 0 ? -1 : hi[E.kf1] + (c - 6);
    if (c < 21) return hp[E.kf2] < 0 ? -1 : hp[E.kf2] + (c - 15);
    return hi[E.kf2] < 0 ? -1 : hi[E.kf2] + (c - 21);
}

#ifndef LIBA_WAVES
#define LIBA_WAVES 2   // 2 workgroups per CU: the kernel is a chain of dependent global loads (latency bound), +36 % windows/s on MI355X
                       // although the allocator then spills ~1 KB per lane to scratch
#endif
static __global__ __launch_bounds__(LIBA_T, LIBA_WAVES) void k_liba_optimize(LibaArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const liba_problem& P = A.P;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Win w;
    liba_carve(P, A.DRmax, A.work + (size_t)b * A.workStride, w);
    w.kfs = P.kfs + (size_t)b * P.cap_kf; w.nKf = min(P.n_kf[b], P.cap_kf);
    w.rig = P.rigs + (size_t)b * P.rig_stride;
    w.pts = P.points + (size_t)b * P.cap_l * 3; w.nPts = min(P.n_points[b], P.cap_l);
    w.edges = P.edges + (size_t)b * P.cap_e; w.nE = min(P.n_edges[b], P.cap_e);
    w.imu = P.imu + (size_t)b * P.cap_i; w.nImu = min(P.n_imu[b], P.cap_i);
    w.huberMono = P.huber_mono; w.huberStereo = P.huber_stereo;
    double* stats = A.stats + (size_t)b * 5;
    // LDS: scratch[4*48] | ctl[16] doubles | hp[cap_kf] hi[cap_kf] freeKf[32] kfStart[33] ictl[16] ints | (16-aligned) cnt table / Cholesky scratch
    double* scratch = (double*)orb_smem;
    double* ctl = scratch + 4 * 48;
    int* hp = (int*)(ctl + 16);
    int* hi = hp + P.cap_kf;
    int* freeKf = hi + P.cap_kf;
    int* kfStart = freeKf + LIBA_MAX_FREE;
    int* ictl = kfStart + LIBA_MAX_FREE + 1;      // [0] np6 [1] DR [2] nfp [3] error flag [4] ok flag [5] loop control
    unsigned char* big = orb_smem + A.cholOff;
    int* cnt = (int*)big;                          // [LIBA_T][LIBA_MAX_FREE] counting-sort table (set-up only)

    // ---- set-up: Hessian indices (g2o: poses by vertex id = KF id first, then the V/G/A triples, ids maxKFid+3*id+1..3)
    for (int k = tid; k < w.nKf; k += LIBA_T) {
        hp[k] = w.kfs[k].pose_fixed ? -1 : 0;
        hi[k] = (w.kfs[k].has_imu && !w.kfs[k].imu_fixed) ? 0 : -1;
    }
    if (tid < 16) ictl[tid] = 0;
    __syncthreads();
    if (tid == 0) {
        int o = 0, nfp = 0;
        for (int k = 0; k < w.nKf; k++) if (hp[k] == 0) { hp[k] = o; o += 6; if (nfp < LIBA_MAX_FREE) freeKf[nfp] = k; nfp++; }
        ictl[0] = o;
        for (int k = 0; k < w.nKf; k++) if (hi[k] == 0) { hi[k] = o; o += 9; }
        ictl[1] = o; ictl[2] = nfp;
        if (nfp > LIBA_MAX_FREE || o > A.DRmax || o == 0) ictl[3] = 1;
    }
    __syncthreads();
    // landmark-major edge order is required (like lba_edge): lmStart from the boundaries
    for (int e = tid; e < w.nE; e += LIBA_T) {
        const int pe = w.edges[e].point, pp = e ? w.edges[e - 1].point : -1;
        if (pe < pp || pe >= w.nPts || w.edges[e].pose < 0 || w.edges[e].pose >= w.nKf || w.edges[e].cam < 0 || w.edges[e].cam >= w.rig->n_cams) ictl[3] = 1;
        else for (int l = pp + 1; l <= pe; l++) w.lmStart[l] = e;
    }
    {
        const int last = w.nE ? w.edges[w.nE - 1].point : -1;
        for (int l = last + 1 + tid; l <= w.nPts; l += LIBA_T) if (l >= 0) w.lmStart[l] = w.nE;
    }
    for (int i = tid; i < w.nImu; i += LIBA_T) {
        const liba_imu_edge& E = w.imu[i];
        if (E.kf1 < 0 || E.kf1 >= w.nKf || E.kf2 < 0 || E.kf2 >= w.nKf || !w.kfs[E.kf1].has_imu || !w.kfs[E.kf2].has_imu) ictl[3] = 1;
    }
    __threadfence_block();
    __syncthreads();
    if (ictl[3]) { if (tid == 0) { stats[0] = -1; stats[1] = 0; stats[2] = 0; stats[3] = 0; stats[4] = 0; } return; }
    const int np6 = ictl[0], DR = ictl[1], nfp = ictl[2];
    // edges of each optimisable pose, in edge order (stable counting sort; thread t owns a contiguous chunk)
    {
        const int chunk = (w.nE + LIBA_T - 1) / LIBA_T, e0 = min(tid * chunk, w.nE), e1 = min(e0 + chunk, w.nE);
        for (int p = 0; p < nfp; p++) cnt[tid * LIBA_MAX_FREE + p] = 0;
        for (int e = e0; e < e1; e++) { const int h = hp[w.edges[e].pose]; if (h >= 0) cnt[tid * LIBA_MAX_FREE + h / 6]++; }
        __syncthreads();
        if (tid < nfp) {
            int run = 0;
            for (int t = 0; t < LIBA_T; t++) { const int c = cnt[t * LIBA_MAX_FREE + tid]; cnt[t * LIBA_MAX_FREE + tid] = run; run += c; }
            kfStart[tid + 1] = run;
        }
        __syncthreads();
        if (tid == 0) { kfStart[0] = 0; for (int p = 0; p < nfp; p++) kfStart[p + 1] += kfStart[p]; }
        __syncthreads();
        for (int e = e0; e < e1; e++) {
            const int h = hp[w.edges[e].pose];
            if (h >= 0) { const int p = h / 6; w.kfEdges[kfStart[p] + cnt[tid * LIBA_MAX_FREE + p]++] = e; }
        }
        __threadfence_block();
        __syncthreads();
    }
    const int mf = P.max_free;
    for (int l = tid; l < w.nPts; l += LIBA_T) {
        int* tab = w.obsTab + (size_t)l * mf;
        for (int s2 = 0; s2 < nfp; s2++) tab[s2] = -1;
        for (int e = w.lmStart[l]; e < w.lmStart[l + 1]; e++) {
            const int h = hp[w.edges[e].pose];
            if (h < 0) continue;
            const int cur = tab[h / 6];
            if (cur < 0) tab[h / 6] = e * 4 + 1;
            else tab[h / 6] = ((cur & 3) == 1 && (cur >> 2) + 1 == e) ? (cur >> 2) * 4 + 2 : (cur >> 2) * 4 + 3;
        }
    }
    __threadfence_block();
    __syncthreads();

    double lambda = A.lambdaInit, ni = 2;
    int nBad = 0, it = 0, trialsTotal = 0;
    const double chi0 = robust_chi(w, hp, hi, scratch);
    // g2o recomputes the active errors at the top of every iteration; the state there is the one the last accepted trial (or the pop after a
    // rejected one) left, whose robust chi2 this kernel already holds: same function, same data, same bits
    double currentChi = chi0;
    for (it = 0; it < A.iterations; it++) {
        double tempChi = currentChi;
        const double iniChi = currentChi;
        // ================= buildSystem =================
        for (int i = tid; i < DR * DR; i += LIBA_T) w.H[i] = 0.0;
        for (int i = tid; i < DR; i += LIBA_T) w.bv[i] = 0.0;
        __syncthreads();
        // landmark side: Hll, bl, Hpl (thread = landmark; its edges are contiguous)
        for (int l = tid; l < w.nPts; l += LIBA_T) {
            double h00 = 0, h10 = 0, h20 = 0, h11 = 0, h21 = 0, h22 = 0, b0 = 0, b1 = 0, b2 = 0;
            VLin L;
            for (int e = w.lmStart[l]; e < w.lmStart[l + 1]; e++) {
                const lba_edge E = w.edges[e];
                vis_linearize<true>(E, w.kfs[E.pose], *w.rig, w.pts + (size_t)3 * l, w.huberMono, w.huberStereo, L);
                const double wt = L.rho1 * (double)E.inv_sigma2;
                const double* Aj = L.A;
                const double s0 = Aj[0] * L.e[0] + Aj[3] * L.e[1] + Aj[6] * L.e[2], s1 = Aj[1] * L.e[0] + Aj[4] * L.e[1] + Aj[7] * L.e[2],
                             s2 = Aj[2] * L.e[0] + Aj[5] * L.e[1] + Aj[8] * L.e[2];
                b0 += -wt * s0; b1 += -wt * s1; b2 += -wt * s2;
                h00 += wt * (Aj[0] * Aj[0] + Aj[3] * Aj[3] + Aj[6] * Aj[6]); h10 += wt * (Aj[1] * Aj[0] + Aj[4] * Aj[3] + Aj[7] * Aj[6]);
                h20 += wt * (Aj[2] * Aj[0] + Aj[5] * Aj[3] + Aj[8] * Aj[6]); h11 += wt * (Aj[1] * Aj[1] + Aj[4] * Aj[4] + Aj[7] * Aj[7]);
                h21 += wt * (Aj[2] * Aj[1] + Aj[5] * Aj[4] + Aj[8] * Aj[7]); h22 += wt * (Aj[2] * Aj[2] + Aj[5] * Aj[5] + Aj[8] * Aj[8]);
                if (hp[E.pose] >= 0) {
                    double* W = w.Hpl + (size_t)e * 18;   // 6x3 column-major
#pragma unroll
                    for (int c = 0; c < 3; c++)
#pragma unroll
                        for (int r = 0; r < 6; r++) W[c * 6 + r] = wt * (L.B[r] * Aj[c] + L.B[6 + r] * Aj[3 + c] + L.B[12 + r] * Aj[6 + c]);
                }
            }
            double* Hl = w.Hll + (size_t)l * 9;
            Hl[0] = h00; Hl[1] = h10; Hl[2] = h20; Hl[3] = h10; Hl[4] = h11; Hl[5] = h21; Hl[6] = h20; Hl[7] = h21; Hl[8] = h22;
            w.bl[(size_t)l * 3] = b0; w.bl[(size_t)l * 3 + 1] = b1; w.bl[(size_t)l * 3 + 2] = b2;
        }
        // pose side: wave = optimisable pose, lanes stride over its edges, butterfly reduction (21 + 6 values)
        for (int p = wave; p < nfp; p += 4) {
            const int k = freeKf[p], h = hp[k];
            double acc[27];
#pragma unroll
            for (int i = 0; i < 27; i++) acc[i] = 0.0;
            VLin L;
            for (int q = kfStart[p] + lane; q < kfStart[p + 1]; q += 64) {
                const lba_edge E = w.edges[w.kfEdges[q]];
                vis_linearize<true>(E, w.kfs[k], *w.rig, w.pts + (size_t)3 * E.point, w.huberMono, w.huberStereo, L);
                const double wt = L.rho1 * (double)E.inv_sigma2;
                int idx = 0;
#pragma unroll
                for (int c = 0; c < 6; c++)
#pragma unroll
                    for (int r = c; r < 6; r++) acc[idx++] += wt * (L.B[r] * L.B[c] + L.B[6 + r] * L.B[6 + c] + L.B[12 + r] * L.B[12 + c]);
#pragma unroll
                for (int r = 0; r < 6; r++) acc[21 + r] += -wt * (L.B[r] * L.e[0] + L.B[6 + r] * L.e[1] + L.B[12 + r] * L.e[2]);
            }
#pragma unroll
            for (int i = 0; i < 27; i++)
                for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off);
            if (lane == 0) {
                int idx = 0;
#pragma unroll
                for (int c = 0; c < 6; c++)
#pragma unroll
                    for (int r = c; r < 6; r++) { w.H[(size_t)(h + c) * DR + h + r] = acc[idx]; w.H[(size_t)(h + r) * DR + h + c] = acc[idx]; idx++; }
#pragma unroll
                for (int r = 0; r < 6; r++) w.bv[h + r] = acc[21 + r];
            }
        }
        // inertial edges: thread = edge linearises; then the quadratic forms are added edge by edge (fixed order), entries spread over threads
        for (int i = tid; i < w.nImu; i += LIBA_T) {
            const liba_imu_edge& E = w.imu[i];
            if (!imu_active(E, hp, hi)) continue;
            double* J = w.imuLin + (size_t)i * LIBA_IMULIN;
            double e9[9], eR[9], c3[3], r0, r1;
            inertial_error(E, w.kfs[E.kf1], w.kfs[E.kf2], e9, eR);
            inertial_jacobian(E, w.kfs[E.kf1], w.kfs[E.kf2], e9, eR, J);
            imu_chi(E, w.kfs[E.kf1], w.kfs[E.kf2], e9, c3);
            huber(c3[0], E.huber, &r0, &r1);
            for (int r = 0; r < 9; r++) J[441 + r] = e9[r];
            J[450] = r1;
        }
        __threadfence_block();
        __syncthreads();
        // rho1 * Omega * J and -rho1 * Omega * e, entries spread over the workgroup
        for (int t = tid; t < w.nImu * 225; t += LIBA_T) {
            const int i = t / 225, k = t - i * 225;
            const liba_imu_edge& E = w.imu[i];
            if (!imu_active(E, hp, hi)) continue;
            double* J = w.imuLin + (size_t)i * LIBA_IMULIN;
            const double r1 = J[450];
            if (k < 216) {
                const int r = k / 24, c = k - r * 24;
                double sacc = 0;
#pragma unroll
                for (int q = 0; q < 9; q++) sacc += E.info[r * 9 + q] * J[q * 24 + c];
                J[216 + k] = r1 * sacc;
            } else {
                const int r = k - 216;
                double sacc = 0;
#pragma unroll
                for (int q = 0; q < 9; q++) sacc += E.info[r * 9 + q] * J[441 + q];
                J[432 + r] = -r1 * sacc;
            }
        }
        __threadfence_block();
        __syncthreads();
        for (int i = 0; i < w.nImu; i++) {
            const liba_imu_edge& E = w.imu[i];
            if (!imu_active(E, hp, hi)) continue;      // uniform
            const double* J = w.imuLin + (size_t)i * LIBA_IMULIN;
            const double* OJ = J + 216;
            const double* Oe = OJ + 216;
            for (int t = tid; t < 24 * 24 + 24; t += LIBA_T) {
                if (t < 576) {
                    const int ca = t / 24, cb = t - ca * 24;
                    const int oa = imu_col_offset(E, hp, hi, ca), ob = imu_col_offset(E, hp, hi, cb);
                    if (oa < 0 || ob < 0) continue;
                    double s = 0;
#pragma unroll
                    for (int r = 0; r < 9; r++) s += J[r * 24 + ca] * OJ[r * 24 + cb];
                    w.H[(size_t)ob * DR + oa] += s;
                } else {
                    const int ca = t - 576, oa = imu_col_offset(E, hp, hi, ca);
                    if (oa < 0) continue;
                    double s = 0;
#pragma unroll
                    for (int r = 0; r < 9; r++) s += J[r * 24 + ca] * Oe[r];
                    w.bv[oa] += s;
                }
            }
            __threadfence_block();
            __syncthreads();
            // EdgeGyroRW / EdgeAccRW: e = b2 - b1, J = [-I, I], no robust kernel (G2oTypes.h:633-700)
            if (tid < 2 * 9) {
                const int wsel = tid / 9, rc = tid - wsel * 9, r = rc / 3, c = rc - r * 3;
                const double* Om = wsel == 0 ? E.info_g : E.info_a;
                const int o1 = hi[E.kf1] < 0 ? -1 : hi[E.kf1] + 3 + 3 * wsel, o2 = hi[E.kf2] < 0 ? -1 : hi[E.kf2] + 3 + 3 * wsel;
                if (o1 >= 0) w.H[(size_t)(o1 + c) * DR + o1 + r] += Om[r * 3 + c];
                if (o2 >= 0) w.H[(size_t)(o2 + c) * DR + o2 + r] += Om[r * 3 + c];
                if (o1 >= 0 && o2 >= 0) { w.H[(size_t)(o2 + c) * DR + o1 + r] += -Om[r * 3 + c]; w.H[(size_t)(o1 + c) * DR + o2 + r] += -Om[c * 3 + r]; }
                if (c == 0) {
                    const liba_keyframe& k1 = w.kfs[E.kf1]; const liba_keyframe& k2 = w.kfs[E.kf2];
                    double oe = 0;
                    for (int q = 0; q < 3; q++) oe += Om[r * 3 + q] * (wsel == 0 ? k2.bg[q] - k1.bg[q] : k2.ba[q] - k1.ba[q]);
                    if (o1 >= 0) w.bv[o1 + r] += oe;
                    if (o2 >= 0) w.bv[o2 + r] += -oe;
                }
            }
            __threadfence_block();
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();

        // ================= lambda trials =================
        double rhoLM = 0;
        int qmax = 0;
        do {
            // _optimizer->push(): back up every vertex that can move
            for (int t = tid; t < w.nKf * (int)LIBA_KFD; t += LIBA_T) {
                const int k = t / (int)LIBA_KFD;
                if (hp[k] >= 0 || hi[k] >= 0) w.kfBak[t] = ((const double*)w.kfs)[t];
            }
            for (int t = tid; t < w.nPts * 3; t += LIBA_T) w.ptBak[t] = w.pts[t];
            if (tid == 0) ictl[4] = 1;
            __syncthreads();
            // Dinv = (Hll + lambda I)^-1
            for (int l = tid; l < w.nPts; l += LIBA_T) {
                double D[9];
#pragma unroll
                for (int k = 0; k < 9; k++) D[k] = w.Hll[(size_t)l * 9 + k] + ((k % 4 == 0) ? lambda : 0.0);
                if (!inv3_sym(D, w.Dinv + (size_t)l * 9)) ictl[4] = 0;
            }
            // S = H + lambda I (lower triangle), right-hand side = b
            for (int t = tid; t < DR * DR; t += LIBA_T) {
                const int c = t / DR, r = t - c * DR;
                if (r >= c) w.S[t] = w.H[t] + (r == c ? lambda : 0.0);
            }
            for (int i = tid; i < DR; i += LIBA_T) w.xp[i] = w.bv[i];
            __threadfence_block();
            __syncthreads();
            // Schur complement (block_solver.hpp:381-432): wave = (pose pi >= pose pj) block; lanes stride over the edges of pi
            const int nTasks = nfp * (nfp + 1) / 2;
            for (int task = wave; task < nTasks; task += 4) {
                int pi = 0, rem = task;
                while (rem > pi) { rem -= pi + 1; pi++; }
                const int pj = rem;
                const int kj = freeKf[pj], hI = hp[freeKf[pi]], hJ = hp[kj];
                double acc[42];
#pragma unroll
                for (int i = 0; i < 42; i++) acc[i] = 0.0;
                for (int q = kfStart[pi] + lane; q < kfStart[pi + 1]; q += 64) {
                    const int e1 = w.kfEdges[q], l = w.edges[e1].point;
                    const int tv = w.obsTab[(size_t)l * mf + pj];
                    if (tv < 0) continue;             // pose pj does not observe this landmark
                    const double* Wi = w.Hpl + (size_t)e1 * 18;
                    const double* Di = w.Dinv + (size_t)l * 9;
                    double Y[18];
#pragma unroll
                    for (int c = 0; c < 3; c++)
#pragma unroll
                        for (int r = 0; r < 6; r++) Y[c * 6 + r] = Wi[r] * Di[c * 3] + Wi[6 + r] * Di[c * 3 + 1] + Wi[12 + r] * Di[c * 3 + 2];
                    {
                        const int n2 = tv & 3;
                        const int eb = n2 == 3 ? w.lmStart[l] : (tv >> 2), ee = n2 == 3 ? w.lmStart[l + 1] : (tv >> 2) + n2;
                        for (int e2 = eb; e2 < ee; e2++) {
                            if (n2 == 3 && w.edges[e2].pose != kj) continue;
                            const double* Wj = w.Hpl + (size_t)e2 * 18;
#pragma unroll
                            for (int c = 0; c < 6; c++)
#pragma unroll
                                for (int r = 0; r < 6; r++) acc[c * 6 + r] += Y[r] * Wj[c] + Y[6 + r] * Wj[6 + c] + Y[12 + r] * Wj[12 + c];
                        }
                    }
                    if (pi == pj) {   // _bschur: b_p - Hpl Dinv bl
                        const double* bl = w.bl + (size_t)l * 3;
#pragma unroll
                        for (int r = 0; r < 6; r++) acc[36 + r] += Y[r] * bl[0] + Y[6 + r] * bl[1] + Y[12 + r] * bl[2];
                    }
                }
#pragma unroll
                for (int i = 0; i < 42; i++)
                    for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off);
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < 6; c++)
#pragma unroll
                        for (int r = 0; r < 6; r++)
                            if (pi != pj || r >= c) w.S[(size_t)(hJ + c) * DR + hI + r] -= acc[c * 6 + r];
                    if (pi == pj) {
#pragma unroll
                        for (int r = 0; r < 6; r++) w.xp[hI + r] -= acc[36 + r];
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
            bool ok2 = ictl[4] != 0;
            if (ok2) ok2 = wg_chol_solve(w.S, DR, DR, w.xp, big);
            __syncthreads();
            double sc[1] = {0.0};
            if (ok2) {
                // back-substitution (block_solver.hpp:461-481) + point update
                for (int l = tid; l < w.nPts; l += LIBA_T) {
                    double cl[3] = {w.bl[(size_t)l * 3], w.bl[(size_t)l * 3 + 1], w.bl[(size_t)l * 3 + 2]};
                    for (int e1 = w.lmStart[l]; e1 < w.lmStart[l + 1]; e1++) {
                        const int h1 = hp[w.edges[e1].pose];
                        if (h1 < 0) continue;
                        const double* Bi = w.Hpl + (size_t)e1 * 18;
#pragma unroll
                        for (int c = 0; c < 3; c++)
#pragma unroll
                            for (int r = 0; r < 6; r++) cl[c] -= Bi[c * 6 + r] * w.xp[h1 + r];
                    }
                    const double* Di = w.Dinv + (size_t)l * 9;
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        const double x = Di[r] * cl[0] + Di[3 + r] * cl[1] + Di[6 + r] * cl[2];
                        sc[0] += x * (lambda * x + w.bl[(size_t)l * 3 + r]);
                        w.pts[(size_t)l * 3 + r] += x;
                    }
                }
                for (int i = tid; i < DR; i += LIBA_T) sc[0] += w.xp[i] * (lambda * w.xp[i] + w.bv[i]);
                // vertex updates: ImuCamPose::Update for the poses, += for velocity / biases
                for (int k = tid; k < w.nKf; k += LIBA_T) {
                    if (hp[k] >= 0) pose_update(w.kfs[k], *w.rig, w.xp + hp[k]);
                    if (hi[k] >= 0) {
                        for (int i = 0; i < 3; i++) { w.kfs[k].v[i] += w.xp[hi[k] + i]; w.kfs[k].bg[i] += w.xp[hi[k] + 3 + i]; w.kfs[k].ba[i] += w.xp[hi[k] + 6 + i]; }
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
            wg_sum<1>(sc, scratch);
            tempChi = robust_chi(w, hp, hi, scratch);
            if (!ok2) tempChi = 1.7976931348623157e308;
            rhoLM = currentChi - tempChi;
            const double scale = (ok2 ? sc[0] : 0.0) + 1e-3;    // computeScale (optimization_algorithm_levenberg.cpp:196-208)
            rhoLM /= scale;
            bool accept = rhoLM > 0 && tempChi < 1.7e308 && tempChi == tempChi;
            if (accept) {
                double alpha = 1. - pow((2 * rhoLM - 1), 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                for (int t = tid; t < w.nKf * (int)LIBA_KFD; t += LIBA_T) {
                    const int k = t / (int)LIBA_KFD;
                    if (hp[k] >= 0 || hi[k] >= 0) ((double*)w.kfs)[t] = w.kfBak[t];
                }
                for (int t = tid; t < w.nPts * 3; t += LIBA_T) w.pts[t] = w.ptBak[t];
                __threadfence_block();
                __syncthreads();
            }
            qmax++; trialsTotal++;
        } while (rhoLM < 0 && qmax < 100);
        if (qmax == 100 || rhoLM == 0) { it++; break; }
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) { it++; break; }
    }
    const double chiEnd = currentChi;
    if (tid == 0) { stats[0] = it; stats[1] = chiEnd; stats[2] = lambda; stats[3] = trialsTotal; stats[4] = chi0; }
    (void)np6;
}

struct ErrArgs { liba_problem P; double* visChi2; uint8_t* visDepth; double* imuChi2; double* robustSum; };

static __global__ __launch_bounds__(LIBA_T) void k_liba_errors(ErrArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    const liba_problem& P = A.P;
    const int b = blockIdx.x, tid = threadIdx.x;
    Win w;
    w.kfs = P.kfs + (size_t)b * P.cap_kf; w.nKf = min(P.n_kf[b], P.cap_kf);
    w.rig = P.rigs + (size_t)b * P.rig_stride;
    w.pts = P.points + (size_t)b * P.cap_l * 3; w.nPts = min(P.n_points[b], P.cap_l);
    w.edges = P.edges + (size_t)b * P.cap_e; w.nE = min(P.n_edges[b], P.cap_e);
    w.imu = P.imu + (size_t)b * P.cap_i; w.nImu = min(P.n_imu[b], P.cap_i);
    w.huberMono = P.huber_mono; w.huberStereo = P.huber_stereo;
    double* scratch = (double*)orb_smem;
    int* hp = (int*)(scratch + 4 * 48);
    int* hi = hp + P.cap_kf;
    for (int k = tid; k < w.nKf; k += LIBA_T) {
        hp[k] = w.kfs[k].pose_fixed ? -1 : 0;
        hi[k] = (w.kfs[k].has_imu && !w.kfs[k].imu_fixed) ? 0 : -1;
    }
    __syncthreads();
    VLin L;
    for (int e = tid; e < w.nE; e += LIBA_T) {
        const lba_edge E = w.edges[e];
        vis_linearize<false>(E, w.kfs[E.pose], *w.rig, w.pts + (size_t)3 * E.point, w.huberMono, w.huberStereo, L);
        if (A.visChi2) A.visChi2[(size_t)b * P.cap_e + e] = L.chi2;
        if (A.visDepth) A.visDepth[(size_t)b * P.cap_e + e] = L.depthPositive ? 1 : 0;
    }
    if (A.imuChi2)
        for (int i = tid; i < w.nImu; i += LIBA_T) {
            const liba_imu_edge& E = w.imu[i];
            double e9[9], c3[3];
            inertial_error(E, w.kfs[E.kf1], w.kfs[E.kf2], e9, nullptr);
            imu_chi(E, w.kfs[E.kf1], w.kfs[E.kf2], e9, c3);
            for (int k = 0; k < 3; k++) A.imuChi2[((size_t)b * P.cap_i + i) * 3 + k] = c3[k];
        }
    if (A.robustSum) {
        const double r = robust_chi(w, hp, hi, scratch);
        if (tid == 0) A.robustSum[b] = r;
    }
}

static bool liba_valid(const liba_problem* p, int batch) {
    return p && batch >= 0 && p->kfs && p->n_kf && p->rigs && p->points && p->n_points && p->edges && p->n_edges && p->n_imu && (p->imu || p->cap_i == 0) &&
           p->cap_kf > 0 && p->cap_l > 0 && p->cap_e > 0 && p->cap_i >= 0 && p->rig_stride >= 0 && p->max_free > 0 && p->max_free <= LIBA_MAX_FREE;
}
extern "C" size_t liba_workspace_bytes(const liba_problem* prob, int batch) {
    if (!liba_valid(prob, batch)) return 0;
    return liba_window_bytes(*prob, 15 * prob->max_free) * (size_t)std::max(batch, 1);
}

extern "C" int liba_optimize(const liba_problem* prob, int batch, double lambda_init, int iterations, void* d_workspace, double* d_stats, void* stream) {
    if (!liba_valid(prob, batch) || !d_workspace || !d_stats || iterations < 0 || !(lambda_init > 0)) return ORB_E_INVALID;
    if (batch == 0) return ORB_OK;
    LibaArgs A;
    A.P = *prob; A.lambdaInit = lambda_init; A.iterations = iterations; A.work = (unsigned char*)d_workspace;
    A.DRmax = 15 * prob->max_free;
    A.workStride = liba_window_bytes(*prob, A.DRmax);
    A.stats = d_stats;
    const size_t head = ((size_t)(4 * 48 + 16) * 8 + (size_t)(2 * prob->cap_kf + LIBA_MAX_FREE + LIBA_MAX_FREE + 1 + 16) * 4 + 15) & ~(size_t)15;
    A.cholOff = head;
    const size_t smem = head + std::max(wg_chol_smem_bytes(A.DRmax), (size_t)LIBA_T * LIBA_MAX_FREE * 4);
    if (smem > 160 * 1024) return ORB_E_INVALID;
    if (orb_lds_optin((const void*)k_liba_optimize, smem) != ORB_OK) return ORB_E_HIP;
    hipLaunchKernelGGL(k_liba_optimize, dim3(batch), dim3(LIBA_T), smem, (hipStream_t)stream, A);
    return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP;
}

extern "C" int liba_compute_errors(const liba_problem* prob, int batch, double* d_vis_chi2, uint8_t* d_vis_depth_pos, double* d_imu_chi2,
                                   double* d_robust_sum, void* stream) {
    if (!liba_valid(prob, batch)) return ORB_E_INVALID;
    if (batch == 0) return ORB_OK;
    ErrArgs A{*prob, d_vis_chi2, d_vis_depth_pos, d_imu_chi2, d_robust_sum};
    const size_t smem = (size_t)4 * 48 * 8 + (size_t)2 * prob->cap_kf * 4 + 16;
    hipLaunchKernelGGL(k_liba_errors, dim3(batch), dim3(LIBA_T), smem, (hipStream_t)stream, A);
    return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP;
}

// ================================================================================================================================================
// Optimizer::PoseInertialOptimizationLastKeyFrame / ...LastFrame (reference src/Optimizer.cc:7665-8067, :8068-8415): the per-frame optimisation of the
// inertial tracking modes.  One wave per frame runs the whole schedule in a single launch: 4 rounds x 10 Gauss-Newton iterations
// (OptimizationAlgorithmGaussNewton + LinearSolverDense, :7669-7672) with EdgeMonoOnlyPose / EdgeStereoOnlyPose, EdgeInertial, EdgeGyroRW, EdgeAccRW;
// chi2 re-classification between rounds (bClose rule, Huber dropped for the last round); the < 30 inliers recovery pass; the Hessian of the final state
// for the next frame's prior.
//   LASTF = false: frame pose / velocity / biases free (15 unknowns), the last key frame fixed; output H = the 15x15 Hessian.
//   LASTF = true : the previous frame's four vertices are free too (30 unknowns, [frame | previous]) and tied down by EdgePriorPoseImu (Huber 5);
//                  chi2Mono = 5.991 in every round; output H = the 30x30 Hessian marginalised over the previous frame (Optimizer::Marginalize,
//                  :5366-5450: JacobiSVD pseudo-inverse with singular values <= 1e-6 dropped; here a one-sided Jacobi SVD spread over the lanes).
// Lanes stride over the reprojection edges (butterfly sums in a fixed order); the inertial / prior blocks and the dense Cholesky are spread over the lanes
// through LDS.
// ================================================================================================================================================
struct PoseInertialArgs {
    liba_keyframe* frames; liba_keyframe* prevs; const liba_rig* rigs; int rigStride;
    const pose_edge* edges; const int32_t* nEdges; int capE; const liba_imu_edge* imu; const liba_prior* priors; int recInit;
    uint8_t* outlier; double* H; int32_t* nGood;
};
#define PIK_WAVE_SYNC() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

template <bool JAC>
static __device__ __forceinline__ void pose_edge_linearize(const pose_edge& E, const liba_keyframe& kf, const liba_rig& rig, VLin& L) {
    lba_edge V;
    V.pose = 0; V.point = 0; V.kind = (int16_t)(E.kind & 0xFF); V.cam = E.cam; V.obs[0] = E.obs[0]; V.obs[1] = E.obs[1]; V.obs[2] = E.obs[2]; V.inv_sigma2 = E.inv_sigma2;
    const double X[3] = {(double)E.xw[0], (double)E.xw[1], (double)E.xw[2]};
    vis_linearize<JAC>(V, kf, rig, X, 0.0, 0.0, L);
}
// EdgePriorPoseImu::computeError / linearizeOplus (G2oTypes.cc:935-968): e[15], J 15x15 row-major over the previous frame's [pose 6 | v 3 | bg 3 | ba 3]
static __device__ void prior_linearize(const liba_prior& C, const liba_keyframe& P, double* e, double* J) {
    double M[9], er[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[i * 3 + j] = C.Rwb[i] * P.Rwb[j] + C.Rwb[3 + i] * P.Rwb[3 + j] + C.Rwb[6 + i] * P.Rwb[6 + j];
    log_so3(M, er);
    const double d[3] = {P.twb[0] - C.twb[0], P.twb[1] - C.twb[1], P.twb[2] - C.twb[2]};
    double et[3];
    mulT31(C.Rwb, d, et);
    for (int i = 0; i < 3; i++) { e[i] = er[i]; e[3 + i] = et[i]; e[6 + i] = P.v[i] - C.vwb[i]; e[9 + i] = P.bg[i] - C.bg[i]; e[12 + i] = P.ba[i] - C.ba[i]; }
    for (int i = 0; i < 225; i++) J[i] = 0.0;
    double iJ[9];
    inv_right_jacobian_so3(er, iJ);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { J[r * 15 + c] = iJ[r * 3 + c]; J[(3 + r) * 15 + 3 + c] = M[r * 3 + c]; }
    for (int k = 6; k < 15; k++) J[k * 15 + k] = 1.0;
}

template <bool LASTF>
static __global__ __launch_bounds__(64) void k_pose_inertial(PoseInertialArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    constexpr int N = LASTF ? 30 : 15;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int ne = min(A.nEdges[b], A.capE);
    // LDS: F | Pv (liba_keyframe) | Hs[N*N] | xs[N] | J24[216] | OJ[216] | e9[9] | Oe[9] | Jp[225] | HJp[225] | ep[15] | Hep[15] | ctl[2] |
    //      chiLast[capE] | level[capE] u8 | outl[capE] u8
    const size_t kfb = (sizeof(liba_keyframe) + 15) & ~(size_t)15;
    liba_keyframe& F = *(liba_keyframe*)orb_smem;
    liba_keyframe& Pv = *(liba_keyframe*)(orb_smem + kfb);
    double* Hs = (double*)(orb_smem + 2 * kfb);
    double* xs = Hs + N * N;
    double* J24 = xs + N;
    double* OJ = J24 + 216;
    double* e9s = OJ + 216;
    double* Oes = e9s + 9;
    double* Jp = Oes + 9;
    double* HJp = Jp + 225;
    double* eps = HJp + 225;
    double* Heps = eps + 15;
    double* ctl = Heps + 15;
    double* chiLast = ctl + 2;
    uint8_t* level = (uint8_t*)(chiLast + A.capE);
    uint8_t* outl = level + A.capE;
    const liba_rig& rig = A.rigs[(size_t)b * A.rigStride];
    const liba_imu_edge& E = A.imu[b];
    const pose_edge* edges = A.edges + (size_t)b * A.capE;
    for (int t = lane; t < (int)LIBA_KFD; t += 64) { ((double*)&F)[t] = ((const double*)&A.frames[b])[t]; ((double*)&Pv)[t] = ((const double*)&A.prevs[b])[t]; }
    for (int e = lane; e < ne; e += 64) { chiLast[e] = 0.0; level[e] = 0; outl[e] = 0; }
    PIK_WAVE_SYNC()
    const double thMono = (double)sqrtf(5.991f), thStereo = (double)sqrtf(7.815f);
    const float chi2MonoKF[4] = {12.f, 7.5f, 5.991f, 5.991f}, chi2Stereo[4] = {15.6f, 9.8f, 7.815f, 7.815f};
    int nBad = 0, nInliers = 0;
    bool robust = true;
    VLin L;
    // unknown index -> column of the EdgeInertial Jacobian ([VP1 6 | VV1 3 | VG1 3 | VA1 3 | VP2 6 | VV2 3], vertex 1 = key frame / previous frame) or -1
    auto icol = [](const int u) -> int { return u < 9 ? 15 + u : (LASTF && u >= 15 ? u - 15 : -1); };
    // J of EdgeInertial, Omega*J, Omega*e in LDS; (LASTF) J of the prior edge, H_prior*J, H_prior*e and its Huber weight in ctl[0]
    auto edge_blocks = [&](const bool weighted) {
        if (lane == 0) {
            double e9[9], eR[9];
            inertial_error(E, Pv, F, e9, eR);
            inertial_jacobian(E, Pv, F, e9, eR, J24);
            for (int r = 0; r < 9; r++) e9s[r] = e9[r];
        }
        if (LASTF && lane == 1) prior_linearize(A.priors[b], Pv, eps, Jp);
        PIK_WAVE_SYNC()
        for (int t = lane; t < 225; t += 64) {
            if (t < 216) { const int r = t / 24, c = t - r * 24; double s = 0; for (int q = 0; q < 9; q++) s += E.info[r * 9 + q] * J24[q * 24 + c]; OJ[t] = s; }
            else { const int r = t - 216; double s = 0; for (int q = 0; q < 9; q++) s += E.info[r * 9 + q] * e9s[q]; Oes[r] = s; }
        }
        if (LASTF) {
            const double* Hp = A.priors[b].H;
            for (int t = lane; t < 240; t += 64) {
                if (t < 225) { const int r = t / 15, c = t - r * 15; double s = 0; for (int q = 0; q < 15; q++) s += Hp[r * 15 + q] * Jp[q * 15 + c]; HJp[t] = s; }
                else { const int r = t - 225; double s = 0; for (int q = 0; q < 15; q++) s += Hp[r * 15 + q] * eps[q]; Heps[r] = s; }
            }
            PIK_WAVE_SYNC()
            if (lane == 0) {
                double chi = 0;
                for (int r = 0; r < 15; r++) chi += eps[r] * Heps[r];
                ctl[0] = (weighted && chi > 25.0) ? 5.0 / sqrt(chi) : 1.0;   // RobustKernelHuber, delta 5 (:8251)
            }
        }
        PIK_WAVE_SYNC()
    };
    // entry (r, c) of the system matrix without the reprojection part
    auto h_entry = [&](const int r, const int c) -> double {
        double v = 0.0;
        const int ar = icol(r), ac = icol(c);
        if (ar >= 0 && ac >= 0) { double s = 0; for (int q = 0; q < 9; q++) s += J24[q * 24 + ar] * OJ[q * 24 + ac]; v += s; }
        // random walks: e = b_frame - b_other, J_frame = +I (unknowns 9..14), J_other = -I (unknowns 24..29, LASTF only)
        const int fr = (r >= 9 && r < 15) ? r - 9 : ((LASTF && r >= 24) ? r - 24 : -1), fc = (c >= 9 && c < 15) ? c - 9 : ((LASTF && c >= 24) ? c - 24 : -1);
        if (fr >= 0 && fc >= 0 && fr / 3 == fc / 3) {
            const double* Om = fr < 3 ? E.info_g : E.info_a;
            const bool rF = r < 15, cF = c < 15;
            if (rF == cF) v += Om[(fr % 3) * 3 + fc % 3];
            else if (rF) v += -Om[(fr % 3) * 3 + fc % 3];          // (frame, other) block = -Omega
            else v += -Om[(fc % 3) * 3 + fr % 3];                  // (other, frame) block = -Omega^T
        }
        if (LASTF && r >= 15 && c >= 15) { double s = 0; for (int q = 0; q < 15; q++) s += Jp[q * 15 + r - 15] * HJp[q * 15 + c - 15]; v += ctl[0] * s; }
        return v;
    };
    auto pose_acc = [&](const double (&acc)[27], const int r, const int c) -> double {
        const int lo = r < c ? r : c, hi = r < c ? c : r;   // acc is packed by columns of the lower triangle: (c, r >= c)
        int idx = 0;
        for (int cc = 0; cc < lo; cc++) idx += 6 - cc;
        idx += hi - lo;
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < 21; k++) if (k == idx) a = acc[k];
        return a;
    };
    for (int it = 0; it < 4; it++) {
        for (int gn = 0; gn < 10; gn++) {
            // ---- reprojection edges of level 0: H_pose (21 unique) + b_pose (6), fixed-order butterfly
            double acc[27];
#pragma unroll
            for (int i = 0; i < 27; i++) acc[i] = 0.0;
            for (int e = lane; e < ne; e += 64) {
                if (level[e]) continue;
                const pose_edge PE = edges[e];
                pose_edge_linearize<true>(PE, F, rig, L);
                chiLast[e] = L.chi2;
                double rho1 = 1.0;
                if (robust) { const double d = (PE.kind & 0xFF) == LBA_EDGE_STEREO ? thStereo : thMono; if (L.chi2 > d * d) rho1 = d / sqrt(L.chi2); }
                const double wt = rho1 * (double)PE.inv_sigma2;
                int idx = 0;
#pragma unroll
                for (int c = 0; c < 6; c++)
#pragma unroll
                    for (int r = c; r < 6; r++) acc[idx++] += wt * (L.B[r] * L.B[c] + L.B[6 + r] * L.B[6 + c] + L.B[12 + r] * L.B[12 + c]);
#pragma unroll
                for (int r = 0; r < 6; r++) acc[21 + r] += -wt * (L.B[r] * L.e[0] + L.B[6 + r] * L.e[1] + L.B[12 + r] * L.e[2]);
            }
#pragma unroll
            for (int i = 0; i < 27; i++)
                for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off);
            edge_blocks(true);
            // ---- assemble H (column-major N x N, full) and b
            for (int t = lane; t < N * N; t += 64) {
                const int c = t / N, r = t - c * N;
                Hs[t] = h_entry(r, c) + ((r < 6 && c < 6) ? pose_acc(acc, r, c) : 0.0);
            }
            if (lane < N) {
                const int u = lane;
                double v = 0.0;
                if (u < 6) {
#pragma unroll
                    for (int k = 0; k < 6; k++) if (k == u) v = acc[21 + k];
                }
                const int au = icol(u);
                if (au >= 0) { double s = 0; for (int q = 0; q < 9; q++) s += J24[q * 24 + au] * Oes[q]; v += -s; }
                const int fu = (u >= 9 && u < 15) ? u - 9 : ((LASTF && u >= 24) ? u - 24 : -1);
                if (fu >= 0) {
                    const double* Om = fu < 3 ? E.info_g : E.info_a;
                    double oe = 0;
                    for (int q = 0; q < 3; q++) oe += Om[(fu % 3) * 3 + q] * (fu < 3 ? F.bg[q] - Pv.bg[q] : F.ba[q] - Pv.ba[q]);
                    v += u < 15 ? -oe : oe;
                }
                if (LASTF && u >= 15) { double s = 0; for (int q = 0; q < 15; q++) s += Jp[q * 15 + u - 15] * Heps[q]; v += -ctl[0] * s; }
                xs[u] = v;
            }
            PIK_WAVE_SYNC()
            // ---- dense Cholesky (lower triangle of the column-major Hs), columns spread over the lanes
            bool ok = true;
            for (int k = 0; k < N; k++) {
                const double dkk = Hs[k * N + k];
                if (!(dkk > 0) || !(dkk < 1.7e308)) { ok = false; break; }   // uniform
                const double sq = sqrt(dkk);
                PIK_WAVE_SYNC()
                if (lane >= k && lane < N) Hs[k * N + lane] = lane == k ? sq : Hs[k * N + lane] / sq;
                PIK_WAVE_SYNC()
                for (int t = lane; t < N * N; t += 64) {
                    const int j = t / N, i = t - j * N;
                    if (j > k && i >= j) Hs[j * N + i] -= Hs[k * N + i] * Hs[k * N + j];
                }
                PIK_WAVE_SYNC()
            }
            if (!ok) break;
            if (lane == 0) {
                for (int i = 0; i < N; i++) { double v = xs[i]; for (int k = 0; k < i; k++) v -= Hs[k * N + i] * xs[k]; xs[i] = v / Hs[i * N + i]; }
                for (int i = N - 1; i >= 0; i--) { double v = xs[i]; for (int k = i + 1; k < N; k++) v -= Hs[i * N + k] * xs[k]; xs[i] = v / Hs[i * N + i]; }
            }
            PIK_WAVE_SYNC()
            if (lane == 0) {
                double x[15];
                for (int i = 0; i < 15; i++) x[i] = xs[i];
                pose_update(F, rig, x);
                for (int i = 0; i < 3; i++) { F.v[i] += x[6 + i]; F.bg[i] += x[9 + i]; F.ba[i] += x[12 + i]; }
            }
            if (LASTF && lane == 1) {
                double x[15];
                for (int i = 0; i < 15; i++) x[i] = xs[15 + i];
                pose_update(Pv, rig, x);
                for (int i = 0; i < 3; i++) { Pv.v[i] += x[6 + i]; Pv.bg[i] += x[9 + i]; Pv.ba[i] += x[12 + i]; }
            }
            PIK_WAVE_SYNC()
        }
        // ---- classification (:7838-7896 / :8267-8330): chi2() of an edge that was optimised is the value of the last computeActiveErrors (state before the
        //      final update); outlier edges are re-evaluated at the current state; isDepthPositive() always reads the current state
        const float chi2Mono = LASTF ? 5.991f : chi2MonoKF[it];
        const float chi2close = 1.5f * chi2Mono;
        int bad = 0, good = 0;
        for (int e0 = 0; e0 < ne; e0 += 64) {
            const int e = e0 + lane;
            bool isBad = false;
            const bool act = e < ne;
            if (act) {
                const pose_edge PE = edges[e];
                pose_edge_linearize<false>(PE, F, rig, L);
                if (outl[e]) chiLast[e] = L.chi2;
                const float chi2 = (float)chiLast[e];
                if ((PE.kind & 0xFF) != LBA_EDGE_STEREO) {
                    const bool bClose = (PE.kind & 0x100) != 0;
                    isBad = (chi2 > chi2Mono && !bClose) || (bClose && chi2 > chi2close) || !L.depthPositive;
                } else isBad = chi2 > chi2Stereo[it];
                outl[e] = isBad; level[e] = isBad;
            }
            bad += __popcll(__ballot(act && isBad));
            good += __popcll(__ballot(act && !isBad));
        }
        nBad = bad; nInliers = good;
        if (it == 2) robust = false;
        PIK_WAVE_SYNC()
        if (ne + (LASTF ? 4 : 3) < 10) break;
    }
    if (nInliers < 30 && !A.recInit) {   // :7904-7934 / :8332-8362
        int bad = 0;
        for (int e0 = 0; e0 < ne; e0 += 64) {
            const int e = e0 + lane;
            bool stillBad = false;
            if (e < ne) {
                const pose_edge PE = edges[e];
                pose_edge_linearize<false>(PE, F, rig, L);
                if ((float)L.chi2 < ((PE.kind & 0xFF) == LBA_EDGE_STEREO ? 24.f : 18.f)) outl[e] = 0; else stillBad = true;
            }
            bad += __popcll(__ballot(stillBad));
        }
        nBad = bad;
        PIK_WAVE_SYNC()
    }
    // ---- the Hessian of the final state: information() without robust weights, inlier reprojection edges only (:8040-8062 / :8366-8400)
    {
        double acc[27];
#pragma unroll
        for (int i = 0; i < 27; i++) acc[i] = 0.0;
        for (int e = lane; e < ne; e += 64) {
            if (outl[e]) continue;
            const pose_edge PE = edges[e];
            pose_edge_linearize<true>(PE, F, rig, L);
            const double wt = (double)PE.inv_sigma2;
            int idx = 0;
#pragma unroll
            for (int c = 0; c < 6; c++)
#pragma unroll
                for (int r = c; r < 6; r++) acc[idx++] += wt * (L.B[r] * L.B[c] + L.B[6 + r] * L.B[6 + c] + L.B[12 + r] * L.B[12 + c]);
        }
#pragma unroll
        for (int i = 0; i < 21; i++)
            for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off);
        edge_blocks(false);
        double* Hout = A.H + (size_t)b * 225;
        if (!LASTF) {
            for (int t = lane; t < 225; t += 64) {
                const int r = t / 15, c = t - r * 15;   // row-major output
                Hout[t] = h_entry(r, c) + ((r < 6 && c < 6) ? pose_acc(acc, r, c) : 0.0);
            }
        } else {
            for (int t = lane; t < N * N; t += 64) {
                const int c = t / N, r = t - c * N;
                Hs[t] = h_entry(r, c) + ((r < 6 && c < 6) ? pose_acc(acc, r, c) : 0.0);
            }
            PIK_WAVE_SYNC()
            // Optimizer::Marginalize(H, 0, 14): H_ff - H_fp * pinv(H_pp) * H_pf.  pinv by a one-sided Jacobi SVD of H_pp: lane k < 15 owns row k of U
            // (columns converge to u_i * sigma_i) and of V; the (p, q) rotations run in a fixed order, <= 30 sweeps, stop when every column pair is orthogonal
            double* U = Jp;      // 15 x 15 row-major (the prior blocks are dead now)
            double* V = HJp;
            for (int t = lane; t < 225; t += 64) { const int r = t / 15, c = t - r * 15; U[t] = Hs[(15 + c) * N + 15 + r]; V[t] = r == c ? 1.0 : 0.0; }
            PIK_WAVE_SYNC()
            for (int sweep = 0; sweep < 30; sweep++) {
                double off = 0.0;
                for (int p = 0; p < 14; p++)
                    for (int q = p + 1; q < 15; q++) {
                        double al = 0, be = 0, ga = 0;
                        if (lane < 15) { const double up = U[lane * 15 + p], uq = U[lane * 15 + q]; al = up * up; be = uq * uq; ga = up * uq; }
                        // fixed-order sum over rows 0..14 (the oracle adds them in the same order)
                        double sal = 0, sbe = 0, sga = 0;
                        for (int k = 0; k < 15; k++) { sal += __shfl(al, k); sbe += __shfl(be, k); sga += __shfl(ga, k); }
                        if (sga == 0.0) continue;
                        off = fmax(off, fabs(sga) / sqrt(sal * sbe + 1e-300));
                        const double zeta = (sbe - sal) / (2.0 * sga);
                        const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                        if (lane < 15) {
                            const double up = U[lane * 15 + p], uq = U[lane * 15 + q]; U[lane * 15 + p] = cs * up - sn * uq; U[lane * 15 + q] = sn * up + cs * uq;
                            const double vp = V[lane * 15 + p], vq = V[lane * 15 + q]; V[lane * 15 + p] = cs * vp - sn * vq; V[lane * 15 + q] = sn * vp + cs * vq;
                        }
                    }
                if (off < 1e-15) break;
            }
            PIK_WAVE_SYNC()
            // s_i^2 = |U[:, i]|^2 ; pinv = sum_i V[:, i] U[:, i]^T / s_i^2 over s_i > 1e-6 ; T = pinv * H_pf ; out = H_ff - H_fp * T
            if (lane < 15) { double s2 = 0; for (int k = 0; k < 15; k++) s2 += U[k * 15 + lane] * U[k * 15 + lane]; eps[lane] = s2; }
            PIK_WAVE_SYNC()
            double* inv = J24;   // 225 doubles over J24 | OJ (contiguous, dead now)
            for (int t = lane; t < 225; t += 64) {
                const int r = t / 15, c = t - r * 15;
                double s = 0;
                for (int i = 0; i < 15; i++) { const double s2 = eps[i]; if (sqrt(s2) > 1e-6) s += V[r * 15 + i] * U[c * 15 + i] / s2; }
                inv[t] = s;
            }
            PIK_WAVE_SYNC()
            for (int t = lane; t < 225; t += 64) {
                const int r = t / 15, c = t - r * 15;
                double s = 0;
                for (int k = 0; k < 15; k++) { double tk = 0; for (int m = 0; m < 15; m++) tk += inv[k * 15 + m] * Hs[c * N + 15 + m]; s += Hs[(15 + k) * N + r] * tk; }
                Hout[t] = Hs[c * N + r] - s;
            }
        }
    }
    for (int t = lane; t < (int)LIBA_KFD; t += 64) { ((double*)&A.frames[b])[t] = ((const double*)&F)[t]; if (LASTF) ((double*)&A.prevs[b])[t] = ((const double*)&Pv)[t]; }
    for (int e = lane; e < A.capE; e += 64) A.outlier[(size_t)b * A.capE + e] = e < ne ? outl[e] : 0;
    if (lane == 0) A.nGood[b] = ne - nBad;
}

static size_t pose_inertial_smem(int cap_e, int N) {
    return 2 * ((sizeof(liba_keyframe) + 15) & ~(size_t)15) + (size_t)(N * N + N + 216 + 216 + 9 + 9 + 225 + 225 + 15 + 15 + 2 + cap_e) * 8 + (size_t)2 * cap_e + 16;
}

extern "C" int liba_pose_inertial_kf(liba_keyframe* d_frames, const liba_keyframe* d_keyframes, const liba_rig* d_rigs, int rig_stride, const pose_edge* d_edges,
                                     const int32_t* d_n_edges, int cap_e, const liba_imu_edge* d_imu, int batch, int rec_init, uint8_t* d_outlier,
                                     double* d_H, int32_t* d_n_good, void* stream) {
    if (!d_frames || !d_keyframes || !d_rigs || !d_edges || !d_n_edges || !d_imu || !d_outlier || !d_H || !d_n_good || cap_e <= 0 || batch < 0 || rig_stride < 0)
        return ORB_E_INVALID;
    if (batch == 0) return ORB_OK;
    PoseInertialArgs A{d_frames, const_cast<liba_keyframe*>(d_keyframes), d_rigs, rig_stride, d_edges, d_n_edges, cap_e, d_imu, nullptr, rec_init, d_outlier, d_H, d_n_good};
    const size_t smem = pose_inertial_smem(cap_e, 15);
    if (smem > 160 * 1024) return ORB_E_INVALID;
    if (orb_lds_optin((const void*)k_pose_inertial<false>, smem) != ORB_OK) return ORB_E_HIP;
    hipLaunchKernelGGL(k_pose_inertial<false>, dim3(batch), dim3(64), smem, (hipStream_t)stream, A);
    return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP;
}

extern "C" int liba_pose_inertial_lastframe(liba_keyframe* d_frames, liba_keyframe* d_prev_frames, const liba_rig* d_rigs, int rig_stride, const pose_edge* d_edges,
                                            const int32_t* d_n_edges, int cap_e, const liba_imu_edge* d_imu, const liba_prior* d_priors, int batch, int rec_init,
                                            uint8_t* d_outlier, double* d_H, int32_t* d_n_good, void* stream) {
    if (!d_frames || !d_prev_frames || !d_rigs || !d_edges || !d_n_edges || !d_imu || !d_priors || !d_outlier || !d_H || !d_n_good || cap_e <= 0 || batch < 0 ||
        rig_stride < 0) return ORB_E_INVALID;
    if (batch == 0) return ORB_OK;
    PoseInertialArgs A{d_frames, d_prev_frames, d_rigs, rig_stride, d_edges, d_n_edges, cap_e, d_imu, d_priors, rec_init, d_outlier, d_H, d_n_good};
    const size_t smem = pose_inertial_smem(cap_e, 30);
    if (smem > 160 * 1024) return ORB_E_INVALID;
    if (orb_lds_optin((const void*)k_pose_inertial<true>, smem) != ORB_OK) return ORB_E_HIP;
    hipLaunchKernelGGL(k_pose_inertial<true>, dim3(batch), dim3(64), smem, (hipStream_t)stream, A);
    return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP;
}
