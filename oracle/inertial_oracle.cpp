// inertial_oracle.cpp — CPU restatement of Optimizer::LocalInertialBA's optimisation problem (reference src/Optimizer.cc:4753-5365):
// the g2o graph of VertexPose / VertexVelocity / VertexGyroBias / VertexAccBias / VertexSBAPointXYZ with EdgeInertial, EdgeGyroRW,
// EdgeAccRW, EdgeMono, EdgeStereo (src/G2oTypes.cc, include/G2oTypes.h), solved by g2o's Levenberg-Marquardt with a user lambda
// (Optimizer.cc:4884-4896) exactly like oracle/lba_oracle.cpp restates it for the visual local BA.
//
// TEST INFRASTRUCTURE ONLY (oracle/README.md): tests/ and bench.py's cpu_baseline leg are the only users.
//
// PARITY UNPINNED for the float32 cv::Mat arithmetic of IMU::Preintegrated::GetDeltaRotation/Velocity/Position (src/ImuTypes.cc:373-394)
// and for IMU::NormalizeRotation (cv::SVDecomp, ImuTypes.cc:31-37), which also sits inside g2o-side ExpSO3 (G2oTypes.cc:1000-1018):
// OpenCV is absent from this image.  Rule R3 (shared with the product): a float cv::Mat product is the double-accumulated sum rounded
// once to float; NormalizeRotation(M) is the orthogonal polar factor of M (= U*Vt of any exact SVD) computed in double by Newton's
// iteration and rounded to float.  A real OpenCV build differs from this by float rounding noise (<~2e-7 per rotation entry).
// Everything else (Eigen double arithmetic of G2oTypes.cc) is restated expression by expression and checked against central differences.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Kf {   // == liba_keyframe: ImuCamPose (G2oTypes.h:41-75) + the estimates of the V / G / A vertices of one key frame
    double Rwb[9], twb[3];          // row-major
    double Rcw[2][9], tcw[2][3];    // per camera, as the ImuCamPose constructor sets them (G2oTypes.cc:43-66)
    double v[3], bg[3], ba[3];
    int32_t pose_fixed, has_imu, imu_fixed, reserved;
};
struct Rig {  // == liba_rig: calibration members of ImuCamPose
    int32_t n_cams, reserved;
    double Rcb[2][9], tcb[2][3], Rbc[2][9], tbc[2][3];
    double bf;
    int32_t model[2];
    double p[2][8];
};
struct VisEdge { int32_t kf, point; int16_t kind, cam; float obs[3]; float inv_sigma2; };   // == lba_edge (kind 0 = EdgeMono, 1 = EdgeStereo)
struct ImuEdge {  // == liba_imu_edge: EdgeInertial + EdgeGyroRW + EdgeAccRW of one preintegration
    int32_t kf1, kf2;
    float dR[9], dV[3], dP[3], JRg[9], JVg[9], JVa[9], JPg[9], JPa[9], b[6], dT, pad;
    double huber;
    double info[81], info_g[9], info_a[9];
};
static_assert(sizeof(Kf) == 376 && sizeof(VisEdge) == 28 && sizeof(ImuEdge) == 1080, "layout");

const double GRAVITY_VALUE = 9.81;   // ImuTypes.h:40

// ---- small dense helpers (row-major 3x3) ----
void mul33(const double* A, const double* B, double* C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j]; }
void mulT33(const double* A, const double* B, double* C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j]; }   // A^T B
void mul31(const double* A, const double* x, double* y) { for (int i = 0; i < 3; i++) y[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2]; }
void mulT31(const double* A, const double* x, double* y) { for (int i = 0; i < 3; i++) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2]; }
void skew(const double* w, double* W) { W[0] = 0; W[1] = -w[2]; W[2] = w[1]; W[3] = w[2]; W[4] = 0; W[5] = -w[0]; W[6] = -w[1]; W[7] = w[0]; W[8] = 0; }

// Rule R3: orthogonal polar factor by Newton's iteration X <- (X + X^-T)/2, rounded to float (IMU::NormalizeRotation, ImuTypes.cc:31-37)
void normalizeRotationF(const double* Min, double* out) {
    double X[9];
    for (int i = 0; i < 9; i++) X[i] = (double)(float)Min[i];
    for (int it = 0; it < 8; it++) {
        const double a = X[0], b = X[1], c = X[2], d = X[3], e = X[4], f = X[5], g = X[6], h = X[7], i = X[8];
        const double C0 = e * i - f * h, C1 = -(d * i - f * g), C2 = d * h - e * g;
        const double det = a * C0 + b * C1 + c * C2, id = 1.0 / det;
        // inverse-transpose = cofactor matrix / det
        const double T[9] = {C0 * id, C1 * id, C2 * id, -(b * i - c * h) * id, (a * i - c * g) * id, -(a * h - b * g) * id,
                             (b * f - c * e) * id, -(a * f - c * d) * id, (a * e - b * d) * id};
        for (int k = 0; k < 9; k++) X[k] = 0.5 * (X[k] + T[k]);
    }
    for (int i = 0; i < 9; i++) out[i] = (double)(float)X[i];
}

// g2o-side ExpSO3 (G2oTypes.cc:995-1018): double Rodrigues, then through IMU::NormalizeRotation on a float cv::Mat
void expSO3(const double* w, double* R) {
    const double x = w[0], y = w[1], z = w[2];
    const double d2 = x * x + y * y + z * z, d = std::sqrt(d2);
    double W[9], WW[9], res[9];
    skew(w, W);
    mul33(W, W, WW);
    if (d < 1e-5) for (int i = 0; i < 9; i++) res[i] = (i % 4 == 0 ? 1.0 : 0.0) + W[i] + 0.5 * WW[i];
    else for (int i = 0; i < 9; i++) res[i] = (i % 4 == 0 ? 1.0 : 0.0) + W[i] * std::sin(d) / d + WW[i] * (1.0 - std::cos(d)) / d2;
    normalizeRotationF(res, R);
}
void logSO3(const double* R, double* w) {  // G2oTypes.cc:1020-1036
    const double tr = R[0] + R[4] + R[8];
    w[0] = (R[7] - R[5]) / 2; w[1] = (R[2] - R[6]) / 2; w[2] = (R[3] - R[1]) / 2;
    const double costheta = (tr - 1.0) * 0.5f;
    if (costheta > 1 || costheta < -1) return;
    const double theta = std::acos(costheta), s = std::sin(theta);
    if (std::fabs(s) < 1e-5) return;
    for (int i = 0; i < 3; i++) w[i] = theta * w[i] / s;
}
void invRightJacobianSO3(const double* v, double* J) {  // G2oTypes.cc:1043-1055
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = std::sqrt(d2);
    double W[9], WW[9];
    skew(v, W); mul33(W, W, WW);
    if (d < 1e-5) { for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0); return; }
    const double k = 1.0 / d2 - (1.0 + std::cos(d)) / (2.0 * d * std::sin(d));
    for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0 ? 1.0 : 0.0) + W[i] / 2 + WW[i] * k;
}
void rightJacobianSO3(const double* v, double* J) {  // G2oTypes.cc:1062-1078
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = std::sqrt(d2);
    double W[9], WW[9];
    skew(v, W); mul33(W, W, WW);
    if (d < 1e-5) { for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0); return; }
    for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0 ? 1.0 : 0.0) - W[i] * (1.0 - std::cos(d)) / d2 + WW[i] * (d - std::sin(d)) / (d2 * d);
}

// ---- IMU::Preintegrated getters (float cv::Mat arithmetic under rule R3), ImuTypes.cc:367-394 ----
struct DeltaBias { float g[3], a[3]; };
DeltaBias deltaBias(const ImuEdge& E, const double* bg, const double* ba) {  // IMU::Bias holds floats (ImuTypes.h:70-84)
    DeltaBias d;
    for (int k = 0; k < 3; k++) { d.a[k] = (float)ba[k] - E.b[k]; d.g[k] = (float)bg[k] - E.b[3 + k]; }
    return d;
}
void fmatvec(const float* M, const float* x, float* y) { for (int i = 0; i < 3; i++) y[i] = (float)((double)M[i * 3] * x[0] + (double)M[i * 3 + 1] * x[1] + (double)M[i * 3 + 2] * x[2]); }
void expSO3f(const float* v, double* R) {  // float ExpSO3 (ImuTypes.cc:49-61), eps = 1e-4
    const float x = v[0], y = v[1], z = v[2];
    const float d2 = x * x + y * y + z * z;
    const float d = std::sqrt(d2);
    const double W[9] = {0, -(double)z, (double)y, (double)z, 0, -(double)x, -(double)y, (double)x, 0};
    double WW[9];
    mul33(W, W, WW);
    for (int i = 0; i < 9; i++) WW[i] = (double)(float)WW[i];
    if (d < 1e-4f) for (int i = 0; i < 9; i++) R[i] = (double)(float)((i % 4 == 0 ? 1.0 : 0.0) + W[i] + 0.5 * WW[i]);
    else {
        const double a = std::sin((double)d) / (double)d, c = (1.0 - std::cos((double)d)) / (double)d2;
        for (int i = 0; i < 9; i++) R[i] = (double)(float)((i % 4 == 0 ? 1.0 : 0.0) + W[i] * a + WW[i] * c);
    }
}
void getDeltaRotation(const ImuEdge& E, const DeltaBias& db, double* dR) {
    float v[3];
    fmatvec(E.JRg, db.g, v);
    double Ex[9], R0[9], M[9];
    expSO3f(v, Ex);
    for (int i = 0; i < 9; i++) R0[i] = (double)E.dR[i];
    mul33(R0, Ex, M);
    normalizeRotationF(M, dR);
}
void getDeltaVP(const float* d0, const float* Jg, const float* Ja, const DeltaBias& db, double* out) {
    float t1[3], t2[3];
    fmatvec(Jg, db.g, t1); fmatvec(Ja, db.a, t2);
    for (int i = 0; i < 3; i++) out[i] = (double)((d0[i] + t1[i]) + t2[i]);
}

// ---- EdgeInertial (G2oTypes.cc:706-800).  J: 9 x 24 row-major, columns [VP1 6 | VV1 3 | VG1 3 | VA1 3 | VP2 6 | VV2 3] ----
void inertialError(const ImuEdge& E, const Kf& k1, const Kf& k2, double* e) {
    const DeltaBias db = deltaBias(E, k1.bg, k1.ba);
    double dR[9], dV[3], dP[3];
    getDeltaRotation(E, db, dR);
    getDeltaVP(E.dV, E.JVg, E.JVa, db, dV);
    getDeltaVP(E.dP, E.JPg, E.JPa, db, dP);
    const double dt = (double)E.dT, g[3] = {0, 0, -GRAVITY_VALUE};
    double A[9], eR[9];
    // dR^T * Rwb1^T * Rwb2
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i * 3 + j] = dR[i] * k1.Rwb[j * 3] + dR[3 + i] * k1.Rwb[j * 3 + 1] + dR[6 + i] * k1.Rwb[j * 3 + 2];
    mul33(A, k2.Rwb, eR);
    logSO3(eR, e);
    double t[3], r[3];
    for (int i = 0; i < 3; i++) t[i] = k2.v[i] - k1.v[i] - g[i] * dt;
    mulT31(k1.Rwb, t, r);
    for (int i = 0; i < 3; i++) e[3 + i] = r[i] - dV[i];
    for (int i = 0; i < 3; i++) t[i] = k2.twb[i] - k1.twb[i] - k1.v[i] * dt - g[i] * dt * dt / 2;
    mulT31(k1.Rwb, t, r);
    for (int i = 0; i < 3; i++) e[6 + i] = r[i] - dP[i];
}
void inertialJacobian(const ImuEdge& E, const Kf& k1, const Kf& k2, double* J) {
    memset(J, 0, 9 * 24 * sizeof(double));
    const DeltaBias db = deltaBias(E, k1.bg, k1.ba);
    const double dbg[3] = {(double)db.g[0], (double)db.g[1], (double)db.g[2]};
    double dR[9];
    getDeltaRotation(E, db, dR);
    const double dt = (double)E.dT, g[3] = {0, 0, -GRAVITY_VALUE};
    double A[9], eR[9], er[3], invJr[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i * 3 + j] = dR[i] * k1.Rwb[j * 3] + dR[3 + i] * k1.Rwb[j * 3 + 1] + dR[6 + i] * k1.Rwb[j * 3 + 2];
    mul33(A, k2.Rwb, eR);
    logSO3(eR, er);
    invRightJacobianSO3(er, invJr);
    auto put = [&](int r0, int c0, const double* M, double s) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) J[(r0 + i) * 24 + c0 + j] = s * M[i * 3 + j]; };
    double T1[9], T2[9], t[3], r[3], S[9];
    // VP1
    {   // -invJr*Rwb2^T*Rwb1, evaluated left to right: (-invJr * Rwb2^T) * Rwb1
        double L[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L[i * 3 + j] = -(invJr[i * 3] * k2.Rwb[j * 3] + invJr[i * 3 + 1] * k2.Rwb[j * 3 + 1] + invJr[i * 3 + 2] * k2.Rwb[j * 3 + 2]);
        mul33(L, k1.Rwb, T2);
        put(0, 0, T2, 1.0);
    }
    for (int i = 0; i < 3; i++) t[i] = k2.v[i] - k1.v[i] - g[i] * dt;
    mulT31(k1.Rwb, t, r); skew(r, S); put(3, 0, S, 1.0);
    for (int i = 0; i < 3; i++) t[i] = k2.twb[i] - k1.twb[i] - k1.v[i] * dt - 0.5 * g[i] * dt * dt;
    mulT31(k1.Rwb, t, r); skew(r, S); put(6, 0, S, 1.0);
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    put(6, 3, I3, -1.0);
    // VV1
    double Rbw1[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rbw1[i * 3 + j] = k1.Rwb[j * 3 + i];
    put(3, 6, Rbw1, -1.0);
    for (int i = 0; i < 9; i++) T1[i] = -Rbw1[i] * dt;
    put(6, 6, T1, 1.0);
    // VG1: -invJr * eR^T * RightJacobianSO3(JRg*dbg) * JRg
    double JRg[9], JVg[9], JVa[9], JPg[9], JPa[9], v[3], Jr[9];
    for (int i = 0; i < 9; i++) { JRg[i] = E.JRg[i]; JVg[i] = E.JVg[i]; JVa[i] = E.JVa[i]; JPg[i] = E.JPg[i]; JPa[i] = E.JPa[i]; }
    mul31(JRg, dbg, v);
    rightJacobianSO3(v, Jr);
    {
        double L[9], L2[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L[i * 3 + j] = -(invJr[i * 3] * eR[j * 3] + invJr[i * 3 + 1] * eR[j * 3 + 1] + invJr[i * 3 + 2] * eR[j * 3 + 2]);
        mul33(L, Jr, L2); mul33(L2, JRg, T2);
        put(0, 9, T2, 1.0);
    }
    put(3, 9, JVg, -1.0); put(6, 9, JPg, -1.0);
    // VA1
    put(3, 12, JVa, -1.0); put(6, 12, JPa, -1.0);
    // VP2
    put(0, 15, invJr, 1.0);
    mul33(Rbw1, k2.Rwb, T1); put(6, 18, T1, 1.0);
    // VV2
    put(3, 21, Rbw1, 1.0);
}

// ---- cameras (GeometricCamera::project / projectJac on Eigen::Vector3d) ----
void camProject(int model, const double* p, const double* v, double* res) {
    if (model == 0) { res[0] = p[0] * v[0] / v[2] + p[2]; res[1] = p[1] * v[1] / v[2] + p[3]; }   // Pinhole.cpp:43-49
    else {  // KannalaBrandt8.cpp:52-66
        const double x2_plus_y2 = v[0] * v[0] + v[1] * v[1];
        const double theta = atan2f(sqrtf((float)x2_plus_y2), (float)v[2]);
        const double psi = atan2f((float)v[1], (float)v[0]);
        const double theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
        const double r = theta + p[4] * theta3 + p[5] * theta5 + p[6] * theta7 + p[7] * theta9;
        res[0] = p[0] * r * std::cos(psi) + p[2];
        res[1] = p[1] * r * std::sin(psi) + p[3];
    }
}
void camProjectJac(int model, const double* p, const double* v, double* J) {  // 2x3 row-major
    if (model == 0) {  // Pinhole.cpp:89-100
        J[0] = p[0] / v[2]; J[1] = 0; J[2] = -p[0] * v[0] / (v[2] * v[2]);
        J[3] = 0; J[4] = p[1] / v[2]; J[5] = -p[1] * v[1] / (v[2] * v[2]);
    } else {  // KannalaBrandt8.cpp:166-196
        const double x2 = v[0] * v[0], y2 = v[1] * v[1], z2 = v[2] * v[2];
        const double r2 = x2 + y2, r = std::sqrt(r2), r3 = r2 * r;
        const double theta = std::atan2(r, v[2]);
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

// ---- EdgeMono / EdgeStereo (G2oTypes.h:337-437, G2oTypes.cc:352-418): A = d e / d point (D x 3), B = d e / d pose (D x 6) ----
struct VLin { int D; double e[3], A[9], B[18], chi2, rho0, rho1; bool depthPositive; };
void visLinearize(const VisEdge& E, const Kf& kf, const Rig& rig, const double* X, double huberMono, double huberStereo, bool jac, VLin& L) {
    memset(&L, 0, sizeof(L));
    const int c = E.cam;
    const double* Rcw = kf.Rcw[c]; const double* tcw = kf.tcw[c];
    double Xc[3], proj[2];
    mul31(Rcw, X, Xc);
    for (int i = 0; i < 3; i++) Xc[i] += tcw[i];
    camProject(rig.model[c], rig.p[c], Xc, proj);
    L.D = E.kind == 1 ? 3 : 2;
    L.e[0] = (double)E.obs[0] - proj[0];
    L.e[1] = (double)E.obs[1] - proj[1];
    if (E.kind == 1) { const double invZ = 1 / Xc[2]; L.e[2] = (double)E.obs[2] - (proj[0] - rig.bf * invZ); }   // ImuCamPose::ProjectStereo
    L.depthPositive = (Rcw[6] * X[0] + Rcw[7] * X[1] + Rcw[8] * X[2] + tcw[2]) > 0.0;
    if (jac) {
        double Xb[3], pj[9] = {0};
        mul31(rig.Rbc[c], Xc, Xb);
        for (int i = 0; i < 3; i++) Xb[i] += rig.tbc[c][i];
        camProjectJac(rig.model[c], rig.p[c], Xc, pj);
        if (E.kind == 1) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2]; pj[8] += rig.bf * (1.0 / (Xc[2] * Xc[2])); }
        const double S[18] = {0, Xb[2], -Xb[1], 1, 0, 0, -Xb[2], 0, Xb[0], 0, 1, 0, Xb[1], -Xb[0], 0, 0, 0, 1};
        for (int r = 0; r < L.D; r++) {
            for (int k = 0; k < 3; k++) L.A[r * 3 + k] = -pj[r * 3] * Rcw[k] - pj[r * 3 + 1] * Rcw[3 + k] - pj[r * 3 + 2] * Rcw[6 + k];
            double M[3];   // (proj_jac * Rcb) row
            for (int k = 0; k < 3; k++) M[k] = pj[r * 3] * rig.Rcb[c][k] + pj[r * 3 + 1] * rig.Rcb[c][3 + k] + pj[r * 3 + 2] * rig.Rcb[c][6 + k];
            for (int k = 0; k < 6; k++) L.B[r * 6 + k] = M[0] * S[k] + M[1] * S[6 + k] + M[2] * S[12 + k];
        }
    }
    const double s = (double)E.inv_sigma2;
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

// ImuCamPose::Update (G2oTypes.cc:196-221).  The `NormalizeRotation(Rwb)` of :206 discards its result: no effect.
void poseUpdate(Kf& k, const Rig& rig, const double* pu) {
    double d[3], E[9], Rn[9];
    mul31(k.Rwb, pu + 3, d);
    for (int i = 0; i < 3; i++) k.twb[i] += d[i];
    expSO3(pu, E);
    mul33(k.Rwb, E, Rn);
    memcpy(k.Rwb, Rn, sizeof(Rn));
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

bool inv3(const double* D, double* out) {  // column-major symmetric 3x3 cofactor inverse
    const double a = D[0], b = D[3], c = D[6], d = D[1], e = D[4], f = D[7], g = D[2], h = D[5], i = D[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C, id = 1.0 / det;
    out[0] = A * id; out[3] = -(b * i - c * h) * id; out[6] = (b * f - c * e) * id;
    out[1] = B * id; out[4] = (a * i - c * g) * id; out[7] = -(a * f - c * d) * id;
    out[2] = C * id; out[5] = -(a * h - b * g) * id; out[8] = (a * e - b * d) * id;
    return std::isfinite(det) && det != 0.0;
}
bool cholSolve(std::vector<double>& S, int n, std::vector<double>& rhs) {  // dense Cholesky (lower), column-major
    for (int k = 0; k < n; k++) {
        double dkk = S[(size_t)k * n + k];
        if (!(dkk > 0) || !std::isfinite(dkk)) return false;
        dkk = std::sqrt(dkk);
        S[(size_t)k * n + k] = dkk;
        for (int i = k + 1; i < n; i++) S[(size_t)k * n + i] /= dkk;
        for (int j = k + 1; j < n; j++) { const double ljk = S[(size_t)k * n + j]; for (int i = j; i < n; i++) S[(size_t)j * n + i] -= S[(size_t)k * n + i] * ljk; }
    }
    for (int i = 0; i < n; i++) { double v = rhs[i]; for (int k = 0; k < i; k++) v -= S[(size_t)k * n + i] * rhs[k]; rhs[i] = v / S[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double v = rhs[i]; for (int k = i + 1; k < n; k++) v -= S[(size_t)i * n + k] * rhs[k]; rhs[i] = v / S[(size_t)i * n + i]; }
    return true;
}

struct Problem {
    Kf* kfs; int nKf; const Rig* rig; double* points; int nPoints; const VisEdge* vis; int nVis; const ImuEdge* imu; int nImu;
    double huberMono, huberStereo;
    std::vector<int> hp, hi;   // reduced-system offsets of the pose block / VGA block of each key frame (-1 = fixed or absent)
    int np6 = 0, DR = 0;
    void index() {
        hp.assign(nKf, -1); hi.assign(nKf, -1);
        int o = 0;
        for (int k = 0; k < nKf; k++) if (!kfs[k].pose_fixed) { hp[k] = o; o += 6; }   // poses first (vertex ids = KF ids), then V/G/A (ids maxKFid+3*id+1..3)
        np6 = o;
        for (int k = 0; k < nKf; k++) if (kfs[k].has_imu && !kfs[k].imu_fixed) { hi[k] = o; o += 9; }
        DR = o;
    }
    // chi2 / robust chi2 of the inertial + random-walk edges of one preintegration
    void imuChi(const ImuEdge& E, double* chiI, double* rho0, double* rho1, double* chiG, double* chiA, double* e9) const {
        double e[9];
        inertialError(E, kfs[E.kf1], kfs[E.kf2], e);
        double c = 0;
        for (int r = 0; r < 9; r++) { double s = 0; for (int q = 0; q < 9; q++) s += E.info[r * 9 + q] * e[q]; c += e[r] * s; }
        *chiI = c;
        if (E.huber > 0) {
            const double dsqr = E.huber * E.huber;
            if (c <= dsqr) { *rho0 = c; *rho1 = 1; } else { const double sq = std::sqrt(c); *rho0 = 2 * sq * E.huber - dsqr; *rho1 = E.huber / sq; }
        } else { *rho0 = c; *rho1 = 1; }
        double eg[3], ea[3];
        for (int i = 0; i < 3; i++) { eg[i] = kfs[E.kf2].bg[i] - kfs[E.kf1].bg[i]; ea[i] = kfs[E.kf2].ba[i] - kfs[E.kf1].ba[i]; }
        double cg = 0, ca = 0;
        for (int r = 0; r < 3; r++) { double s = 0, t = 0; for (int q = 0; q < 3; q++) { s += E.info_g[r * 3 + q] * eg[q]; t += E.info_a[r * 3 + q] * ea[q]; } cg += eg[r] * s; ca += ea[r] * t; }
        *chiG = cg; *chiA = ca;
        if (e9) memcpy(e9, e, sizeof(e));
    }
    bool imuActive(const ImuEdge& E) const {  // an edge is active iff not all its vertices are fixed (sparse_optimizer.cpp:232-235)
        return hp[E.kf1] >= 0 || hi[E.kf1] >= 0 || hp[E.kf2] >= 0 || hi[E.kf2] >= 0;
    }
    double robustChi() const {   // activeRobustChi2 (sparse_optimizer.cpp:100-114)
        double r = 0;
        for (int i = 0; i < nImu; i++) {
            if (!imuActive(imu[i])) continue;
            double c, r0, r1, cg, ca;
            imuChi(imu[i], &c, &r0, &r1, &cg, &ca, nullptr);
            r += r0;
            if (hi[imu[i].kf1] >= 0 || hi[imu[i].kf2] >= 0) r += cg + ca;
        }
        VLin L;
        for (int e = 0; e < nVis; e++) { visLinearize(vis[e], kfs[vis[e].kf], *rig, points + 3 * vis[e].point, huberMono, huberStereo, false, L); r += L.rho0; }
        return r;
    }
};
}  // namespace

extern "C" {
// error of one EdgeInertial after perturbing its six vertices by `d` (24 values, column layout of the Jacobian) through their oplus
void oib_imu_error(const void* edge_, const void* kf1_, const void* kf2_, const void* rig_, const double* d, double* e9) {
    Kf k1 = *(const Kf*)kf1_, k2 = *(const Kf*)kf2_;
    const Rig& rig = *(const Rig*)rig_;
    if (d) {
        poseUpdate(k1, rig, d);
        for (int i = 0; i < 3; i++) { k1.v[i] += d[6 + i]; k1.bg[i] += d[9 + i]; k1.ba[i] += d[12 + i]; k2.v[i] += d[21 + i]; }
        poseUpdate(k2, rig, d + 15);
    }
    inertialError(*(const ImuEdge*)edge_, k1, k2, e9);
}
void oib_imu_jacobian(const void* edge_, const void* kf1_, const void* kf2_, double* J) { inertialJacobian(*(const ImuEdge*)edge_, *(const Kf*)kf1_, *(const Kf*)kf2_, J); }
// visual edge: error after perturbing pose (6) and point (3); Jacobians B (D x 6) and A (D x 3)
void oib_vis_error(const void* edge_, const void* kf_, const void* rig_, const double* X, const double* dpose, const double* dpoint, double* e3, double* A, double* B) {
    Kf k = *(const Kf*)kf_;
    const Rig& rig = *(const Rig*)rig_;
    double x[3] = {X[0], X[1], X[2]};
    if (dpose) poseUpdate(k, rig, dpose);
    if (dpoint) for (int i = 0; i < 3; i++) x[i] += dpoint[i];
    VLin L;
    visLinearize(*(const VisEdge*)edge_, k, rig, x, 0, 0, A != nullptr, L);
    memcpy(e3, L.e, sizeof(L.e));
    if (A) { memcpy(A, L.A, sizeof(L.A)); memcpy(B, L.B, sizeof(L.B)); }
}
void oib_pose_update(void* kf_, const void* rig_, const double* pu) { poseUpdate(*(Kf*)kf_, *(const Rig*)rig_, pu); }

// per-edge chi2 (what the outlier pass of Optimizer.cc:5237-5275 reads) + the robust total
void oib_errors(const void* kfs_, int n_kf, const void* rig_, const double* points, int n_points, const void* vis_, int n_vis, const void* imu_, int n_imu,
                double huberMono, double huberStereo, double* vis_chi2, uint8_t* vis_depth_pos, double* imu_chi2, double* robust_sum) {
    Problem P{(Kf*)kfs_, n_kf, (const Rig*)rig_, (double*)points, n_points, (const VisEdge*)vis_, n_vis, (const ImuEdge*)imu_, n_imu, huberMono, huberStereo};
    P.index();
    VLin L;
    for (int e = 0; e < n_vis; e++) {
        visLinearize(P.vis[e], P.kfs[P.vis[e].kf], *P.rig, points + 3 * P.vis[e].point, huberMono, huberStereo, false, L);
        if (vis_chi2) vis_chi2[e] = L.chi2;
        if (vis_depth_pos) vis_depth_pos[e] = L.depthPositive;
    }
    for (int i = 0; i < n_imu && imu_chi2; i++) { double r0, r1; P.imuChi(P.imu[i], &imu_chi2[3 * i], &r0, &r1, &imu_chi2[3 * i + 1], &imu_chi2[3 * i + 2], nullptr); }
    if (robust_sum) *robust_sum = P.robustChi();
}

// optimizer.optimize(iterations) with OptimizationAlgorithmLevenberg + setUserLambdaInit(lambda_init) (Optimizer.cc:4884-4896, :5225).
// kfs / points updated in place.  stats: [0] iterations, [1] final robust chi2, [2] final lambda, [3] LM trials, [4] initial robust chi2
void oib_optimize(void* kfs_, int n_kf, const void* rig_, double* points, int n_points, const void* vis_, int n_vis, const void* imu_, int n_imu,
                  double huberMono, double huberStereo, double lambda_init, int iterations, double* stats) {
    Problem P{(Kf*)kfs_, n_kf, (const Rig*)rig_, points, n_points, (const VisEdge*)vis_, n_vis, (const ImuEdge*)imu_, n_imu, huberMono, huberStereo};
    P.index();
    const int DR = P.DR;
    std::vector<int> lmStart(n_points + 1, 0);
    for (int e = 0; e < n_vis; e++) lmStart[P.vis[e].point + 1]++;
    for (int l = 0; l < n_points; l++) lmStart[l + 1] += lmStart[l];
    // edges must be landmark-major (like lba_edge); checked here so a wrong caller fails loudly
    for (int e = 1; e < n_vis; e++) if (P.vis[e].point < P.vis[e - 1].point) { if (stats) stats[0] = -1; return; }
    std::vector<double> H((size_t)DR * DR), b(DR), Hll((size_t)n_points * 9), bl((size_t)n_points * 3), Hpl((size_t)n_vis * 18);
    double lambda = lambda_init, ni = 2;
    int nBad = 0, it = 0, trialsTotal = 0;
    double currentChi = 0;
    const double chi0 = P.robustChi();
    for (it = 0; it < iterations; it++) {
        currentChi = P.robustChi();
        double tempChi = currentChi;
        const double iniChi = currentChi;
        // ---- buildSystem: H (dense over the non-marginalised vertices), b, Hll, bl, Hpl
        std::fill(H.begin(), H.end(), 0.0); std::fill(b.begin(), b.end(), 0.0); std::fill(Hll.begin(), Hll.end(), 0.0); std::fill(bl.begin(), bl.end(), 0.0);
        for (int i = 0; i < n_imu; i++) {
            const ImuEdge& E = P.imu[i];
            if (!P.imuActive(E)) continue;
            double e[9], J[9 * 24], chi, r0, r1, cg, ca;
            P.imuChi(E, &chi, &r0, &r1, &cg, &ca, e);
            inertialJacobian(E, P.kfs[E.kf1], P.kfs[E.kf2], J);
            const int off[6] = {P.hp[E.kf1], P.hi[E.kf1], P.hi[E.kf1] < 0 ? -1 : P.hi[E.kf1] + 3, P.hi[E.kf1] < 0 ? -1 : P.hi[E.kf1] + 6, P.hp[E.kf2], P.hi[E.kf2]};
            const int col0[6] = {0, 6, 9, 12, 15, 21}, dim[6] = {6, 3, 3, 3, 6, 3};
            double OJ[9 * 24], Oe[9];   // rho1 * Omega * J, -rho1 * Omega * e
            for (int r = 0; r < 9; r++) {
                for (int c = 0; c < 24; c++) { double s = 0; for (int q = 0; q < 9; q++) s += E.info[r * 9 + q] * J[q * 24 + c]; OJ[r * 24 + c] = r1 * s; }
                double s = 0; for (int q = 0; q < 9; q++) s += E.info[r * 9 + q] * e[q]; Oe[r] = -r1 * s;
            }
            for (int a = 0; a < 6; a++) {
                if (off[a] < 0) continue;
                for (int ca2 = 0; ca2 < dim[a]; ca2++) {
                    double s = 0; for (int r = 0; r < 9; r++) s += J[r * 24 + col0[a] + ca2] * Oe[r];
                    b[off[a] + ca2] += s;
                    for (int bb = 0; bb < 6; bb++) {
                        if (off[bb] < 0) continue;
                        for (int cb = 0; cb < dim[bb]; cb++) { double t = 0; for (int r = 0; r < 9; r++) t += J[r * 24 + col0[a] + ca2] * OJ[r * 24 + col0[bb] + cb]; H[(size_t)(off[bb] + cb) * DR + off[a] + ca2] += t; }
                    }
                }
            }
            // EdgeGyroRW / EdgeAccRW (G2oTypes.h:633-700): e = b2 - b1, J = [-I, I], no robust kernel
            for (int w = 0; w < 2; w++) {
                const double* Om = w == 0 ? E.info_g : E.info_a;
                const int o1 = P.hi[E.kf1] < 0 ? -1 : P.hi[E.kf1] + 3 + 3 * w, o2 = P.hi[E.kf2] < 0 ? -1 : P.hi[E.kf2] + 3 + 3 * w;
                if (o1 < 0 && o2 < 0) continue;
                double er[3];
                for (int k = 0; k < 3; k++) er[k] = w == 0 ? P.kfs[E.kf2].bg[k] - P.kfs[E.kf1].bg[k] : P.kfs[E.kf2].ba[k] - P.kfs[E.kf1].ba[k];
                for (int r = 0; r < 3; r++) {
                    double oe = 0; for (int q = 0; q < 3; q++) oe += Om[r * 3 + q] * er[q];
                    if (o1 >= 0) b[o1 + r] += oe;        // J1^T (-Omega e) = +Omega e
                    if (o2 >= 0) b[o2 + r] += -oe;
                    for (int c = 0; c < 3; c++) {
                        if (o1 >= 0) H[(size_t)(o1 + c) * DR + o1 + r] += Om[r * 3 + c];
                        if (o2 >= 0) H[(size_t)(o2 + c) * DR + o2 + r] += Om[r * 3 + c];
                        if (o1 >= 0 && o2 >= 0) { H[(size_t)(o2 + c) * DR + o1 + r] += -Om[r * 3 + c]; H[(size_t)(o1 + c) * DR + o2 + r] += -Om[c * 3 + r]; }
                    }
                }
            }
        }
        VLin L;
        for (int e = 0; e < n_vis; e++) {
            const VisEdge& E = P.vis[e];
            visLinearize(E, P.kfs[E.kf], *P.rig, points + 3 * E.point, huberMono, huberStereo, true, L);
            const double w = L.rho1 * (double)E.inv_sigma2;
            const int hp = P.hp[E.kf], l = E.point;
            for (int r = 0; r < 3; r++) {
                double s = 0; for (int d = 0; d < L.D; d++) s += L.A[d * 3 + r] * L.e[d];
                bl[(size_t)l * 3 + r] += -w * s;
                for (int c = 0; c < 3; c++) { double t = 0; for (int d = 0; d < L.D; d++) t += L.A[d * 3 + r] * L.A[d * 3 + c]; Hll[(size_t)l * 9 + c * 3 + r] += w * t; }
            }
            if (hp >= 0) {
                for (int r = 0; r < 6; r++) {
                    double s = 0; for (int d = 0; d < L.D; d++) s += L.B[d * 6 + r] * L.e[d];
                    b[hp + r] += -w * s;
                    for (int c = 0; c < 6; c++) { double t = 0; for (int d = 0; d < L.D; d++) t += L.B[d * 6 + r] * L.B[d * 6 + c]; H[(size_t)(hp + c) * DR + hp + r] += w * t; }
                    for (int c = 0; c < 3; c++) { double t = 0; for (int d = 0; d < L.D; d++) t += L.B[d * 6 + r] * L.A[d * 3 + c]; Hpl[(size_t)e * 18 + c * 6 + r] = w * t; }
                }
            }
        }
        double rhoLM = 0;
        int qmax = 0;
        std::vector<Kf> kfBak;
        std::vector<double> ptBak;
        do {
            kfBak.assign(P.kfs, P.kfs + n_kf);
            ptBak.assign(points, points + (size_t)n_points * 3);
            std::vector<double> S = H, coeff(DR, 0.0), Dinv((size_t)n_points * 9), db((size_t)n_points * 3);
            for (int i = 0; i < DR; i++) S[(size_t)i * DR + i] += lambda;
            bool ok2 = true;
            for (int l = 0; l < n_points; l++) {
                double D[9];
                for (int k = 0; k < 9; k++) D[k] = Hll[(size_t)l * 9 + k] + ((k % 4 == 0) ? lambda : 0.0);
                if (!inv3(D, &Dinv[(size_t)l * 9])) ok2 = false;
                for (int r = 0; r < 3; r++) db[(size_t)l * 3 + r] = Dinv[(size_t)l * 9 + r] * bl[(size_t)l * 3] + Dinv[(size_t)l * 9 + 3 + r] * bl[(size_t)l * 3 + 1] + Dinv[(size_t)l * 9 + 6 + r] * bl[(size_t)l * 3 + 2];
                for (int e1 = lmStart[l]; e1 < lmStart[l + 1]; e1++) {
                    const int h1 = P.hp[P.vis[e1].kf];
                    if (h1 < 0) continue;
                    const double* Bi = &Hpl[(size_t)e1 * 18];
                    double BD[18];
                    for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) BD[c * 6 + r] = Bi[r] * Dinv[(size_t)l * 9 + c * 3] + Bi[6 + r] * Dinv[(size_t)l * 9 + c * 3 + 1] + Bi[12 + r] * Dinv[(size_t)l * 9 + c * 3 + 2];
                    for (int r = 0; r < 6; r++) coeff[h1 + r] += Bi[r] * db[(size_t)l * 3] + Bi[6 + r] * db[(size_t)l * 3 + 1] + Bi[12 + r] * db[(size_t)l * 3 + 2];
                    for (int e2 = lmStart[l]; e2 < lmStart[l + 1]; e2++) {
                        const int h2 = P.hp[P.vis[e2].kf];
                        if (h2 < 0) continue;
                        const double* Bj = &Hpl[(size_t)e2 * 18];
                        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) S[(size_t)(h2 + c) * DR + h1 + r] -= BD[r] * Bj[c] + BD[6 + r] * Bj[6 + c] + BD[12 + r] * Bj[12 + c];
                    }
                }
            }
            std::vector<double> xp(DR);
            for (int i = 0; i < DR; i++) xp[i] = b[i] - coeff[i];
            if (ok2) ok2 = cholSolve(S, DR, xp);
            std::vector<double> xl((size_t)n_points * 3, 0.0);
            if (ok2) {
                for (int l = 0; l < n_points; l++) {
                    double cl[3] = {bl[(size_t)l * 3], bl[(size_t)l * 3 + 1], bl[(size_t)l * 3 + 2]};
                    for (int e1 = lmStart[l]; e1 < lmStart[l + 1]; e1++) {
                        const int h1 = P.hp[P.vis[e1].kf];
                        if (h1 < 0) continue;
                        const double* Bi = &Hpl[(size_t)e1 * 18];
                        for (int c = 0; c < 3; c++) for (int r = 0; r < 6; r++) cl[c] -= Bi[c * 6 + r] * xp[h1 + r];
                    }
                    for (int r = 0; r < 3; r++) xl[(size_t)l * 3 + r] = Dinv[(size_t)l * 9 + r] * cl[0] + Dinv[(size_t)l * 9 + 3 + r] * cl[1] + Dinv[(size_t)l * 9 + 6 + r] * cl[2];
                }
                for (int k = 0; k < n_kf; k++) {
                    if (P.hp[k] >= 0) poseUpdate(P.kfs[k], *P.rig, &xp[P.hp[k]]);
                    if (P.hi[k] >= 0) for (int i = 0; i < 3; i++) { P.kfs[k].v[i] += xp[P.hi[k] + i]; P.kfs[k].bg[i] += xp[P.hi[k] + 3 + i]; P.kfs[k].ba[i] += xp[P.hi[k] + 6 + i]; }
                }
                for (size_t k = 0; k < (size_t)n_points * 3; k++) points[k] += xl[k];
            }
            tempChi = P.robustChi();
            if (!ok2) tempChi = 1.7976931348623157e308;
            rhoLM = currentChi - tempChi;
            double scale = 0;
            if (ok2) {
                for (int i = 0; i < DR; i++) scale += xp[i] * (lambda * xp[i] + b[i]);
                for (size_t k = 0; k < (size_t)n_points * 3; k++) scale += xl[k] * (lambda * xl[k] + bl[k]);
            }
            scale += 1e-3;
            rhoLM /= scale;
            if (rhoLM > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rhoLM - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                std::copy(kfBak.begin(), kfBak.end(), P.kfs);
                memcpy(points, ptBak.data(), ptBak.size() * 8);
            }
            qmax++; trialsTotal++;
        } while (rhoLM < 0 && qmax < 100);
        if (qmax == 100 || rhoLM == 0) { it++; break; }
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) { it++; break; }
    }
    if (stats) { stats[0] = it; stats[1] = P.robustChi(); stats[2] = lambda; stats[3] = trialsTotal; stats[4] = chi0; }
}
}  // extern "C"

// ---- Optimizer::PoseInertialOptimizationLastKeyFrame (reference src/Optimizer.cc:7665-8067) --------------------------------------------
// One frame: VertexPose / Velocity / GyroBias / AccBias free, the last key frame's four vertices fixed; EdgeMonoOnlyPose / EdgeStereoOnlyPose
// (G2oTypes.h:387-421, 463-491; G2oTypes.cc:371-389, 420-441), EdgeInertial, EdgeGyroRW, EdgeAccRW; OptimizationAlgorithmGaussNewton with
// LinearSolverDense (:7669-7672); 4 rounds x optimize(10), chi2 re-classification with the bClose rule, Huber dropped for the last round,
// the < 30 inliers recovery pass, and the 15x15 Hessian of the final state handed to the next frame (ConstraintPoseImu, :8040-8062).
namespace {
struct PEdge { float xw[3]; float obs[3]; float inv_sigma2; int16_t kind, cam; };   // == pose_edge; kind bit 8 (0x100) = bClose (mTrackDepth < 10)
struct PLin { int D; double e[3], B[18], chi2; bool depthPositive; };
void poseEdgeLinearize(const PEdge& E, const Kf& kf, const Rig& rig, bool jac, PLin& L) {
    VisEdge V; V.kf = 0; V.point = 0; V.kind = (int16_t)(E.kind & 0xFF); V.cam = E.cam; V.obs[0] = E.obs[0]; V.obs[1] = E.obs[1]; V.obs[2] = E.obs[2]; V.inv_sigma2 = E.inv_sigma2;
    const double X[3] = {(double)E.xw[0], (double)E.xw[1], (double)E.xw[2]};
    VLin T;
    visLinearize(V, kf, rig, X, 0, 0, jac, T);
    L.D = T.D; memcpy(L.e, T.e, sizeof(L.e)); memcpy(L.B, T.B, sizeof(L.B)); L.chi2 = T.chi2; L.depthPositive = T.depthPositive;
}
bool cholSolve15(double* S, int n, double* x) {   // LinearSolverDense: dense Cholesky of the (here 15x15) system; column-major lower
    for (int k = 0; k < n; k++) {
        double dkk = S[k * n + k];
        if (!(dkk > 0) || !std::isfinite(dkk)) return false;
        dkk = std::sqrt(dkk);
        S[k * n + k] = dkk;
        for (int i = k + 1; i < n; i++) S[k * n + i] /= dkk;
        for (int j = k + 1; j < n; j++) { const double ljk = S[k * n + j]; for (int i = j; i < n; i++) S[j * n + i] -= S[k * n + i] * ljk; }
    }
    for (int i = 0; i < n; i++) { double v = x[i]; for (int k = 0; k < i; k++) v -= S[k * n + i] * x[k]; x[i] = v / S[i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double v = x[i]; for (int k = i + 1; k < n; k++) v -= S[i * n + k] * x[k]; x[i] = v / S[i * n + i]; }
    return true;
}
}  // namespace

extern "C" int oib_pose_inertial_kf(void* frame_, const void* keyframe_, const void* rig_, const void* edges_, int n_edges, const void* imu_, int rec_init,
                                    uint8_t* outlier, double* H15) {
    Kf& F = *(Kf*)frame_;
    const Kf& K = *(const Kf*)keyframe_;
    const Rig& rig = *(const Rig*)rig_;
    const PEdge* edges = (const PEdge*)edges_;
    const ImuEdge& E = *(const ImuEdge*)imu_;
    const double thMono = (double)std::sqrt(5.991f), thStereo = (double)std::sqrt(7.815f);   // const float thHuberMono = sqrt(5.991)
    std::vector<double> chiLast(n_edges, 0.0);
    std::vector<char> level(n_edges, 0), depthOk(n_edges, 1);
    for (int i = 0; i < n_edges; i++) outlier[i] = 0;
    const float chi2Mono[4] = {12, 7.5, 5.991, 5.991}, chi2Stereo[4] = {15.6, 9.8, 7.815, 7.815};
    int nBad = 0, nInliers = 0;
    bool robust = true;
    PLin L;
    for (int it = 0; it < 4; it++) {
        for (int gn = 0; gn < 10; gn++) {   // optimizer.optimize(its[it]): Gauss-Newton, no step control
            double H[225] = {0}, b[15] = {0};
            for (int i = 0; i < n_edges; i++) {
                if (level[i]) continue;
                poseEdgeLinearize(edges[i], F, rig, true, L);
                chiLast[i] = L.chi2;
                double rho1 = 1.0;
                if (robust) { const double d = (edges[i].kind & 0xFF) == 1 ? thStereo : thMono, dsqr = d * d; if (L.chi2 > dsqr) rho1 = d / std::sqrt(L.chi2); }
                const double w = rho1 * (double)edges[i].inv_sigma2;
                for (int r = 0; r < 6; r++) {
                    double s = 0; for (int d = 0; d < L.D; d++) s += L.B[d * 6 + r] * L.e[d];
                    b[r] += -w * s;
                    for (int c = 0; c < 6; c++) { double t = 0; for (int d = 0; d < L.D; d++) t += L.B[d * 6 + r] * L.B[d * 6 + c]; H[c * 15 + r] += w * t; }
                }
            }
            {   // EdgeInertial (vertices 4, 5 = this frame's pose and velocity are free), no robust kernel (:7797-7804)
                double e9[9], J[9 * 24];
                inertialError(E, K, F, e9);
                inertialJacobian(E, K, F, J);
                for (int a = 0; a < 9; a++) {
                    double s = 0;
                    for (int r = 0; r < 9; r++) { double oe = 0; for (int q = 0; q < 9; q++) oe += E.info[r * 9 + q] * e9[q]; s += J[r * 24 + 15 + a] * oe; }
                    b[a] += -s;
                    for (int c = 0; c < 9; c++) {
                        double t = 0;
                        for (int r = 0; r < 9; r++) { double oj = 0; for (int q = 0; q < 9; q++) oj += E.info[r * 9 + q] * J[q * 24 + 15 + c]; t += J[r * 24 + 15 + a] * oj; }
                        H[c * 15 + a] += t;
                    }
                }
                for (int w2 = 0; w2 < 2; w2++) {   // EdgeGyroRW / EdgeAccRW: e = b_frame - b_kf, J = I on the free vertex
                    const double* Om = w2 == 0 ? E.info_g : E.info_a;
                    const int o = 9 + 3 * w2;
                    double er[3];
                    for (int k = 0; k < 3; k++) er[k] = w2 == 0 ? F.bg[k] - K.bg[k] : F.ba[k] - K.ba[k];
                    for (int r = 0; r < 3; r++) {
                        double oe = 0; for (int q = 0; q < 3; q++) oe += Om[r * 3 + q] * er[q];
                        b[o + r] += -oe;
                        for (int c = 0; c < 3; c++) H[(o + c) * 15 + o + r] += Om[r * 3 + c];
                    }
                }
            }
            double x[15];
            memcpy(x, b, sizeof(x));
            if (!cholSolve15(H, 15, x)) break;   // cannot happen with positive definite inertial / random-walk information; mirrors `ok` ending the loop
            poseUpdate(F, rig, x);
            for (int i = 0; i < 3; i++) { F.v[i] += x[6 + i]; F.bg[i] += x[9 + i]; F.ba[i] += x[12 + i]; }
        }
        // classification (:7838-7896).  e->chi2() of an edge that took part in the optimisation is the value of the LAST computeActiveErrors,
        // i.e. at the state before the final update (Gauss-Newton does not re-evaluate); outlier edges are recomputed at the current state
        nBad = 0; nInliers = 0;
        const float chi2close = 1.5 * chi2Mono[it];
        for (int pass = 0; pass < 2; pass++)
            for (int i = 0; i < n_edges; i++) {
                const bool stereo = (edges[i].kind & 0xFF) == 1;
                if ((int)stereo != pass) continue;          // the reference walks vpEdgesMono first, then vpEdgesStereo
                poseEdgeLinearize(edges[i], F, rig, false, L);      // isDepthPositive() always reads the current estimate
                depthOk[i] = L.depthPositive;
                if (outlier[i]) chiLast[i] = L.chi2;
                const float chi2 = (float)chiLast[i];
                bool bad;
                if (!stereo) { const bool bClose = (edges[i].kind & 0x100) != 0; bad = (chi2 > chi2Mono[it] && !bClose) || (bClose && chi2 > chi2close) || !depthOk[i]; }
                else bad = chi2 > chi2Stereo[it];
                outlier[i] = bad; level[i] = bad;
                if (bad) nBad++; else nInliers++;
            }
        if (it == 2) robust = false;
        if (n_edges + 3 < 10) break;
    }
    if (nInliers < 30 && !rec_init) {   // :7904-7934
        nBad = 0;
        for (int i = 0; i < n_edges; i++) {
            poseEdgeLinearize(edges[i], F, rig, false, L);
            const bool stereo = (edges[i].kind & 0xFF) == 1;
            if ((float)L.chi2 < (stereo ? 24.f : 18.f)) outlier[i] = 0; else nBad++;
        }
    }
    // H of the final state (:8040-8060): EdgeInertial::GetHessian2 (columns of vertices 4, 5), the random-walk blocks, the inlier reprojection edges
    for (int i = 0; i < 225; i++) H15[i] = 0;
    {
        double J[9 * 24];
        inertialJacobian(E, K, F, J);
        for (int a = 0; a < 9; a++) for (int c = 0; c < 9; c++) {
            double t = 0;
            for (int r = 0; r < 9; r++) { double oj = 0; for (int q = 0; q < 9; q++) oj += E.info[r * 9 + q] * J[q * 24 + 15 + c]; t += J[r * 24 + 15 + a] * oj; }
            H15[a * 15 + c] += t;   // row-major 15x15
        }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { H15[(9 + r) * 15 + 9 + c] += E.info_g[r * 3 + c]; H15[(12 + r) * 15 + 12 + c] += E.info_a[r * 3 + c]; }
    }
    for (int i = 0; i < n_edges; i++) {
        if (outlier[i]) continue;
        poseEdgeLinearize(edges[i], F, rig, true, L);
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) { double t = 0; for (int d = 0; d < L.D; d++) t += L.B[d * 6 + r] * L.B[d * 6 + c]; H15[r * 15 + c] += (double)edges[i].inv_sigma2 * t; }
    }
    return n_edges - nBad;
}

// ---- Optimizer::PoseInertialOptimizationLastFrame (reference src/Optimizer.cc:8068-8415) ----------------------------------------------------
// As above, but the previous FRAME's four vertices are free too (30 unknowns) and tied down by EdgePriorPoseImu (G2oTypes.h:737-784, G2oTypes.cc:935-
// 968; Huber delta 5) built from the ConstraintPoseImu the previous call left; chi2Mono is 5.991 in all rounds; the final 30x30 Hessian is marginalised
// over the previous frame (Optimizer::Marginalize, :5366-5450: JacobiSVD pseudo-inverse, singular values <= 1e-6 dropped) into the next prior.
namespace {
struct Prior { double Rwb[9], twb[3], vwb[3], bg[3], ba[3], H[225]; };   // == liba_prior: ConstraintPoseImu (H row-major, as its constructor leaves it)
// EdgePriorPoseImu::computeError / linearizeOplus on key-frame state P (the previous frame): e[15], J 15x15 row-major over [pose 6 | v 3 | bg 3 | ba 3]
void priorLinearize(const Prior& C, const Kf& P, double* e, double* J) {
    double M[9], er[3];
    mulT33(C.Rwb, P.Rwb, M);              // Rwb^T * VP->estimate().Rwb
    logSO3(M, er);
    double d[3] = {P.twb[0] - C.twb[0], P.twb[1] - C.twb[1], P.twb[2] - C.twb[2]}, et[3];
    mulT31(C.Rwb, d, et);
    for (int i = 0; i < 3; i++) { e[i] = er[i]; e[3 + i] = et[i]; e[6 + i] = P.v[i] - C.vwb[i]; e[9 + i] = P.bg[i] - C.bg[i]; e[12 + i] = P.ba[i] - C.ba[i]; }
    if (J) {
        for (int i = 0; i < 225; i++) J[i] = 0.0;
        double iJ[9];
        invRightJacobianSO3(er, iJ);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { J[r * 15 + c] = iJ[r * 3 + c]; J[(3 + r) * 15 + 3 + c] = M[r * 3 + c]; }
        for (int k = 6; k < 15; k++) J[k * 15 + k] = 1.0;
    }
}
// pseudo-inverse of an n x n matrix by one-sided (Hestenes) Jacobi SVD in double; singular values <= 1e-6 dropped (Optimizer.cc:5406-5414)
void pinvJacobi(const double* Ain, int n, double* out) {
    std::vector<double> U(Ain, Ain + n * n), V(n * n, 0.0);   // row-major; columns of U converge to u_i * sigma_i
    for (int i = 0; i < n; i++) V[i * n + i] = 1.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < n; k++) { alpha += U[k * n + p] * U[k * n + p]; beta += U[k * n + q] * U[k * n + q]; gamma += U[k * n + p] * U[k * n + q]; }
                if (gamma == 0.0) continue;
                off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta + 1e-300));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < n; k++) {
                    const double up = U[k * n + p], uq = U[k * n + q]; U[k * n + p] = c * up - sn * uq; U[k * n + q] = sn * up + c * uq;
                    const double vp = V[k * n + p], vq = V[k * n + q]; V[k * n + p] = c * vp - sn * vq; V[k * n + q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    // A = U' S V^T with U'[:,i] = U[:,i] / s_i  ->  pinv = V S^-1 U'^T = sum_i V[:,i] U[:,i]^T / s_i^2
    for (int i = 0; i < n * n; i++) out[i] = 0.0;
    for (int i = 0; i < n; i++) {
        double s2 = 0;
        for (int k = 0; k < n; k++) s2 += U[k * n + i] * U[k * n + i];
        const double sv = std::sqrt(s2);
        if (!(sv > 1e-6)) continue;
        for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) out[r * n + c] += V[r * n + i] * U[c * n + i] / s2;
    }
}
const int kImuCol30[24] = {15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 0, 1, 2, 3, 4, 5, 6, 7, 8};   // EdgeInertial column -> unknown ([frame 15 | previous 15])
}  // namespace

extern "C" int oib_pose_inertial_lastframe(void* frame_, void* prev_, const void* rig_, const void* edges_, int n_edges, const void* imu_, const void* prior_,
                                           int rec_init, uint8_t* outlier, double* H15) {
    Kf& F = *(Kf*)frame_;
    Kf& Pv = *(Kf*)prev_;
    const Rig& rig = *(const Rig*)rig_;
    const PEdge* edges = (const PEdge*)edges_;
    const ImuEdge& E = *(const ImuEdge*)imu_;
    const Prior& C = *(const Prior*)prior_;
    const double thMono = (double)std::sqrt(5.991f), thStereo = (double)std::sqrt(7.815f);
    std::vector<double> chiLast(n_edges, 0.0);
    std::vector<char> level(n_edges, 0), depthOk(n_edges, 1);
    for (int i = 0; i < n_edges; i++) outlier[i] = 0;
    const float chi2Mono[4] = {5.991, 5.991, 5.991, 5.991}, chi2Stereo[4] = {15.6f, 9.8f, 7.815f, 7.815f};
    int nBad = 0, nInliers = 0;
    bool robust = true;
    PLin L;
    const int N = 30;
    auto addInertialAndPrior = [&](double* H, double* b, bool weighted) {   // H column-major N x N
        double e9[9], J[9 * 24];
        inertialError(E, Pv, F, e9);
        inertialJacobian(E, Pv, F, J);
        for (int a = 0; a < 24; a++) {
            if (b) { double s = 0; for (int r = 0; r < 9; r++) { double oe = 0; for (int q = 0; q < 9; q++) oe += E.info[r * 9 + q] * e9[q]; s += J[r * 24 + a] * oe; } b[kImuCol30[a]] += -s; }
            for (int c = 0; c < 24; c++) {
                double t = 0;
                for (int r = 0; r < 9; r++) { double oj = 0; for (int q = 0; q < 9; q++) oj += E.info[r * 9 + q] * J[q * 24 + c]; t += J[r * 24 + a] * oj; }
                H[kImuCol30[c] * N + kImuCol30[a]] += t;
            }
        }
        for (int w2 = 0; w2 < 2; w2++) {   // EdgeGyroRW / EdgeAccRW(previous, frame): e = b_frame - b_prev, J_prev = -I, J_frame = +I
            const double* Om = w2 == 0 ? E.info_g : E.info_a;
            const int of = 9 + 3 * w2, op = 24 + 3 * w2;
            double er[3];
            for (int k = 0; k < 3; k++) er[k] = w2 == 0 ? F.bg[k] - Pv.bg[k] : F.ba[k] - Pv.ba[k];
            for (int r = 0; r < 3; r++) {
                double oe = 0; for (int q = 0; q < 3; q++) oe += Om[r * 3 + q] * er[q];
                if (b) { b[of + r] += -oe; b[op + r] += oe; }
                for (int c = 0; c < 3; c++) {
                    H[(of + c) * N + of + r] += Om[r * 3 + c]; H[(op + c) * N + op + r] += Om[r * 3 + c];
                    H[(op + c) * N + of + r] += -Om[r * 3 + c]; H[(of + c) * N + op + r] += -Om[c * 3 + r];
                }
            }
        }
        {   // EdgePriorPoseImu on the previous frame, Huber delta 5 (:8244-8252)
            double e[15], J15[225], He[15];
            priorLinearize(C, Pv, e, J15);
            double chi = 0;
            for (int r = 0; r < 15; r++) { double s = 0; for (int q = 0; q < 15; q++) s += C.H[r * 15 + q] * e[q]; He[r] = s; chi += e[r] * s; }
            double rho1 = 1.0;
            if (weighted && chi > 25.0) rho1 = 5.0 / std::sqrt(chi);
            for (int a = 0; a < 15; a++) {
                if (b) { double s = 0; for (int r = 0; r < 15; r++) s += J15[r * 15 + a] * He[r]; b[15 + a] += -rho1 * s; }
                for (int c = 0; c < 15; c++) {
                    double t = 0;
                    for (int r = 0; r < 15; r++) { double hj = 0; for (int q = 0; q < 15; q++) hj += C.H[r * 15 + q] * J15[q * 15 + c]; t += J15[r * 15 + a] * hj; }
                    H[(15 + c) * N + 15 + a] += rho1 * t;
                }
            }
        }
    };
    for (int it = 0; it < 4; it++) {
        for (int gn = 0; gn < 10; gn++) {
            std::vector<double> H(N * N, 0.0), b(N, 0.0);
            for (int i = 0; i < n_edges; i++) {
                if (level[i]) continue;
                poseEdgeLinearize(edges[i], F, rig, true, L);
                chiLast[i] = L.chi2;
                double rho1 = 1.0;
                if (robust) { const double d = (edges[i].kind & 0xFF) == 1 ? thStereo : thMono, dsqr = d * d; if (L.chi2 > dsqr) rho1 = d / std::sqrt(L.chi2); }
                const double w = rho1 * (double)edges[i].inv_sigma2;
                for (int r = 0; r < 6; r++) {
                    double s = 0; for (int d = 0; d < L.D; d++) s += L.B[d * 6 + r] * L.e[d];
                    b[r] += -w * s;
                    for (int c = 0; c < 6; c++) { double t = 0; for (int d = 0; d < L.D; d++) t += L.B[d * 6 + r] * L.B[d * 6 + c]; H[c * N + r] += w * t; }
                }
            }
            addInertialAndPrior(H.data(), b.data(), true);
            std::vector<double> x = b;
            if (!cholSolve15(H.data(), N, x.data())) break;
            poseUpdate(F, rig, &x[0]);
            poseUpdate(Pv, rig, &x[15]);
            for (int i = 0; i < 3; i++) { F.v[i] += x[6 + i]; F.bg[i] += x[9 + i]; F.ba[i] += x[12 + i]; Pv.v[i] += x[21 + i]; Pv.bg[i] += x[24 + i]; Pv.ba[i] += x[27 + i]; }
        }
        nBad = 0; nInliers = 0;
        const float chi2close = 1.5 * chi2Mono[it];
        for (int i = 0; i < n_edges; i++) {
            const bool stereo = (edges[i].kind & 0xFF) == 1;
            poseEdgeLinearize(edges[i], F, rig, false, L);
            depthOk[i] = L.depthPositive;
            if (outlier[i]) chiLast[i] = L.chi2;
            const float chi2 = (float)chiLast[i];
            bool bad;
            if (!stereo) { const bool bClose = (edges[i].kind & 0x100) != 0; bad = (chi2 > chi2Mono[it] && !bClose) || (bClose && chi2 > chi2close) || !depthOk[i]; }
            else bad = chi2 > chi2Stereo[it];
            outlier[i] = bad; level[i] = bad;
            if (bad) nBad++; else nInliers++;
        }
        if (it == 2) robust = false;
        if (n_edges + 4 < 10) break;
    }
    if (nInliers < 30 && !rec_init) {
        nBad = 0;
        for (int i = 0; i < n_edges; i++) {
            poseEdgeLinearize(edges[i], F, rig, false, L);
            const bool stereo = (edges[i].kind & 0xFF) == 1;
            if ((float)L.chi2 < (stereo ? 24.f : 18.f)) outlier[i] = 0; else nBad++;
        }
    }
    // 30x30 Hessian of the final state (:8366-8400), marginalised over the previous frame (:8402) -> the frame's 15x15 prior
    std::vector<double> H(N * N, 0.0);
    addInertialAndPrior(H.data(), nullptr, false);
    for (int i = 0; i < n_edges; i++) {
        if (outlier[i]) continue;
        poseEdgeLinearize(edges[i], F, rig, true, L);
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) { double t = 0; for (int d = 0; d < L.D; d++) t += L.B[d * 6 + r] * L.B[d * 6 + c]; H[c * N + r] += (double)edges[i].inv_sigma2 * t; }
    }
    double Hpp[225], inv[225];
    for (int r = 0; r < 15; r++) for (int c = 0; c < 15; c++) Hpp[r * 15 + c] = H[(15 + c) * N + 15 + r];
    pinvJacobi(Hpp, 15, inv);
    for (int r = 0; r < 15; r++)
        for (int c = 0; c < 15; c++) {
            double s = 0;
            for (int k = 0; k < 15; k++) { double t = 0; for (int m = 0; m < 15; m++) t += inv[k * 15 + m] * H[c * N + 15 + m]; s += H[(15 + k) * N + r] * t; }   // H_fp * inv * H_pf
            H15[r * 15 + c] = H[c * N + r] - s;
        }
    return n_edges - nBad;
}
