// frame_oracle.cpp — CPU restatement of the Frame-constructor steps between extractor and matcher.
// TEST INFRASTRUCTURE ONLY (oracle/README.md): used by tests/ to check the HIP path; never shipped, never measured as the product.
//
// PARITY UNPINNED for ofr_undistort: cv::undistortPoints lives in OpenCV (un-vendored, absent from this image).  Restated from the
// published OpenCV 3.x algorithm (cvUndistortPoints: double arithmetic, 5 fixed iterations, P = K re-projection), anchored on the
// reference's call sites Frame.cc:903 and :937.  Written independently of the product kernel (different expression layout, all 12
// coefficient slots present) so that agreement is evidence, not tautology.
#include <cmath>
#include <cstdint>
#include <cstring>

namespace {
struct KeyPoint { float x, y, size, angle, response; int32_t octave, class_id; };

void undistort(const float K[4], const float dist5[5], float xin, float yin, float* xo, float* yo) {
    double A[3][3] = {{(double)K[0], 0, (double)K[2]}, {0, (double)K[1], (double)K[3]}, {0, 0, 1}};
    double k[14] = {0};
    for (int i = 0; i < 5; i++) k[i] = (double)dist5[i];   // (k1, k2, p1, p2, k3); k[5..13] = 0
    double RR[3][3];
    // RR = P * R, R = I, P = K
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int t = 0; t < 3; t++) s += A[i][t] * (t == j ? 1.0 : 0.0); RR[i][j] = s; }
    const double fx = A[0][0], fy = A[1][1], ifx = 1. / fx, ify = 1. / fy, cx = A[0][2], cy = A[1][2];
    double x = xin, y = yin;
    double x0 = x = (x - cx) * ifx;
    double y0 = y = (y - cy) * ify;
    for (int j = 0; j < 5; j++) {
        double r2 = x * x + y * y;
        double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
    double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
    double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
    *xo = (float)(xx * ww);
    *yo = (float)(yy * ww);
}
}  // namespace

namespace {
// ---- Frame::ComputeStereoFishEyeMatches (Frame.cc:1281-1325) + KannalaBrandt8::TriangulateMatches (KannalaBrandt8.cpp:334-400) ----
// PARITY UNPINNED: float32 cv::Mat arithmetic, libm float functions and cv::SVD::compute on a 4x4 CV_32F matrix.  Rule R4 (shared with the
// product): a transcendental function is evaluated in double on the float argument and rounded to float where the reference holds a float;
// a float matrix product is the double-accumulated sum rounded once; the last right singular vector of A is the eigenvector of the smallest
// eigenvalue of A^T A by cyclic Jacobi in double (8 sweeps), rounded to float (its sign cancels in x3D = v[0..2] / v[3]).
inline int popc8(const uint8_t* a, const uint8_t* b) {
    int d = 0;
    for (int i = 0; i < 8; i++) { uint32_t x, y; memcpy(&x, a + 4 * i, 4); memcpy(&y, b + 4 * i, 4); d += __builtin_popcount(x ^ y); }
    return d;
}
void kb8Unproject(const float* p, float u, float v, float* ray) {   // KannalaBrandt8.cpp:101-124, precision = 1e-6
    const float pwx = (u - p[2]) / p[0], pwy = (v - p[3]) / p[1];
    float scale = 1.f;
    float theta_d = sqrtf(pwx * pwx + pwy * pwy);
    theta_d = fminf(fmaxf(-(float)(M_PI / 2.0), theta_d), (float)(M_PI / 2.0));   // CV_PI / 2.f in float context
    if (theta_d > 1e-8) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
            const float k0_theta2 = p[4] * theta2, k1_theta4 = p[5] * theta4, k2_theta6 = p[6] * theta6, k3_theta8 = p[7] * theta8;
            const float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                    (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < 1e-6f) break;
        }
        scale = (float)std::tan((double)theta) / theta_d;
    }
    ray[0] = pwx * scale; ray[1] = pwy * scale; ray[2] = 1.f;
}
void kb8ProjectF(const float* p, const float* X, float* uv) {   // KannalaBrandt8.cpp:28-42 (cv::Point3f overload)
    const float x2_plus_y2 = X[0] * X[0] + X[1] * X[1];
    const float theta = (float)std::atan2((double)sqrtf(x2_plus_y2), (double)X[2]);
    const float psi = (float)std::atan2((double)X[1], (double)X[0]);
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const float r = theta + p[4] * theta3 + p[5] * theta5 + p[6] * theta7 + p[7] * theta9;
    uv[0] = (float)((double)(p[0] * r) * std::cos((double)psi) + (double)p[2]);
    uv[1] = (float)((double)(p[1] * r) * std::sin((double)psi) + (double)p[3]);
}
void nullVector4(const float* A, float* v4) {   // last row of vt of cv::SVD::compute(A): rule R4
    double M[16], V[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += (double)A[k * 4 + i] * (double)A[k * 4 + j]; M[i * 4 + j] = s; V[i * 4 + j] = i == j; }
    for (int sweep = 0; sweep < 8; sweep++)
        for (int pI = 0; pI < 3; pI++)
            for (int q = pI + 1; q < 4; q++) {
                const double apq = M[pI * 4 + q];
                if (apq == 0.0) continue;
                const double th = (M[q * 4 + q] - M[pI * 4 + pI]) / (2.0 * apq);
                const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 4; k++) { const double a = M[k * 4 + pI], b = M[k * 4 + q]; M[k * 4 + pI] = c * a - sn * b; M[k * 4 + q] = sn * a + c * b; }
                for (int k = 0; k < 4; k++) { const double a = M[pI * 4 + k], b = M[q * 4 + k]; M[pI * 4 + k] = c * a - sn * b; M[q * 4 + k] = sn * a + c * b; }
                for (int k = 0; k < 4; k++) { const double a = V[k * 4 + pI], b = V[k * 4 + q]; V[k * 4 + pI] = c * a - sn * b; V[k * 4 + q] = sn * a + c * b; }
            }
    int m = 0;
    for (int i = 1; i < 4; i++) if (M[i * 4 + i] < M[m * 4 + m]) m = i;
    for (int k = 0; k < 4; k++) v4[k] = (float)V[k * 4 + m];
}
inline float fdot3(const float* a, const float* b) { return (float)((double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2]); }
// returns z1 (> 0) and x3D, or -1
float triangulateMatches(const float* p1, const float* p2, const KeyPoint& kp1, const KeyPoint& kp2, const float* R12, const float* t12, float sigmaLevel,
                         float unc, float* x3D) {
    float r1[3], r2[3], r21[3];
    kb8Unproject(p1, kp1.x, kp1.y, r1);
    kb8Unproject(p2, kp2.x, kp2.y, r2);
    for (int i = 0; i < 3; i++) r21[i] = fdot3(R12 + 3 * i, r2);
    // cv::norm of a CV_32F Mat accumulates in double and returns double
    const double n1 = std::sqrt((double)r1[0] * r1[0] + (double)r1[1] * r1[1] + (double)r1[2] * r1[2]);
    const double n2 = std::sqrt((double)r21[0] * r21[0] + (double)r21[1] * r21[1] + (double)r21[2] * r21[2]);
    const double dotp = (double)r1[0] * r21[0] + (double)r1[1] * r21[1] + (double)r1[2] * r21[2];   // Mat::dot returns double
    const float cosParallaxRays = (float)(dotp / (n1 * n2));
    if (cosParallaxRays > 0.9998) return -1;
    float R21[9], t21[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R21[i * 3 + j] = R12[j * 3 + i];
    for (int i = 0; i < 3; i++) t21[i] = -fdot3(R21 + 3 * i, t12);
    const float Tcw1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    float Tcw2[12];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Tcw2[i * 4 + j] = R21[i * 3 + j]; Tcw2[i * 4 + 3] = t21[i]; }
    float A[16];
    for (int c = 0; c < 4; c++) {
        A[c] = r1[0] * Tcw1[8 + c] - Tcw1[c];
        A[4 + c] = r1[1] * Tcw1[8 + c] - Tcw1[4 + c];
        A[8 + c] = r2[0] * Tcw2[8 + c] - Tcw2[c];
        A[12 + c] = r2[1] * Tcw2[8 + c] - Tcw2[4 + c];
    }
    float v4[4];
    nullVector4(A, v4);
    for (int i = 0; i < 3; i++) x3D[i] = v4[i] / v4[3];
    const float z1 = x3D[2];
    if (z1 <= 0) return -1;
    const float z2 = fdot3(R21 + 6, x3D) + t21[2];
    if (z2 <= 0) return -1;
    float uv1[2];
    kb8ProjectF(p1, x3D, uv1);
    const float errX1 = uv1[0] - kp1.x, errY1 = uv1[1] - kp1.y;
    if ((errX1 * errX1 + errY1 * errY1) > 5.991 * sigmaLevel) return -1;
    float x3D2[3], uv2[2];
    for (int i = 0; i < 3; i++) x3D2[i] = fdot3(R21 + 3 * i, x3D) + t21[i];
    kb8ProjectF(p2, x3D2, uv2);
    const float errX2 = uv2[0] - kp2.x, errY2 = uv2[1] - kp2.y;
    if ((errX2 * errX2 + errY2 * errY2) > 5.991 * unc) return -1;
    return z1;
}
}  // namespace

extern "C" {
// KannalaBrandt8::TriangulateMatches for the matcher oracle (SearchForTriangulation on fisheye key frames): returns z1 or -1
float ofr_triangulate_matches(const float* p1, const float* p2, const void* kp1, const void* kp2, const float* R12, const float* t12, float sigmaLevel,
                              float unc, float* x3D) {
    return triangulateMatches(p1, p2, *(const KeyPoint*)kp1, *(const KeyPoint*)kp2, R12, t12, sigmaLevel, unc, x3D);
}
float ofr_kb8_project(const float* p, const float* X, float* uv) { kb8ProjectF(p, X, uv); return uv[0]; }

// Frame::UndistortKeyPoints (Frame.cc:874-925).  cam = fx, fy, cx, cy, k1, k2, p1, p2, k3
void ofr_undistort_keypoints(const void* kps_in, int n, const float cam[9], void* kps_out) {
    const KeyPoint* in = (const KeyPoint*)kps_in;
    KeyPoint* out = (KeyPoint*)kps_out;
    for (int i = 0; i < n; i++) {
        KeyPoint kp = in[i];
        if (cam[4] != 0.0f) undistort(cam, cam + 4, in[i].x, in[i].y, &kp.x, &kp.y);
        out[i] = kp;
    }
}

// Frame::ComputeImageBounds (Frame.cc:926-953) + grid scalars (Frame.cc:394-397): out = mnMinX, mnMaxX, mnMinY, mnMaxY, wInv, hInv
void ofr_image_bounds(const float cam[9], int cols, int rows, float out[6]) {
    if (cam[4] != 0.0f) {
        float m[4][2] = {{0.0f, 0.0f}, {(float)cols, 0.0f}, {0.0f, (float)rows}, {(float)cols, (float)rows}};
        for (auto& p : m) undistort(cam, cam + 4, p[0], p[1], &p[0], &p[1]);
        out[0] = std::fmin(m[0][0], m[2][0]); out[1] = std::fmax(m[1][0], m[3][0]);
        out[2] = std::fmin(m[0][1], m[1][1]); out[3] = std::fmax(m[2][1], m[3][1]);
    } else {
        out[0] = 0.0f; out[1] = (float)cols; out[2] = 0.0f; out[3] = (float)rows;
    }
    out[4] = (float)64 / (float)(out[1] - out[0]);
    out[5] = (float)48 / (float)(out[3] - out[2]);
}

// Frame::ComputeStereoFromRGBD (Frame.cc:1136-1157)
void ofr_stereo_from_rgbd(const void* kps, const void* kps_un, int n, const float* depth, int row_stride, float mbf, float* u_right, float* depth_out) {
    const KeyPoint* k = (const KeyPoint*)kps;
    const KeyPoint* ku = (const KeyPoint*)kps_un;
    for (int i = 0; i < n; i++) {
        u_right[i] = -1; depth_out[i] = -1;
        const float v = k[i].y, u = k[i].x;
        const float d = depth[(size_t)(int)v * row_stride + (int)u];
        if (d > 0) { depth_out[i] = d; u_right[i] = ku[i].x - mbf / d; }
    }
}
// Frame::ComputeStereoFishEyeMatches (Frame.cc:1281-1325).  rig = left params[8], right params[8], mRlr[9] (row-major), mtlr[3].
// Outputs sized Nleft / Nright: mvLeftToRightMatch, mvRightToLeftMatch, mvDepth, mvStereo3Dpoints (zeros where empty).  Returns nMatches.
int ofr_stereo_fisheye(const void* kl_, const uint8_t* dl, int nleft, int monoLeft, const void* kr_, const uint8_t* dr, int nright, int monoRight,
                       const float* rig, const float* levelSigma2, int32_t* l2r, int32_t* r2l, float* depth, float* p3d) {
    const KeyPoint* kl = (const KeyPoint*)kl_;
    const KeyPoint* kr = (const KeyPoint*)kr_;
    for (int i = 0; i < nleft; i++) { l2r[i] = -1; depth[i] = -1.0f; p3d[3 * i] = p3d[3 * i + 1] = p3d[3 * i + 2] = 0.f; }
    for (int i = 0; i < nright; i++) r2l[i] = -1;
    int nMatches = 0;
    const int nq = nleft - monoLeft, nt = nright - monoRight;
    for (int q = 0; q < nq; q++) {
        // BFMatcher(NORM_HAMMING).knnMatch(k = 2): ascending train scan, strict '<' (the rule of omo_knn2)
        int d0 = 256, d1 = 256, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            const int d = popc8(dl + (size_t)(q + monoLeft) * 32, dr + (size_t)(j + monoRight) * 32);
            if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
            else if (d < d1) { d1 = d; i1 = j; }
        }
        if (i1 < 0) continue;                                   // (*it).size() >= 2
        if (!((float)d0 < (float)d1 * 0.7)) continue;           // float distance * double 0.7
        const KeyPoint& k1 = kl[q + monoLeft];
        const KeyPoint& k2 = kr[i0 + monoRight];
        float x3D[3];
        const float dep = triangulateMatches(rig, rig + 8, k1, k2, rig + 16, rig + 25, levelSigma2[k1.octave], levelSigma2[k2.octave], x3D);
        if (dep > 0.0001f) {
            l2r[q + monoLeft] = i0 + monoRight;
            r2l[i0 + monoRight] = q + monoLeft;
            for (int c = 0; c < 3; c++) p3d[3 * (q + monoLeft) + c] = x3D[c];
            depth[q + monoLeft] = dep;
            nMatches++;
        }
    }
    return nMatches;
}
}
