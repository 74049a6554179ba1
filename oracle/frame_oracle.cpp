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

extern "C" {
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
}
