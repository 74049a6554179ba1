// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
//
// CPU restatement of ORB-SLAM3's ORBextractor (reference src/ORBextractor.cc, include/ORBextractor.h)
// in dependency-free C++17.  The reference delegates its image arithmetic to OpenCV 3.x, which is NOT
// vendored and NOT installed here, and the reference ships no tests or golden vectors, so:
//
//     *** PARITY UNPINNED for the OpenCV-delegated steps ***
//     (cv::resize, cv::FAST, cv::GaussianBlur, cv::fastAtan2, cvRound).  They are restated from the
//     published OpenCV 3.2/3.3 plain-C++ (non-IPP, scalar) algorithms; see SURVEY.md Appendix B.
//
// Steps that live in the reference's own source are followed line by line and cited below.
// Two rules the reference leaves to the platform are fixed here (and matched by the HIP path):
//   R1 (octree tie-break)  reference sorts pair<int,ExtractorNode*> (ORBextractor.cc:679-683), i.e.
//      equal-size nodes are ordered by heap address.  Here: by creation sequence number, later-created
//      node first (what a monotonically growing heap gives).
//   R2 (sin/cos)  reference calls libm cosf/sinf (ORBextractor.cc:110-111).  Here: det_sincos(), a fixed
//      double-precision polynomial rounded once to float (bit-identical on host and device; equals a
//      correctly-rounded cosf/sinf except in astronomically rare double-rounding cases).
//
// Build: see oracle/Makefile (g++ -O3 -ffp-contract=off; no -ffast-math).
#include <algorithm>
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <chrono>
#include <list>
#include <thread>
#include <utility>
#include <vector>

namespace {

struct KeyPoint {  // cv::KeyPoint layout, 28 bytes
    float x, y, size, angle, response;
    int octave, class_id;
};

const int PATCH_SIZE = 31;       // ORBextractor.cc:70
const int HALF_PATCH_SIZE = 15;  // :71
const int EDGE_THRESHOLD = 19;   // :72

const int8_t kPattern[1024] = {
#include "orb_pattern_oracle.inc"
};

// ---- OpenCV scalar helpers (recalled semantics, Appendix B5) -------------------------------------------
inline int cvRound(float v) { return (int)lrintf(v); }    // round-half-even (default FP env)
inline int cvRound(double v) { return (int)lrint(v); }
inline int cvFloor(double v) { return (int)std::floor(v); }
inline int cvCeil(double v) { return (int)std::ceil(v); }
inline short sat_s16(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
inline int reflect101(int p, int len) {  // BORDER_REFLECT_101, valid for |overshoot| < len
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

// cv::fastAtan2 (Appendix B4), degrees in [0,360)
inline float fastAtan2(float y, float x) {
    const float s = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
    const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// Rule R2: deterministic sin/cos of a float angle (radians, any finite value of moderate size).
// Only IEEE +,-,* on doubles and one double->float rounding: identical on x86 and gfx950 when compiled
// without FMA contraction.
inline void det_sincos(float angle, float* s_out, float* c_out) {
    const double x = (double)angle;
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632673412561417e+00;  // first 33 bits of pi/2
    const double PIO2_LO = 6.07710050650619224932e-11;  // pi/2 - PIO2_HI
    const double kd = std::floor(x * TWO_OVER_PI + 0.5);
    const int k = (int)kd;
    const double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    const double z = r * r;
    // Taylor coefficients (|r| <= pi/4 + eps: truncation error < 1e-17)
    double ps = -1.0 / 355687428096000.0;            // -1/17!
    ps = ps * z + 1.0 / 1307674368000.0;             //  1/15!
    ps = ps * z - 1.0 / 6227020800.0;                // -1/13!
    ps = ps * z + 1.0 / 39916800.0;                  //  1/11!
    ps = ps * z - 1.0 / 362880.0;                    // -1/9!
    ps = ps * z + 1.0 / 5040.0;                      //  1/7!
    ps = ps * z - 1.0 / 120.0;                       // -1/5!
    ps = ps * z + 1.0 / 6.0;                         //  1/3!  (sign folded below)
    const double sr = r - r * z * ps;
    double pc = 1.0 / 20922789888000.0;              //  1/16!
    pc = pc * z - 1.0 / 87178291200.0;               // -1/14!
    pc = pc * z + 1.0 / 479001600.0;                 //  1/12!
    pc = pc * z - 1.0 / 3628800.0;                   // -1/10!
    pc = pc * z + 1.0 / 40320.0;                     //  1/8!
    pc = pc * z - 1.0 / 720.0;                       // -1/6!
    pc = pc * z + 1.0 / 24.0;                        //  1/4!
    pc = pc * z - 0.5;                               // -1/2!
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

// ---- cv::resize INTER_LINEAR, CV_8UC1 (Appendix B2) ----------------------------------------------------
void resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                      int dstride) {
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = std::min(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        ialpha[dx * 2] = sat_s16(cvRound((1.f - fx) * 2048));
        ialpha[dx * 2 + 1] = sat_s16(cvRound(fx * 2048));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = sat_s16(cvRound((1.f - fy) * 2048));
        ibeta[dy * 2 + 1] = sat_s16(cvRound(fy * 2048));
    }
    std::vector<int> r0(dw), r1(dw);
    auto hresize = [&](int sy, std::vector<int>& D) {
        sy = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);  // clip(sy, 0, sh)
        const uint8_t* S = src + (size_t)sy * sstride;
        int dx = 0;
        for (; dx < xmax; dx++) {
            int sx = xofs[dx];
            D[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
        }
        for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * 2048;
    };
    for (int dy = 0; dy < dh; dy++) {
        hresize(yofs[dy], r0);
        hresize(yofs[dy] + 1, r1);
        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

// ---- cv::GaussianBlur 7x7 sigma=2 BORDER_REFLECT_101, CV_8UC1 (Appendix B3) --------------------------------
void gaussian7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    // getGaussianKernel(7, 2, CV_32F) -> convertTo(CV_32S, 256): restated, not hard-coded
    int k[7];
    {
        double g[7], sum = 0;
        for (int i = 0; i < 7; i++) {
            double x = i - 3;
            g[i] = std::exp(-0.5 / (2.0 * 2.0) * x * x);
            g[i] = (double)(float)g[i];
            sum += g[i];
        }
        for (int i = 0; i < 7; i++) k[i] = cvRound((double)((float)(g[i] * (1. / sum))) * 256.0);
    }
    std::vector<int> rows((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int i = 0; i < 7; i++) acc += k[i] * S[reflect101(x + i - 3, w)];
            rows[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int j = 0; j < 7; j++) acc += k[j] * rows[(size_t)reflect101(y + j - 3, h) * w + x];
            int v = (acc + 32768) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}

// ---- cv::FAST 9_16 with NMS on a ROI (Appendix B1) ------------------------------------------------------
struct FastPt { int x, y, score; };

int cornerScore16(const uint8_t* ptr, const int pixel[25], int threshold) {
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[N];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

void fast9_16_nms(const uint8_t* img, int step, int cols, int rows, int threshold,
                  std::vector<FastPt>& out) {
    static const int offsets16[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1},
                                         {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                         {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};
    const int K = 8, N = 16 + K + 1;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = offsets16[k][0] + offsets16[k][1] * step;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    out.clear();
    threshold = std::min(std::max(threshold, 0), 255);
    uint8_t threshold_tab[512];
    for (int i = -255; i <= 255; i++)
        threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    if (cols < 7 || rows < 7) return;
    std::vector<uint8_t> bufmem((size_t)cols * 3, 0);
    std::vector<int> cpmem((size_t)(cols + 1) * 3, 0);
    uint8_t* buf[3] = {bufmem.data(), bufmem.data() + cols, bufmem.data() + 2 * cols};
    int* cpbuf[3] = {cpmem.data() + 1, cpmem.data() + (cols + 1) + 1, cpmem.data() + 2 * (cols + 1) + 1};
    for (int i = 3; i < rows - 2; i++) {
        const uint8_t* ptr = img + (size_t)i * step + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3];
        memset(curr, 0, cols);
        int ncorners = 0;
        if (i < rows - 3) {
            for (int j = 3; j < cols - 3; j++, ptr++) {
                int v = ptr[0];
                const uint8_t* tab = &threshold_tab[0] - v + 255;
                int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
                d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
                d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
                d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
                d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
                d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
                if (d & 1) {
                    int vt = v - threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x < vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                curr[j] = (uint8_t)cornerScore16(ptr, pixel, threshold);
                                break;
                            }
                        } else
                            count = 0;
                    }
                }
                if (d & 2) {
                    int vt = v + threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x > vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                curr[j] = (uint8_t)cornerScore16(ptr, pixel, threshold);
                                break;
                            }
                        } else
                            count = 0;
                    }
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; k++) {
            int j = cornerpos[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])
                out.push_back({j, i - 1, score});
        }
    }
}

// ---- ORBextractor restatement ---------------------------------------------------------------------------
struct ExtractorNode {  // ORBextractor.h:30-46
    std::vector<KeyPoint> vKeys;
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::list<ExtractorNode>::iterator lit;
    bool bNoMore = false;
    long seq = 0;  // rule R1: creation sequence (replaces the heap address in the sort key)

    void DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4) {
        // ORBextractor.cc:479-535
        const int halfX = (int)std::ceil(static_cast<float>(URx - ULx) / 2);
        const int halfY = (int)std::ceil(static_cast<float>(BRy - ULy) / 2);
        n1.ULx = ULx; n1.ULy = ULy;
        n1.URx = ULx + halfX; n1.URy = ULy;
        n1.BLx = ULx; n1.BLy = ULy + halfY;
        n1.BRx = ULx + halfX; n1.BRy = ULy + halfY;
        n2.ULx = n1.URx; n2.ULy = n1.URy;
        n2.URx = URx; n2.URy = URy;
        n2.BLx = n1.BRx; n2.BLy = n1.BRy;
        n2.BRx = URx; n2.BRy = ULy + halfY;
        n3.ULx = n1.BLx; n3.ULy = n1.BLy;
        n3.URx = n1.BRx; n3.URy = n1.BRy;
        n3.BLx = BLx; n3.BLy = BLy;
        n3.BRx = n1.BRx; n3.BRy = BLy;
        n4.ULx = n3.URx; n4.ULy = n3.URy;
        n4.URx = n2.BRx; n4.URy = n2.BRy;
        n4.BLx = n3.BRx; n4.BLy = n3.BRy;
        n4.BRx = BRx; n4.BRy = BRy;
        for (size_t i = 0; i < vKeys.size(); i++) {
            const KeyPoint& kp = vKeys[i];
            if (kp.x < n1.URx) {
                if (kp.y < n1.BRy) n1.vKeys.push_back(kp);
                else n3.vKeys.push_back(kp);
            } else if (kp.y < n1.BRy)
                n2.vKeys.push_back(kp);
            else
                n4.vKeys.push_back(kp);
        }
        if (n1.vKeys.size() == 1) n1.bNoMore = true;
        if (n2.vKeys.size() == 1) n2.bNoMore = true;
        if (n3.vKeys.size() == 1) n3.bNoMore = true;
        if (n4.vKeys.size() == 1) n4.bNoMore = true;
    }
};

struct OrbOracle {
    int nfeatures, nlevels, iniThFAST, minThFAST;
    double scaleFactor;  // double member, ORBextractor.h:96
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;

    // per-call state (exposed for stage-level parity)
    struct Level {
        int w = 0, h = 0;
        std::vector<uint8_t> img, blurred;
        std::vector<KeyPoint> toDistribute;  // vToDistributeKeys (ROI-relative to minBorder)
        std::vector<KeyPoint> kps;           // after octree + orientation, level coordinates
        std::vector<uint8_t> desc;
    };
    std::vector<Level> L;

    OrbOracle(int nf, float sf, int nl, int ini, int mn)
        : nfeatures(nf), nlevels(nl), iniThFAST(ini), minThFAST(mn), scaleFactor(sf) {
        // ORBextractor.cc:408-468
        mvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f;
        mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) {
            mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);
            mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
        }
        mvInvScaleFactor.resize(nlevels);
        mvInvLevelSigma2.resize(nlevels);
        for (int i = 0; i < nlevels; i++) {
            mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
            mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
        }
        mnFeaturesPerLevel.resize(nlevels);
        float factor = (float)(1.0f / scaleFactor);
        float nDesiredFeaturesPerScale =
            nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sumFeatures = 0;
        for (int level = 0; level < nlevels - 1; level++) {
            mnFeaturesPerLevel[level] = cvRound(nDesiredFeaturesPerScale);
            sumFeatures += mnFeaturesPerLevel[level];
            nDesiredFeaturesPerScale *= factor;
        }
        mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);
        umax.resize(HALF_PATCH_SIZE + 1);
        int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
        int vmin = cvCeil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = cvRound(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
        L.resize(nlevels);
    }

    void ComputePyramid(const uint8_t* image, int cols, int rows, int stride) {
        // ORBextractor.cc:1158-1183.  The 19-px BORDER_REFLECT_101 frame around every level is never read by
        // this stage; planes are stored un-bordered (bordered copies are produced on request).
        for (int level = 0; level < nlevels; ++level) {
            float scale = mvInvScaleFactor[level];
            int w = cvRound((float)cols * scale), h = cvRound((float)rows * scale);
            L[level].w = w;
            L[level].h = h;
            L[level].img.assign((size_t)w * h, 0);
            if (level != 0)
                resize_linear_u8(L[level - 1].img.data(), L[level - 1].w, L[level - 1].h, L[level - 1].w,
                                 L[level].img.data(), w, h, w);
            else
                for (int y = 0; y < rows; y++) memcpy(&L[0].img[(size_t)y * w], image + (size_t)y * stride, cols);
        }
    }

    float IC_Angle(const Level& lv, float ptx, float pty) const {
        // ORBextractor.cc:75-102
        int m_01 = 0, m_10 = 0;
        const int step = lv.w;
        const uint8_t* center = &lv.img[(size_t)cvRound(pty) * step + cvRound(ptx)];
        for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
        for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
            int v_sum = 0;
            int d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int val_plus = center[u + v * step], val_minus = center[u - v * step];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        return fastAtan2((float)m_01, (float)m_10);
    }

    std::vector<KeyPoint> DistributeOctTree(const std::vector<KeyPoint>& vToDistributeKeys, const int minX,
                                            const int maxX, const int minY, const int maxY, const int N) {
        // ORBextractor.cc:537-761
        long seqCounter = 0;
        const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
        const float hX = static_cast<float>(maxX - minX) / nIni;
        std::list<ExtractorNode> lNodes;
        std::vector<ExtractorNode*> vpIniNodes(nIni);
        for (int i = 0; i < nIni; i++) {
            ExtractorNode ni;
            ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
            ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
            ni.BLx = ni.ULx; ni.BLy = maxY - minY;
            ni.BRx = ni.URx; ni.BRy = maxY - minY;
            ni.seq = seqCounter++;
            lNodes.push_back(ni);
            vpIniNodes[i] = &lNodes.back();
        }
        for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
            const KeyPoint& kp = vToDistributeKeys[i];
            vpIniNodes[(size_t)(kp.x / hX)]->vKeys.push_back(kp);
        }
        auto lit = lNodes.begin();
        while (lit != lNodes.end()) {
            if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
            else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
            else lit++;
        }
        bool bFinish = false;
        typedef std::pair<int, ExtractorNode*> SP;
        auto less_r1 = [](const SP& a, const SP& b) {  // rule R1
            if (a.first != b.first) return a.first < b.first;
            return a.second->seq < b.second->seq;
        };
        std::vector<SP> vSizeAndPointerToNode;
        auto pushChild = [&](ExtractorNode& n, int* nToExpand) {
            if (n.vKeys.size() > 0) {
                n.seq = seqCounter++;
                lNodes.push_front(n);
                if (n.vKeys.size() > 1) {
                    if (nToExpand) (*nToExpand)++;
                    vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), &lNodes.front()));
                    lNodes.front().lit = lNodes.begin();
                }
            }
        };
        while (!bFinish) {
            int prevSize = (int)lNodes.size();
            lit = lNodes.begin();
            int nToExpand = 0;
            vSizeAndPointerToNode.clear();
            while (lit != lNodes.end()) {
                if (lit->bNoMore) { lit++; continue; }
                ExtractorNode n1, n2, n3, n4;
                lit->DivideNode(n1, n2, n3, n4);
                pushChild(n1, &nToExpand);
                pushChild(n2, &nToExpand);
                pushChild(n3, &nToExpand);
                pushChild(n4, &nToExpand);
                lit = lNodes.erase(lit);
            }
            if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
                bFinish = true;
            } else if (((int)lNodes.size() + nToExpand * 3) > N) {
                while (!bFinish) {
                    prevSize = (int)lNodes.size();
                    std::vector<SP> vPrev = vSizeAndPointerToNode;
                    vSizeAndPointerToNode.clear();
                    std::sort(vPrev.begin(), vPrev.end(), less_r1);
                    for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                        ExtractorNode n1, n2, n3, n4;
                        vPrev[j].second->DivideNode(n1, n2, n3, n4);
                        pushChild(n1, nullptr);
                        pushChild(n2, nullptr);
                        pushChild(n3, nullptr);
                        pushChild(n4, nullptr);
                        lNodes.erase(vPrev[j].second->lit);
                        if ((int)lNodes.size() >= N) break;
                    }
                    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
                }
            }
        }
        std::vector<KeyPoint> vResultKeys;
        vResultKeys.reserve(nfeatures);
        for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
            std::vector<KeyPoint>& vNodeKeys = it->vKeys;
            KeyPoint* pKP = &vNodeKeys[0];
            float maxResponse = pKP->response;
            for (size_t k = 1; k < vNodeKeys.size(); k++)
                if (vNodeKeys[k].response > maxResponse) {
                    pKP = &vNodeKeys[k];
                    maxResponse = vNodeKeys[k].response;
                }
            vResultKeys.push_back(*pKP);
        }
        return vResultKeys;
    }

    void ComputeKeyPointsOctTree() {
        // ORBextractor.cc:763-878
        const float W = 30;
        for (int level = 0; level < nlevels; ++level) {
            Level& lv = L[level];
            const int minBorderX = EDGE_THRESHOLD - 3;
            const int minBorderY = minBorderX;
            const int maxBorderX = lv.w - EDGE_THRESHOLD + 3;
            const int maxBorderY = lv.h - EDGE_THRESHOLD + 3;
            std::vector<KeyPoint>& vToDistributeKeys = lv.toDistribute;
            vToDistributeKeys.clear();
            lv.kps.clear();
            const float width = (float)(maxBorderX - minBorderX);
            const float height = (float)(maxBorderY - minBorderY);
            const int nCols = (int)(width / W);
            const int nRows = (int)(height / W);
            if (nCols <= 0 || nRows <= 0) continue;  // reference would divide by zero; levels this small are rejected upstream
            const int wCell = (int)std::ceil(width / nCols);
            const int hCell = (int)std::ceil(height / nRows);
            std::vector<FastPt> vKeysCell;
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minBorderY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBorderY - 3) continue;
                if (maxY > maxBorderY) maxY = (float)maxBorderY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minBorderX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBorderX - 6) continue;
                    if (maxX > maxBorderX) maxX = (float)maxBorderX;
                    const int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
                    const uint8_t* roi = &lv.img[(size_t)y0 * lv.w + x0];
                    fast9_16_nms(roi, lv.w, x1 - x0, y1 - y0, iniThFAST, vKeysCell);
                    if (vKeysCell.empty()) fast9_16_nms(roi, lv.w, x1 - x0, y1 - y0, minThFAST, vKeysCell);
                    for (const FastPt& p : vKeysCell) {
                        KeyPoint kp;
                        kp.x = (float)p.x + j * wCell;
                        kp.y = (float)p.y + i * hCell;
                        kp.size = 7.f;
                        kp.angle = -1.f;
                        kp.response = (float)p.score;
                        kp.octave = 0;
                        kp.class_id = -1;
                        vToDistributeKeys.push_back(kp);
                    }
                }
            }
            std::vector<KeyPoint>& keypoints = lv.kps;
            keypoints = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                          mnFeaturesPerLevel[level]);
            const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
            for (size_t i = 0; i < keypoints.size(); i++) {
                keypoints[i].x += minBorderX;
                keypoints[i].y += minBorderY;
                keypoints[i].octave = level;
                keypoints[i].size = (float)scaledPatchSize;
            }
        }
        for (int level = 0; level < nlevels; ++level)
            for (KeyPoint& kp : L[level].kps) kp.angle = IC_Angle(L[level], kp.x, kp.y);
    }

    void computeOrbDescriptor(const KeyPoint& kpt, const Level& lv, uint8_t* desc) const {
        // ORBextractor.cc:106-145 (on the blurred plane)
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        float angle = (float)kpt.angle * factorPI;
        float a, b;
        det_sincos(angle, &b, &a);  // rule R2: a = cos, b = sin
        const int step = lv.w;
        const uint8_t* center = &lv.blurred[(size_t)cvRound(kpt.y) * step + cvRound(kpt.x)];
        const int8_t* pattern = kPattern;
        auto GET = [&](int idx) -> int {
            float px = (float)pattern[idx * 2], py = (float)pattern[idx * 2 + 1];
            return center[cvRound(px * b + py * a) * step + cvRound(px * a - py * b)];
        };
        for (int i = 0; i < 32; ++i, pattern += 32) {
            int val = 0;
            for (int k = 0; k < 8; k++) {
                int t0 = GET(2 * k), t1 = GET(2 * k + 1);
                val |= (t0 < t1) << k;
            }
            desc[i] = (uint8_t)val;
        }
    }

    // ORBextractor::operator() — ORBextractor.cc:1074-1156.  Returns monoIndex (or -1 for an empty image).
    int extract(const uint8_t* image, int cols, int rows, int stride, int lap0, int lap1,
                std::vector<KeyPoint>& _keypoints, std::vector<uint8_t>& descriptors) {
        _keypoints.clear();
        descriptors.clear();
        if (!image || cols <= 0 || rows <= 0) return -1;
        // Geometries on which the reference itself has undefined behaviour are reported (-3) instead of reproduced: a level without a
        // FAST cell (width/30 == 0 -> division by zero, ORBextractor.cc:779-785) or with nIni == round(W/H) == 0 root nodes
        // (hX = W/0, vpIniNodes[kp.pt.x/hX] on an empty vector, :541-568).  The product rejects the same cases in orbx_create.
        for (int level = 0; level < nlevels; ++level) {
            const int w = cvRound((float)cols * mvInvScaleFactor[level]), h = cvRound((float)rows * mvInvScaleFactor[level]);
            const float fw = (float)(w - 2 * (EDGE_THRESHOLD - 3)), fh = (float)(h - 2 * (EDGE_THRESHOLD - 3));
            if (fw / 30.f < 1.f || fh / 30.f < 1.f || (int)std::round(fw / fh) < 1) return -3;
        }
        ComputePyramid(image, cols, rows, stride);
        ComputeKeyPointsOctTree();
        int nkeypoints = 0;
        for (int level = 0; level < nlevels; ++level) nkeypoints += (int)L[level].kps.size();
        _keypoints.assign(nkeypoints, KeyPoint{0, 0, 0, -1, 0, 0, -1});
        descriptors.assign((size_t)nkeypoints * 32, 0);
        int monoIndex = 0, stereoIndex = nkeypoints - 1;
        for (int level = 0; level < nlevels; ++level) {
            Level& lv = L[level];
            std::vector<KeyPoint>& keypoints = lv.kps;
            int nkeypointsLevel = (int)keypoints.size();
            lv.desc.assign((size_t)nkeypointsLevel * 32, 0);
            if (nkeypointsLevel == 0) continue;
            lv.blurred.assign((size_t)lv.w * lv.h, 0);
            gaussian7_u8(lv.img.data(), lv.w, lv.h, lv.w, lv.blurred.data(), lv.w);
            for (int i = 0; i < nkeypointsLevel; i++) computeOrbDescriptor(keypoints[i], lv, &lv.desc[(size_t)i * 32]);
            float scale = mvScaleFactor[level];
            int i = 0;
            for (auto keypoint = keypoints.begin(); keypoint != keypoints.end(); ++keypoint) {
                KeyPoint kp = *keypoint;  // lv.kps keeps level coordinates for stage-level checks
                if (level != 0) { kp.x *= scale; kp.y *= scale; }
                if (kp.x >= lap0 && kp.x <= lap1) {
                    _keypoints.at(stereoIndex) = kp;
                    memcpy(&descriptors[(size_t)stereoIndex * 32], &lv.desc[(size_t)i * 32], 32);
                    stereoIndex--;
                } else {
                    _keypoints.at(monoIndex) = kp;
                    memcpy(&descriptors[(size_t)monoIndex * 32], &lv.desc[(size_t)i * 32], 32);
                    monoIndex++;
                }
                i++;
            }
        }
        return monoIndex;
    }
};

}  // namespace

// ---- C API for ctypes (tests / smoke / cpu_baseline only) ---------------------------------------------
extern "C" {

void* oro_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) {
    return new OrbOracle(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
}
void oro_destroy(void* h) { delete (OrbOracle*)h; }

// tables: out arrays sized nlevels (scale, invScale, sigma2, invSigma2, featuresPerLevel) and 16 (umax)
void oro_tables(void* h, float* scale, float* invScale, float* sigma2, float* invSigma2, int* nfeat, int* umax16) {
    OrbOracle* o = (OrbOracle*)h;
    for (int i = 0; i < o->nlevels; i++) {
        scale[i] = o->mvScaleFactor[i];
        invScale[i] = o->mvInvScaleFactor[i];
        sigma2[i] = o->mvLevelSigma2[i];
        invSigma2[i] = o->mvInvLevelSigma2[i];
        nfeat[i] = o->mnFeaturesPerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax16[i] = o->umax[i];
}

// Full extraction.  kps: cap x 28 B, desc: cap x 32 B.  Returns monoIndex (-1 empty image, -2 cap too small).
int oro_extract(void* h, const uint8_t* img, int W, int H, int stride, int lap0, int lap1, void* kps,
                uint8_t* desc, int cap, int* n_out) {
    OrbOracle* o = (OrbOracle*)h;
    std::vector<KeyPoint> k;
    std::vector<uint8_t> d;
    int mono = o->extract(img, W, H, stride, lap0, lap1, k, d);
    *n_out = (int)k.size();
    if (mono < 0) return mono;
    if ((int)k.size() > cap) return -2;
    if (!k.empty()) {
        memcpy(kps, k.data(), k.size() * sizeof(KeyPoint));
        memcpy(desc, d.data(), d.size());
    }
    return mono;
}

// ---- stage-level accessors (valid after oro_extract) ----
int oro_level_size(void* h, int level, int* w, int* hh) {
    OrbOracle* o = (OrbOracle*)h;
    *w = o->L[level].w;
    *hh = o->L[level].h;
    return 0;
}
void oro_level_image(void* h, int level, uint8_t* out) {
    OrbOracle* o = (OrbOracle*)h;
    memcpy(out, o->L[level].img.data(), o->L[level].img.size());
}
int oro_level_blurred(void* h, int level, uint8_t* out) {  // returns 0 if level had no keypoints (not blurred)
    OrbOracle* o = (OrbOracle*)h;
    if (o->L[level].blurred.empty()) return 0;
    memcpy(out, o->L[level].blurred.data(), o->L[level].blurred.size());
    return 1;
}
int oro_level_candidates(void* h, int level, int* xys, int cap) {  // vToDistributeKeys: (x,y,score) triples
    OrbOracle* o = (OrbOracle*)h;
    const auto& v = o->L[level].toDistribute;
    int n = (int)v.size();
    for (int i = 0; i < n && i < cap; i++) {
        xys[i * 3] = (int)v[i].x;
        xys[i * 3 + 1] = (int)v[i].y;
        xys[i * 3 + 2] = (int)v[i].response;
    }
    return n;
}
int oro_level_keypoints(void* h, int level, void* kps, uint8_t* desc, int cap) {  // octree order, level coords
    OrbOracle* o = (OrbOracle*)h;
    const auto& v = o->L[level].kps;
    int n = (int)v.size();
    if (n <= cap && n > 0) {
        memcpy(kps, v.data(), (size_t)n * sizeof(KeyPoint));
        if (desc && !o->L[level].desc.empty()) memcpy(desc, o->L[level].desc.data(), (size_t)n * 32);
    }
    return n;
}

// ---- primitive-level entry points (property tests of the recalled OpenCV semantics) ----
void oro_resize_linear(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    resize_linear_u8(src, sw, sh, sw, dst, dw, dh, dw);
}
void oro_gaussian7(const uint8_t* src, int w, int h, uint8_t* dst) { gaussian7_u8(src, w, h, w, dst, w); }
int oro_fast(const uint8_t* img, int w, int h, int threshold, int* xys, int cap) {
    std::vector<FastPt> v;
    fast9_16_nms(img, w, w, h, threshold, v);
    int n = (int)v.size();
    for (int i = 0; i < n && i < cap; i++) {
        xys[i * 3] = v[i].x;
        xys[i * 3 + 1] = v[i].y;
        xys[i * 3 + 2] = v[i].score;
    }
    return n;
}
float oro_fast_atan2(float y, float x) { return fastAtan2(y, x); }
void oro_sincos(float a, float* s, float* c) { det_sincos(a, s, c); }
int oro_cvround(float v) { return cvRound(v); }
void oro_bordered_level(void* h, int level, uint8_t* out) {  // (w+38) x (h+38) BORDER_REFLECT_101 copy (mvImagePyramid parent)
    OrbOracle* o = (OrbOracle*)h;
    const auto& lv = o->L[level];
    const int B = EDGE_THRESHOLD, ow = lv.w + 2 * B, oh = lv.h + 2 * B;
    for (int y = 0; y < oh; y++)
        for (int x = 0; x < ow; x++)
            out[(size_t)y * ow + x] = lv.img[(size_t)reflect101(y - B, lv.h) * lv.w + reflect101(x - B, lv.w)];
}

// cpu_baseline leg of bench.py: `nthreads` workers, one extractor each, one frame per thread at a time (the reference
// extracts one image on one thread, Frame.cc:111-114); frames are taken round-robin from the B-frame sample.
// Returns wall seconds for nthreads*frames_per_thread extractions (std::chrono::steady_clock like Frame.cc:109-118).
double oro_bench_extract_mt(const uint8_t* frames, int B, int W, int H, int nfeatures, float scaleFactor, int nlevels,
                            int iniTh, int minTh, int lap0, int lap1, int nthreads, int frames_per_thread, long* kp_total) {
    std::vector<std::thread> th;
    std::vector<long> kp(nthreads, 0);
    auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([&, t] {
            OrbOracle o(nfeatures, scaleFactor, nlevels, iniTh, minTh);
            std::vector<KeyPoint> k;
            std::vector<uint8_t> d;
            for (int i = 0; i < frames_per_thread; i++) {
                const uint8_t* f = frames + (size_t)((t * frames_per_thread + i) % B) * W * H;
                o.extract(f, W, H, W, lap0, lap1, k, d);
                kp[t] += (long)k.size();
            }
        });
    for (auto& x : th) x.join();
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (kp_total) { *kp_total = 0; for (long v : kp) *kp_total += v; }
    return s;
}

// ---- Frame::ComputeStereoMatches (reference src/Frame.cc:955-1133) over two extractor states (left/right pyramids of the
// last oro_extract calls).  Follows the reference line by line; ORBmatcher::DescriptorDistance is restated inline.
static int descDist(const uint8_t* a, const uint8_t* b) {
    int d = 0;
    for (int i = 0; i < 32; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}
void oro_stereo_matches(void* hL, void* hR, const void* kpsL_, const uint8_t* descL, int N, const void* kpsR_, const uint8_t* descR,
                        int Nr, float mb, float mbf, float* mvuRight, float* mvDepth) {
    OrbOracle* L = (OrbOracle*)hL;
    OrbOracle* R = (OrbOracle*)hR;
    const KeyPoint* mvKeys = (const KeyPoint*)kpsL_;
    const KeyPoint* mvKeysRight = (const KeyPoint*)kpsR_;
    const std::vector<float>& mvScaleFactors = L->mvScaleFactor;
    const std::vector<float>& mvInvScaleFactors = L->mvInvScaleFactor;
    for (int i = 0; i < N; i++) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; }
    const int thOrbDist = (100 + 50) / 2;
    const int nRows = L->L[0].h;
    std::vector<std::vector<size_t>> vRowIndices(nRows);
    for (int iR = 0; iR < Nr; iR++) {
        const KeyPoint& kp = mvKeysRight[iR];
        const float& kpY = kp.y;
        const float r = 2.0f * mvScaleFactors[mvKeysRight[iR].octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);   // (the reference indexes unchecked; keypoints keep a 19-px margin)
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < N; iL++) {
        const KeyPoint& kpL = mvKeys[iL];
        const int& levelL = kpL.octave;
        const float& vL = kpL.y;
        const float& uL = kpL.x;
        const std::vector<size_t>& vCandidates = vRowIndices[(size_t)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = 100;
        size_t bestIdxR = 0;
        const uint8_t* dL = descL + (size_t)iL * 32;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            const KeyPoint& kpR = mvKeysRight[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float& uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = descDist(dL, descR + iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = mvKeysRight[bestIdxR].x;
            const float scaleFactor = mvInvScaleFactors[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor);
            const float scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5;
            const OrbOracle::Level& PL = L->L[kpL.octave];
            const OrbOracle::Level& PR = R->L[kpL.octave];
            auto pix = [](const OrbOracle::Level& P, int y, int x) -> int { return P.img[(size_t)reflect101(y, P.h) * P.w + reflect101(x, P.w)]; };
            short IL[11][11];
            const int cL = pix(PL, (int)scaledvL, (int)scaleduL);
            for (int a = 0; a < 11; a++)
                for (int b = 0; b < 11; b++) IL[a][b] = (short)(pix(PL, (int)scaledvL - w + a, (int)scaleduL - w + b) - cL);
            int bestDist2 = 2147483647, bestincR = 0;
            const int Lw = 5;
            std::vector<float> vDists(2 * Lw + 1);
            const float iniu = scaleduR0 + Lw - w, endu = scaleduR0 + Lw + w + 1;
            if (iniu < 0 || endu >= PR.w) continue;
            for (int incR = -Lw; incR <= +Lw; incR++) {
                const int cR = pix(PR, (int)scaledvL, (int)scaleduR0 + incR);
                double sum = 0;
                for (int a = 0; a < 11; a++)
                    for (int b = 0; b < 11; b++) {
                        const short v = (short)(pix(PR, (int)scaledvL - w + a, (int)scaleduR0 + incR - w + b) - cR);
                        sum += std::abs((int)IL[a][b] - (int)v);
                    }
                float dist = (float)sum;
                if (dist < bestDist2) { bestDist2 = (int)dist; bestincR = incR; }
                vDists[Lw + incR] = dist;
            }
            if (bestincR == -Lw || bestincR == Lw) continue;
            const float dist1 = vDists[Lw + bestincR - 1], dist2 = vDists[Lw + bestincR], dist3 = vDists[Lw + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = mvScaleFactors[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDist2, iL));
            }
        }
    }
    if (vDistIdx.empty()) return;   // (the reference reads vDistIdx[0] unconditionally)
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        mvuRight[vDistIdx[i].second] = -1;
        mvDepth[vDistIdx[i].second] = -1;
    }
}

}  // extern "C"