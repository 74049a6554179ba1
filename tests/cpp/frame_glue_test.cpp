// Compile-and-run check of integration/Frame_hip.cc (Frame::ComputeStereoMatches / UndistortKeyPoints / ComputeStereoFishEyeMatches under the
// reference's member signatures) against the mock declarations of tests/cpp/mock_orbslam3, with the extractor adapter's -DORBHIP_WITH_OPENCV
// branch playing ORB_SLAM3::ORBextractor exactly as in an integrated tree.  Expected values: the oracle's restatements of the three reference
// loops (oracle/orb_oracle.cpp, oracle/frame_oracle.cpp) on the same inputs, bitwise.
// Built by tests/test_glue.py against the emulated library (CPU tier) or the real liborbhip.so (GPU tier).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "Frame.h"

extern "C" {
void* oro_create(int, float, int, int, int);
void oro_destroy(void*);
int oro_extract(void*, const uint8_t*, int, int, int, int, int, void*, uint8_t*, int, int*);
void oro_stereo_matches(void* hL, void* hR, const void* kpsL, const uint8_t* descL, int N, const void* kpsR, const uint8_t* descR, int Nr, float mb, float mbf,
                        float* mvuRight, float* mvDepth);
void ofr_undistort_keypoints(const void* kps, int n, const float* cam9, void* out);
int ofr_stereo_fisheye(const void* kl, const uint8_t* dl, int nl, int monoL, const void* kr, const uint8_t* dr, int nr, int monoR, const float* rig28,
                       const float* levelSigma2, int32_t* l2r, int32_t* r2l, float* depth, float* p3d);
}

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

using namespace ORB_SLAM3;
float Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv;

static cv::Mat scene_image(int W, int H, unsigned seed) {
    cv::Mat img(H, W, CV_8UC1);
    std::memset(img.data, 105, (size_t)W * H);
    unsigned st = seed;
    auto r = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
    for (int k = 0; k < 260; k++) {
        const int cx = r() % W, cy = r() % H, hw = 3 + r() % 22, hh = 3 + r() % 22, g = 20 + r() % 215;
        for (int y = cy - hh; y <= cy + hh; y++)
            for (int x = cx - hw; x <= cx + hw; x++)
                if (x >= 0 && x < W && y >= 0 && y < H) img.data[(size_t)y * W + x] = (unsigned char)g;
    }
    return img;
}

int main() {
    const int W = 480, H = 360;
    const float fx = 458.654f, bf = 47.90639384423901f;
    // ---- 1. Frame::ComputeStereoMatches: a rectified pair with three disparity bands ----
    {
        cv::Mat left = scene_image(W, H, 99u), right(H, W, CV_8UC1);
        unsigned st = 5u;
        for (int y = 0; y < H; y++) {
            const int d = y < H / 3 ? 7 : (y < 2 * H / 3 ? 19 : 33);
            for (int x = 0; x < W; x++) {
                st = st * 1664525u + 1013904223u;
                const int v = (int)left.data[(size_t)y * W + (x + d) % W] + (int)((st >> 28) % 5) - 2;
                right.data[(size_t)y * W + x] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
        ORBextractor exL(600, 1.2f, 8, 20, 7), exR(600, 1.2f, 8, 20, 7);
        Frame F;
        std::vector<int> lap = {0, 0};
        exL(left, cv::Mat(), F.mvKeys, F.mDescriptors, lap);        // Frame::ExtractORB (Frame.cc:488-495) on both images
        exR(right, cv::Mat(), F.mvKeysRight, F.mDescriptorsRight, lap);
        F.N = (int)F.mvKeys.size();
        F.mpORBextractorLeft = &exL; F.mpORBextractorRight = &exR;
        F.mbf = bf; F.mb = bf / fx;
        CHECK(F.N > 300 && F.mvKeysRight.size() > 300);
        CHECK(exL.mvImagePyramid.empty() && !exL.keepHostPyramid());   // Frame_hip.cc is linked: no host pyramid is produced (its only reader runs on the device)
        F.ComputeStereoMatches();                                       // key points / descriptors / pyramids where the two calls left them on the device
        // the oracle on its own extraction of the same two images (extractor parity is bit-exact, so its key points are the frame's)
        void *oL = oro_create(600, 1.2f, 8, 20, 7), *oR = oro_create(600, 1.2f, 8, 20, 7);
        std::vector<cv::KeyPoint> kl(4 * 600), kr(4 * 600);
        std::vector<uint8_t> dl((size_t)4 * 600 * 32), dr((size_t)4 * 600 * 32);
        int nl = 0, nr = 0;
        oro_extract(oL, left.data, W, H, W, 0, 0, kl.data(), dl.data(), 4 * 600, &nl);
        oro_extract(oR, right.data, W, H, W, 0, 0, kr.data(), dr.data(), 4 * 600, &nr);
        CHECK(nl == F.N && nr == (int)F.mvKeysRight.size());
        CHECK(std::memcmp(kl.data(), F.mvKeys.data(), (size_t)nl * sizeof(cv::KeyPoint)) == 0 && std::memcmp(dl.data(), F.mDescriptors.data, (size_t)nl * 32) == 0);
        std::vector<float> our(nl), odp(nl);
        oro_stereo_matches(oL, oR, kl.data(), dl.data(), nl, kr.data(), dr.data(), nr, F.mb, F.mbf, our.data(), odp.data());
        oro_destroy(oL); oro_destroy(oR);
        CHECK((int)F.mvuRight.size() == F.N && (int)F.mvDepth.size() == F.N);
        CHECK(std::memcmp(F.mvuRight.data(), our.data(), (size_t)nl * 4) == 0 && std::memcmp(F.mvDepth.data(), odp.data(), (size_t)nl * 4) == 0);   // bitwise
        int nm = 0, nband = 0;
        for (int i = 0; i < F.N; i++)
            if (F.mvuRight[i] >= 0) {
                nm++;
                const int band = std::min((int)F.mvKeys[i].pt.y / (H / 3), 2);
                nband += std::fabs(F.mvKeys[i].pt.x - F.mvuRight[i] - (band == 0 ? 7.f : (band == 1 ? 19.f : 33.f))) < 1.5f;
            }
        CHECK(nm > 100 && nband > nm * 8 / 10);   // the planted disparities are recovered
        // the general path (inputs that are not the last call's outputs — here one untouched-by-stereo field differs — are uploaded): same floats
        {
            Frame U = F;
            U.mvKeysRight[0].class_id = 7;
            U.mvuRight.clear(); U.mvDepth.clear();
            U.ComputeStereoMatches();
            CHECK(std::memcmp(U.mvuRight.data(), our.data(), (size_t)nl * 4) == 0 && std::memcmp(U.mvDepth.data(), odp.data(), (size_t)nl * 4) == 0);
        }
        // no right key points: every entry stays -1
        Frame G = F;
        G.mvKeysRight.clear(); G.mDescriptorsRight = cv::Mat();
        G.ComputeStereoMatches();
        for (int i = 0; i < G.N; i++) CHECK(G.mvuRight[i] == -1.f && G.mvDepth[i] == -1.f);
        std::printf("frame glue ComputeStereoMatches: %d of %d key points matched, %d on their band's disparity\n", nm, F.N, nband);

        // ---- 2. Frame::UndistortKeyPoints on the same key points: EuRoC radial-tangential coefficients (Examples/Monocular/EuRoC.yaml:14-17) ----
        Pinhole cam(458.654f, 457.296f, 367.215f, 248.375f);
        F.mpCamera = &cam;
        F.mK = cam.toK();
        F.mDistCoef = cv::Mat(4, 1, CV_32F);
        const float dist[4] = {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f};
        for (int i = 0; i < 4; i++) F.mDistCoef.at<float>(i) = dist[i];
        F.UndistortKeyPoints();
        const float cam9[9] = {458.654f, 457.296f, 367.215f, 248.375f, dist[0], dist[1], dist[2], dist[3], 0.f};
        std::vector<cv::KeyPoint> oun(F.N);
        ofr_undistort_keypoints(F.mvKeys.data(), F.N, cam9, oun.data());
        CHECK((int)F.mvKeysUn.size() == F.N && std::memcmp(F.mvKeysUn.data(), oun.data(), (size_t)F.N * sizeof(cv::KeyPoint)) == 0);   // bitwise
        float moved = 0;
        for (int i = 0; i < F.N; i++) {
            moved = std::max(moved, std::fabs(F.mvKeysUn[i].pt.x - F.mvKeys[i].pt.x));
            CHECK(F.mvKeysUn[i].octave == F.mvKeys[i].octave && F.mvKeysUn[i].angle == F.mvKeys[i].angle && F.mvKeysUn[i].response == F.mvKeys[i].response);
        }
        CHECK(moved > 1.f);
        F.UndistortKeyPoints();   // the cached camera record serves the second call
        CHECK(std::memcmp(F.mvKeysUn.data(), oun.data(), (size_t)F.N * sizeof(cv::KeyPoint)) == 0);
        F.mDistCoef.at<float>(0) = 0.f;   // k1 == 0: plain copy (Frame.cc:879-883)
        F.UndistortKeyPoints();
        CHECK(std::memcmp(F.mvKeysUn.data(), F.mvKeys.data(), (size_t)F.N * sizeof(cv::KeyPoint)) == 0);
        std::printf("frame glue UndistortKeyPoints: %d key points, largest shift %.2f px\n", F.N, moved);
    }
    // ---- 3. Frame::ComputeStereoFishEyeMatches: a fisheye rig looking at one plane; the right camera sees the left image shifted ----
    {
        const int FW = 512, FH = 512;
        cv::Mat left = scene_image(FW, FH, 1234u), right(FH, FW, CV_8UC1);
        for (int y = 0; y < FH; y++)
            for (int x = 0; x < FW; x++) right.data[(size_t)y * FW + x] = left.data[(size_t)y * FW + (x + 11) % FW];
        ORBextractor exL(500, 1.2f, 8, 20, 7), exR(500, 1.2f, 8, 20, 7);
        Frame F;
        std::vector<int> lapL = {150, 511}, lapR = {0, 360};   // vLappingArea of the two cameras (TUM-VI yaml: Camera.lappingBegin / lappingEnd)
        F.monoLeft = exL(left, cv::Mat(), F.mvKeys, F.mDescriptors, lapL);
        F.monoRight = exR(right, cv::Mat(), F.mvKeysRight, F.mDescriptorsRight, lapR);
        F.Nleft = (int)F.mvKeys.size(); F.Nright = (int)F.mvKeysRight.size(); F.N = F.Nleft + F.Nright;
        KannalaBrandt8 camL(std::vector<float>{190.978f, 190.973f, 254.932f, 256.897f, 0.0034823894f, 0.0007150348f, -0.0020532361f, 0.00020293673f});
        KannalaBrandt8 camR(std::vector<float>{190.442f, 190.434f, 252.597f, 254.917f, 0.0034003171f, 0.0017669277f, -0.0026631445f, 0.00032994600f});
        F.mpCamera = &camL; F.mpCamera2 = &camR;
        F.mRlr = cv::Mat::eye(3, 3, CV_32F);
        F.mRlr.at<float>(0, 2) = 0.003f; F.mRlr.at<float>(2, 0) = -0.003f;
        F.mtlr = cv::Mat(3, 1, CV_32F);
        F.mtlr.at<float>(0) = 0.101f; F.mtlr.at<float>(1) = 0.0004f; F.mtlr.at<float>(2) = -0.0011f;
        F.mvLevelSigma2 = exL.GetScaleSigmaSquares();
        CHECK(F.monoLeft > 0 && F.monoLeft < F.Nleft && F.monoRight > 0 && F.monoRight < F.Nright);
        F.ComputeStereoFishEyeMatches();
        float rig28[28];
        for (int i = 0; i < 8; i++) { rig28[i] = camL.mvParameters[i]; rig28[8 + i] = camR.mvParameters[i]; }
        for (int i = 0; i < 9; i++) rig28[16 + i] = F.mRlr.at<float>(i / 3, i % 3);
        for (int i = 0; i < 3; i++) rig28[25 + i] = F.mtlr.at<float>(i);
        std::vector<int32_t> ol2r(F.Nleft), or2l(F.Nright);
        std::vector<float> odep(F.Nleft), op3((size_t)F.Nleft * 3);
        const int onm = ofr_stereo_fisheye(F.mvKeys.data(), F.mDescriptors.data, F.Nleft, F.monoLeft, F.mvKeysRight.data(), F.mDescriptorsRight.data, F.Nright, F.monoRight,
                                           rig28, F.mvLevelSigma2.data(), ol2r.data(), or2l.data(), odep.data(), op3.data());
        CHECK((int)F.mvLeftToRightMatch.size() == F.Nleft && (int)F.mvRightToLeftMatch.size() == F.Nright && (int)F.mvDepth.size() == F.Nleft &&
              (int)F.mvuRight.size() == F.Nleft && (int)F.mvStereo3Dpoints.size() == F.Nleft && F.mnCloseMPs == 0);
        int nm = 0;
        for (int i = 0; i < F.Nleft; i++) {
            CHECK(F.mvLeftToRightMatch[i] == ol2r[i] && F.mvuRight[i] == -1.f);
            if (ol2r[i] >= 0) {
                nm++;
                CHECK(std::fabs(F.mvDepth[i] - odep[i]) <= 2e-6f * std::fabs(odep[i]));   // device libm vs host libm in the triangulation (bar of test_frame_parity)
                CHECK(!F.mvStereo3Dpoints[i].empty());
                for (int c = 0; c < 3; c++) CHECK(std::fabs(F.mvStereo3Dpoints[i].at<float>(c) - op3[(size_t)i * 3 + c]) <= 2e-5f * (1.f + std::fabs(op3[(size_t)i * 3 + c])));
            } else {
                CHECK(F.mvDepth[i] == -1.f && F.mvStereo3Dpoints[i].empty());
            }
        }
        for (int i = 0; i < F.Nright; i++) CHECK(F.mvRightToLeftMatch[i] == or2l[i]);
        for (int i = 0; i < F.monoLeft; i++) CHECK(F.mvLeftToRightMatch[i] == -1);   // outside the lapping area: never matched
        CHECK(nm == onm && nm > 20);
        std::printf("frame glue ComputeStereoFishEyeMatches: %d matches of %d / %d lapping-area key points\n", nm, F.Nleft - F.monoLeft, F.Nright - F.monoRight);
    }
    std::printf("frame_glue_test OK\n");
    return 0;
}
