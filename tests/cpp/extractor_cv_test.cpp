// Boundary check (GPU tier; CPU tier on the emulated library): orbslam3_hip::ORBextractor compiled WITH its -DORBHIP_WITH_OPENCV branch — the
// reference's own operator() signature (include/ORBextractor.h:57-59) — against the mock cv:: declarations of tests/cpp/mock_orbslam3, and run:
// keypoints / descriptors through operator() must be the bytes the POD form extract() returns, mvImagePyramid must hold every level.
#include <cstdio>
#include <cstring>
#include <vector>

#include "orbslam3_hip/ORBextractor.h"

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    const int W = 320, H = 240;
    cv::Mat img(H, W, CV_8UC1);
    unsigned s = 12345u;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {   // blocks of random grey levels: corners at every block junction
            s = s * 1664525u + 1013904223u;
            const unsigned b = ((x / 20) * 7919u + (y / 20) * 104729u) * 2654435761u;
            img.data[(size_t)y * W + x] = (unsigned char)(40 + (b >> 24) % 170 + ((s >> 28) & 3));
        }
    orbslam3_hip::ORBextractor ex(300, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
    std::vector<int> lap = {0, 0};
    const int mono = ex(img, cv::Mat(), kps, desc, lap);
    CHECK(mono >= 0 && kps.size() > 50 && desc.rows == (int)kps.size() && desc.cols == 32 && desc.type() == CV_8U);
    std::vector<orb_keypoint> k2;
    std::vector<uint8_t> d2;
    const int mono2 = ex.extract(img.data, W, H, (int)img.step, k2, d2, lap);
    CHECK(mono2 == mono && k2.size() == kps.size());
    CHECK(std::memcmp(k2.data(), kps.data(), k2.size() * sizeof(orb_keypoint)) == 0);
    CHECK(std::memcmp(d2.data(), desc.data, d2.size()) == 0);
    CHECK((int)ex.mvImagePyramid.size() == 8 && ex.mvImagePyramid[0].cols == W && ex.mvImagePyramid[0].rows == H);
    CHECK(std::memcmp(ex.mvImagePyramid[0].data, img.data, (size_t)W * H) == 0);   // level 0 is the image itself
    int w7 = 0, h7 = 0;
    const std::vector<uint8_t> l7 = ex.pyramidLevel(7, 0, w7, h7);
    CHECK(ex.mvImagePyramid[7].cols == w7 && ex.mvImagePyramid[7].rows == h7 && std::memcmp(ex.mvImagePyramid[7].data, l7.data(), l7.size()) == 0);
    cv::Mat empty;
    CHECK(ex(empty, cv::Mat(), kps, desc, lap) == -1);   // ORBextractor.cc:1078-1079
    // a failing call must not throw (Frame::ExtractORB runs on bare std::threads): an image too small for eight levels makes orbx_create fail inside
    // operator() — reported through GlueGuard, -1 and empty outputs returned, and the object keeps working afterwards
    {
        const unsigned long f0 = orbslam3_hip::glue_failures();
        cv::Mat tiny(30, 40, CV_8UC1);
        CHECK(ex(tiny, cv::Mat(), kps, desc, lap) == -1 && kps.empty() && desc.empty());
        CHECK(orbslam3_hip::glue_failures() == f0 + 1);
        const orbslam3_hip::GlueLastError le = orbslam3_hip::glue_last_error();
        CHECK(std::strcmp(le.function, "ORBextractor::operator()") == 0 && std::strstr(le.what, "orbx_create") != nullptr);
        CHECK(ex(img, cv::Mat(), kps, desc, lap) == mono && kps.size() == k2.size());
        CHECK(std::memcmp(d2.data(), desc.data, d2.size()) == 0);
    }
    std::printf("extractor_cv_test OK: %zu keypoints through operator()(cv::InputArray, ...)\n", k2.size());
    return 0;
}
