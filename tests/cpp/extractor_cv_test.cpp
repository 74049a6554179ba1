// Boundary check (GPU tier; CPU tier on the emulated library): orbslam3_hip::ORBextractor compiled WITH its -DORBHIP_WITH_OPENCV branch — the
// reference's own operator() signature (include/ORBextractor.h:57-59) — against the mock cv:: declarations of tests/cpp/mock_orbslam3, and run:
// keypoints / descriptors through operator() must be the bytes the POD form extract() returns, mvImagePyramid must hold every level.
#include <cstddef>
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
    // mvImagePyramid (on by default without integration/Frame_hip.cc): headers over the handle's pinned slab, every level inside its 19-px
    // BORDER_REFLECT_101 frame (ORBextractor.cc:1164-1179)
    CHECK(ex.keepHostPyramid());
    CHECK((int)ex.mvImagePyramid.size() == 8 && ex.mvImagePyramid[0].cols == W && ex.mvImagePyramid[0].rows == H);
    for (int y = 0; y < H; y++) CHECK(std::memcmp(ex.mvImagePyramid[0].ptr(y), img.data + (size_t)y * W, (size_t)W) == 0);   // level 0 is the image itself
    {
        auto refl = [](int p, int len) { if (p < 0) p = -p; if (p >= len) p = 2 * (len - 1) - p; return p; };
        const cv::Mat& L0 = ex.mvImagePyramid[0];
        for (int y = -19; y < H + 19; y++)
            for (int x = -19; x < W + 19; x++)
                CHECK(L0.data[(ptrdiff_t)y * (ptrdiff_t)L0.step + x] == img.data[(size_t)refl(y, H) * W + refl(x, W)]);
    }
    const unsigned char* slab0 = ex.mvImagePyramid[0].data;
    // the same levels without the host option: the device plane + the host-side reflect of orbx_copy_level must give the slab's bytes
    std::vector<std::vector<uint8_t>> slabLevels(8);
    std::vector<size_t> slabStep(8);
    for (int l = 0; l < 8; l++) {
        const cv::Mat& M = ex.mvImagePyramid[l];
        slabStep[l] = M.step;
        slabLevels[l].assign(M.data - 19 * (ptrdiff_t)M.step - 19, M.data - 19 * (ptrdiff_t)M.step - 19 + (size_t)(M.rows + 38) * M.step);
    }
    ex.setKeepHostPyramid(false);
    CHECK(ex(img, cv::Mat(), kps, desc, lap) == mono && ex.mvImagePyramid.empty());   // nothing stale is left behind
    for (int l = 0; l < 8; l++) {
        int w = 0, h = 0;
        const std::vector<uint8_t> bl = ex.pyramidLevel(l, 19, w, h);
        for (int y = 0; y < h + 38; y++) CHECK(std::memcmp(bl.data() + (size_t)y * (w + 38), slabLevels[l].data() + (size_t)y * slabStep[l], (size_t)(w + 38)) == 0);
        const std::vector<uint8_t> pl = ex.pyramidLevel(l, 0, w, h);
        for (int y = 0; y < h; y++) CHECK(std::memcmp(pl.data() + (size_t)y * w, slabLevels[l].data() + (size_t)(y + 19) * slabStep[l] + 19, (size_t)w) == 0);
    }
    ex.setKeepHostPyramid(true);
    CHECK(ex(img, cv::Mat(), kps, desc, lap) == mono && (int)ex.mvImagePyramid.size() == 8);
    CHECK(ex.mvImagePyramid[0].data == slab0);   // persistent storage: no allocation per call
    {
        int w = 0, h = 0;
        const std::vector<uint8_t> b3 = ex.pyramidLevel(3, 19, w, h);   // (served from the slab now)
        const cv::Mat& M = ex.mvImagePyramid[3];
        CHECK(M.cols == w && M.rows == h);
        for (int y = 0; y < h + 38; y++) CHECK(std::memcmp(b3.data() + (size_t)y * (w + 38), M.data + ((ptrdiff_t)y - 19) * (ptrdiff_t)M.step - 19, (size_t)(w + 38)) == 0);
    }
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
