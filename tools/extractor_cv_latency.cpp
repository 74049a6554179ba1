// Runs ON THE GPU BOX: latency of the drop-in call the reference actually makes — orbslam3_hip::ORBextractor::operator()(cv::InputArray, cv::InputArray,
// vector<cv::KeyPoint>&, cv::OutputArray, vector<int>&) (reference include/ORBextractor.h:57-59, called by Frame::ExtractORB, src/Frame.cc:488-495) —
// with mvImagePyramid produced on the host (the default without integration/Frame_hip.cc) and without it, next to the flattened extract() and to
// Frame::ComputeStereoMatches through the adapter (device-resident inputs vs the upload path).  cv:: is the mock of tests/cpp/mock_orbslam3 (OpenCV
// is absent from this image): cv::Mat::create allocates like the real one, cv::KeyPoint has the real layout.
//   g++ -std=c++17 -O2 -DORBHIP_WITH_OPENCV -I tests/cpp/mock_orbslam3 -I include tools/extractor_cv_latency.cpp -L <libdir> -lorbhip -Wl,-rpath,<libdir> -lpthread -o <exe>
//   <exe> [raw 8-bit image file, W, H [, nFeatures [, right image file]]]      (no arguments: a synthetic 752 x 480 scene)
// Prints one JSON object.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "orbslam3_hip/ORBextractor.h"

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static cv::Mat synth(int W, int H, unsigned seed, int shift) {
    cv::Mat img(H, W, CV_8UC1);
    std::memset(img.data, 110, (size_t)W * H);
    unsigned st = seed;
    auto r = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
    for (int k = 0; k < 700; k++) {
        const int cx = (int)(r() % W) - shift, cy = r() % H, hw = 3 + r() % 24, hh = 3 + r() % 24, g = 20 + r() % 215;
        for (int y = cy - hh; y <= cy + hh; y++)
            for (int x = cx - hw; x <= cx + hw; x++)
                if (x >= 0 && x < W && y >= 0 && y < H) img.data[(size_t)y * W + x] = (unsigned char)g;
    }
    return img;
}

static bool load(const char* path, cv::Mat& m) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    const size_t n = std::fread(m.data, 1, (size_t)m.rows * m.cols, f);
    std::fclose(f);
    return n == (size_t)m.rows * m.cols;
}

template <class F> static void timeit(F&& fn, int reps, double& med, double& mean) {
    std::vector<double> t(reps);
    for (int i = 0; i < reps; i++) { const double t0 = now_us(); fn(); t[i] = now_us() - t0; }
    mean = 0;
    for (double v : t) mean += v / reps;
    std::sort(t.begin(), t.end());
    med = t[reps / 2];
}

int main(int argc, char** argv) {
    int W = 752, H = 480, NF = 1000;
    if (argc >= 4) { W = std::atoi(argv[2]); H = std::atoi(argv[3]); }
    if (argc >= 5) NF = std::atoi(argv[4]);
    cv::Mat left = synth(W, H, 99u, 0), right = synth(W, H, 99u, 9);
    if (argc >= 4 && !load(argv[1], left)) { std::printf("{\"error\": \"cannot read %s\"}\n", argv[1]); return 1; }
    if (argc >= 6 && !load(argv[5], right)) { std::printf("{\"error\": \"cannot read %s\"}\n", argv[5]); return 1; }
    const int reps = 400, warm = 30;
    orbslam3_hip::ORBextractor exL(NF, 1.2f, 8, 20, 7), exR(NF, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> kL, kR, k0;
    cv::Mat dL, dR, d0;
    std::vector<int> lap = {0, 0};
    double medPyr, meanPyr, medNo, meanNo, medPod, meanPod, medSt, meanSt, medUp, meanUp;
    // 1. operator() with mvImagePyramid on the host (19-px bordered levels, the reference's layout)
    exL.setKeepHostPyramid(true);
    for (int i = 0; i < warm; i++) exL(left, cv::Mat(), kL, dL, lap);
    timeit([&]() { exL(left, cv::Mat(), kL, dL, lap); }, reps, medPyr, meanPyr);
    const size_t nlev = exL.mvImagePyramid.size();
    size_t pyrBytes = 0;
    for (const cv::Mat& m : exL.mvImagePyramid) pyrBytes += (size_t)(m.rows + 38) * (m.cols + 38);
    k0 = kL; d0 = dL;
    // 2. operator() without it (integration/Frame_hip.cc linked: its only reader runs on the device)
    exL.setKeepHostPyramid(false);
    for (int i = 0; i < warm; i++) exL(left, cv::Mat(), kL, dL, lap);
    timeit([&]() { exL(left, cv::Mat(), kL, dL, lap); }, reps, medNo, meanNo);
    const bool same = kL.size() == k0.size() && !kL.empty() && std::memcmp(kL.data(), k0.data(), kL.size() * sizeof(cv::KeyPoint)) == 0 &&
                      std::memcmp(dL.data, d0.data, kL.size() * 32) == 0;
    // 3. the flattened form (std::vector outputs)
    std::vector<orb_keypoint> pk;
    std::vector<uint8_t> pd;
    timeit([&]() { exL.extract(left.data, W, H, W, pk, pd, lap); }, reps, medPod, meanPod);
    // 4. Frame::ComputeStereoMatches through the adapter after the pair's two calls: inputs still on the device vs uploaded
    exR.setKeepHostPyramid(false);
    exL(left, cv::Mat(), kL, dL, lap);
    exR(right, cv::Mat(), kR, dR, lap);
    std::vector<float> ur, dp, ur2, dp2;
    const float bf = 47.90639384423901f, fx = 458.654f;
    auto stereo = [&](std::vector<float>& u, std::vector<float>& d) {
        exL.ComputeStereoMatches(exR, reinterpret_cast<const orb_keypoint*>(kL.data()), dL.data, (int)kL.size(), reinterpret_cast<const orb_keypoint*>(kR.data()), dR.data,
                                 (int)kR.size(), bf / fx, bf, u, d);
    };
    for (int i = 0; i < warm; i++) stereo(ur, dp);
    timeit([&]() { stereo(ur, dp); }, reps, medSt, meanSt);
    std::vector<cv::KeyPoint> kR2 = kR;
    if (!kR2.empty()) kR2[0].class_id = 7;   // not the last call's bytes any more: the upload path (class_id is not read by the search)
    auto stereoUp = [&](std::vector<float>& u, std::vector<float>& d) {
        exL.ComputeStereoMatches(exR, reinterpret_cast<const orb_keypoint*>(kL.data()), dL.data, (int)kL.size(), reinterpret_cast<const orb_keypoint*>(kR2.data()), dR.data,
                                 (int)kR2.size(), bf / fx, bf, u, d);
    };
    for (int i = 0; i < warm; i++) stereoUp(ur2, dp2);
    timeit([&]() { stereoUp(ur2, dp2); }, reps, medUp, meanUp);
    int nst = 0;
    for (float v : ur) nst += v >= 0;
    const bool stereoSame = ur.size() == ur2.size() && !ur.empty() && std::memcmp(ur.data(), ur2.data(), ur.size() * 4) == 0 && std::memcmp(dp.data(), dp2.data(), dp.size() * 4) == 0;
    std::printf("{\"width\": %d, \"height\": %d, \"nfeatures\": %d, \"keypoints\": %zu, \"reps\": %d, "
                "\"operator_call_with_host_pyramid_us\": {\"median\": %.1f, \"mean\": %.1f}, \"host_pyramid_levels\": %zu, \"host_pyramid_bytes\": %zu, "
                "\"operator_call_no_host_pyramid_us\": {\"median\": %.1f, \"mean\": %.1f}, \"extract_pod_us\": {\"median\": %.1f, \"mean\": %.1f}, "
                "\"outputs_identical_with_and_without_pyramid\": %s, "
                "\"compute_stereo_matches_device_resident_us\": {\"median\": %.1f, \"mean\": %.1f}, \"compute_stereo_matches_upload_path_us\": {\"median\": %.1f, \"mean\": %.1f}, "
                "\"stereo_matches\": %d, \"stereo_paths_identical\": %s, \"glue_failures\": %lu}\n",
                W, H, NF, kL.size(), reps, medPyr, meanPyr, nlev, pyrBytes, medNo, meanNo, medPod, meanPod, same ? "true" : "false", medSt, meanSt, medUp, meanUp, nst,
                stereoSame ? "true" : "false", orbslam3_hip::glue_failures());
    return same && stereoSame ? 0 : 2;
}
