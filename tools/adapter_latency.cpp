// Runs ON THE GPU BOX: latency of the C++ adapter's per-frame call  orbslam3_hip::ORBmatcher::SearchByProjection  (one packed upload, grid build,
// search, one packed download, one synchronisation) on a 1000-keypoint frame with 1000 projected map points.  Inputs are random: the figure of
// interest is the call's fixed cost, not the match count.
//   g++ -std=c++17 -O2 -I include tools/adapter_latency.cpp -L <libdir> -lorbhip -Wl,-rpath,<libdir> -o /tmp/adapter_latency && /tmp/adapter_latency
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

#include "orbslam3_hip/ORBmatcher.h"

int main() {
    const int N = 1000, NQ = 1000;
    std::mt19937 rng(7);
    std::vector<orb_keypoint> k(N);
    std::vector<uint8_t> d((size_t)N * 32), qd((size_t)NQ * 32);
    for (auto& b : d) b = (uint8_t)rng();
    for (auto& b : qd) b = (uint8_t)rng();
    for (int i = 0; i < N; i++) {
        k[i].x = 16.f + (float)(rng() % 720); k[i].y = 16.f + (float)(rng() % 448); k[i].size = 31.f; k[i].angle = (float)(rng() % 360);
        k[i].response = 20.f; k[i].octave = (int)(rng() % 8); k[i].class_id = -1;
    }
    std::vector<orbm_query> q(NQ);
    for (int i = 0; i < NQ; i++) {
        q[i] = orbm_query{};
        q[i].u = 16.f + (float)(rng() % 720); q[i].v = 16.f + (float)(rng() % 448); q[i].radius = 15.f;
        q[i].min_level = 0; q[i].max_level = 7; q[i].flags = ORBM_Q_VALID | ORBM_Q_HAS_OBS; q[i].angle = (float)(rng() % 360);
    }
    orbslam3_hip::FrameView F;
    F.N = N; F.keysUn = k.data(); F.descriptors = d.data();
    F.grid = orbm_grid_params{0.f, 0.f, 64.f / 752.f, 48.f / 480.f};
    orbslam3_hip::ORBmatcher m(0.9f, true);
    std::vector<int> km, qm;
    int nm = 0;
    for (int i = 0; i < 20; i++) nm = m.SearchByProjection(F, q, qd, ORBM_MODE_BEST_ONLY, 100, km, qm);
    const int R = 300;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < R; i++) nm = m.SearchByProjection(F, q, qd, ORBM_MODE_BEST_ONLY, 100, km, qm);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
    std::printf("adapter SearchByProjection: %.1f us per call (%d keypoints, %d queries, %d matches)\n", us, N, NQ, nm);
    return 0;
}
