// Threading contract of the boundary (SURVEY.md 8(b) "Threading"; round-4 verdict, item 2).
//
// The reference extracts the left and the right image on two bare std::threads (Frame.cc:111-114, 1209-1212), uses ORBmatcher objects from the
// Tracking, LocalMapping and LoopClosing threads at the same time, and runs Optimizer::LocalBundleAdjustment on the LocalMapping thread
// (LocalMapping.cc:201) while Tracking keeps extracting and matching.  Here, concurrently and `iters` times each:
//   thread 1 / 2   ORBextractor::extract on two extractor objects (left / right image)
//   thread 3       ORBmatcher::SearchByProjection (motion model) + SearchByBoW on its own matcher objects, a failing orbx_create (image too small)
//   thread 4       LbaLinearizer::optimize(5) on a small window (poses / points reset every turn)
//   thread 5       extractor handles of another size created and destroyed, a failing orbx_create (bad configuration) whose message must be this thread's own
//                  (orbx_last_error(NULL) is per thread), PoseOptimizer::optimize
// and every output of every turn is compared, bit for bit (poses: 1e-12), with what the same objects returned single-threaded before the threads
// started.  Built by tests/test_threads.py against the emulated library (CPU tier; also with -fsanitize=thread) and the real liborbhip.so (GPU tier).
//
//   threads_test [W H nfeatures iters lm_iterations with_pose_optimizer]
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "orbslam3_hip/ORBextractor.h"
#include "orbslam3_hip/ORBmatcher.h"
#include "orbslam3_hip/Optimizer.h"

static std::atomic<int> g_fail{0};
#define TCHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); g_fail++; return; } } while (0)

static std::vector<uint8_t> make_image(int W, int H, int dx, int dy, unsigned seed) {
    std::vector<uint8_t> img((size_t)W * H, 110);
    unsigned s = seed;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int r = 0; r < (W * H) / 550; r++) {
        const int cx = (int)(rnd() % W) + dx, cy = (int)(rnd() % H) + dy, hw = 3 + rnd() % 25, hh = 3 + rnd() % 25, g = 20 + rnd() % 215;
        for (int y = cy - hh; y <= cy + hh; y++)
            for (int x = cx - hw; x <= cx + hw; x++)
                if (x >= 0 && x < W && y >= 0 && y < H) img[(size_t)y * W + x] = (uint8_t)g;
    }
    for (size_t i = 0; i < img.size(); i++) img[i] = (uint8_t)std::min(255, std::max(0, (int)img[i] + (int)(rnd() % 5) - 2));
    return img;
}

struct Extracted { int mono = 0; std::vector<orb_keypoint> k; std::vector<uint8_t> d; };
static bool same(const Extracted& a, const Extracted& b) {
    return a.mono == b.mono && a.k.size() == b.k.size() && a.d == b.d && std::memcmp(a.k.data(), b.k.data(), a.k.size() * sizeof(orb_keypoint)) == 0;
}

struct MatchScene {
    orbslam3_hip::FrameView F;
    std::vector<orbm_query> q;
    const std::vector<uint8_t>* qdesc;
    orbslam3_hip::ORBmatcher::KeyFrameView K1, K2;
    std::vector<uint8_t> valid, mp2;
    std::vector<float> angA, angB;
};
struct MatchResult { int nm = 0, nb = 0; std::vector<int> kpMatch, qMatch, fm; };
static void run_match(const MatchScene& S, MatchResult& R) {
    orbslam3_hip::ORBmatcher m(0.9f, true);      // stack-local and stateless in the reference; here: owns its device buffers
    R.nm = m.SearchByProjection(S.F, S.q, *S.qdesc, ORBM_MODE_BEST_ONLY, ORBM_TH_HIGH, R.kpMatch, R.qMatch);
    orbslam3_hip::ORBmatcher mb(0.7f, true);
    R.nb = mb.SearchByBoW(S.K1, S.angA.data(), S.K2, S.angB.data(), -1, R.fm);
}

struct Window {
    orbslam3_hip::LbaLinearizer L;
    std::vector<double> poses, points;
    void build() {
        lba_camera cam{};
        cam.model = LBA_CAM_PINHOLE; cam.p[0] = 458.654f; cam.p[1] = 457.296f; cam.p[2] = 367.215f; cam.p[3] = 248.375f; cam.bf = 47.906f; cam.trl_q[3] = 1;
        L.addCamera(cam);
        unsigned s = 4242u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
        for (int k = 0; k < 5; k++) {
            const float a = 0.04f * k;
            const float Tcw[12] = {std::cos(a), 0, std::sin(a), 0.1f * k, 0, 1, 0, 0.02f * k, -std::sin(a), 0, std::cos(a), 0.03f * k};
            double p7[7];
            orbslam3_hip::LbaLinearizer::poseFromTcw(Tcw, 4, p7);
            L.addPose(p7, k == 0);
            poses.insert(poses.end(), p7, p7 + 7);
        }
        for (int l = 0; l < 60; l++) {
            const double X[3] = {((int)(rnd() % 400) - 200) / 100.0, ((int)(rnd() % 300) - 150) / 100.0, 4.0 + (rnd() % 300) / 100.0};
            L.addPoint(X);
            points.insert(points.end(), X, X + 3);
            for (int k = 0; k < 5; k++) {
                if ((l + k) % 4 == 0) continue;
                const int kind = (l % 3 == 0) ? LBA_EDGE_STEREO : LBA_EDGE_MONO;
                const float u = 367.f + 80.f * (float)X[0] + (float)(rnd() % 7) - 3.f, v = 248.f + 80.f * (float)X[1] + (float)(rnd() % 7) - 3.f;
                L.addEdge(k, l, kind, 0, u, v, u - 9.f, 1.0f / (1.44f * (1 + k % 3)));
            }
        }
    }
    // one LocalBundleAdjustment turn from the initial state -> (iterations, final chi2, poses)
    int lmIts = 5;
    void run(int& its, double& chi, std::vector<double>& out) {
        for (int k = 0; k < 5; k++) L.setPose(k, &poses[7 * k]);
        for (int l = 0; l < 60; l++) L.setPoint(l, &points[3 * l]);
        its = L.optimize(lmIts, nullptr, &chi);
        out.assign(L.pose(0), L.pose(0) + 35);
    }
};

struct PoseScene {
    orbslam3_hip::PoseOptimizer PO;
    double p0[7];
    void build() {
        lba_camera cam{};
        cam.model = LBA_CAM_PINHOLE; cam.p[0] = 458.654f; cam.p[1] = 457.296f; cam.p[2] = 367.215f; cam.p[3] = 248.375f; cam.bf = 47.906f; cam.trl_q[3] = 1;
        PO.addCamera(cam);
        unsigned s = 99u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
        for (int l = 0; l < 100; l++) {
            const float X[3] = {((int)(rnd() % 400) - 200) / 100.0f, ((int)(rnd() % 300) - 150) / 100.0f, 4.0f + (rnd() % 300) / 100.0f};
            float u = 367.215f + 458.654f * X[0] / X[2] + ((int)(rnd() % 200) - 100) / 100.0f, v = 248.375f + 457.296f * X[1] / X[2] + ((int)(rnd() % 200) - 100) / 100.0f;
            if (l % 11 == 0) { u += 35.f; v -= 20.f; }
            const float s2 = 1.0f / (1.44f * (1 + l % 3));
            if (l % 2) PO.addStereo(X, u, v, u - 47.906f / X[2], s2); else PO.addMono(X, u, v, s2);
        }
        const float Tcw0[12] = {std::cos(0.01f), 0, std::sin(0.01f), 0.03f, 0, 1, 0, -0.02f, -std::sin(0.01f), 0, std::cos(0.01f), 0.04f};
        orbslam3_hip::LbaLinearizer::poseFromTcw(Tcw0, 4, p0);
    }
    int run(double out[7], std::vector<bool>& outl) { std::memcpy(out, p0, sizeof(p0)); return PO.optimize(out, outl); }
};

int main(int argc, char** argv) {
    const int W = argc > 1 ? std::atoi(argv[1]) : 640, H = argc > 2 ? std::atoi(argv[2]) : 480, NF = argc > 3 ? std::atoi(argv[3]) : 1000;
    const int iters = argc > 4 ? std::atoi(argv[4]) : 200;
    const std::vector<uint8_t> imgL = make_image(W, H, 0, 0, 777u), imgR = make_image(W, H, 5, -3, 777u);
    const std::vector<int> lap = {0, 1000};

    // ---- single-threaded references, from the very objects the threads will use
    orbslam3_hip::ORBextractor exL(NF, 1.2f, 8, 20, 7), exR(NF, 1.2f, 8, 20, 7);
    Extracted refL, refR;
    refL.mono = exL.extract(imgL.data(), W, H, W, refL.k, refL.d, lap);
    refR.mono = exR.extract(imgR.data(), W, H, W, refR.k, refR.d, lap);
    if (refL.k.size() < 60 || refR.k.size() < 60) { std::printf("FAIL: too few key points (%zu, %zu)\n", refL.k.size(), refR.k.size()); return 1; }

    MatchScene MS;
    const std::vector<float> sf = exL.GetScaleFactors();
    MS.F.N = (int)refR.k.size(); MS.F.keysUn = refR.k.data(); MS.F.descriptors = refR.d.data();
    MS.F.grid = orbm_grid_params{0.f, 0.f, 64.f / W, 48.f / H};
    MS.q.resize(refL.k.size());
    for (size_t i = 0; i < refL.k.size(); i++) {
        orbm_query& q = MS.q[i];
        std::memset(&q, 0, sizeof(q));
        q.u = refL.k[i].x + 5.f; q.v = refL.k[i].y - 3.f; q.radius = 15.f * sf[refL.k[i].octave]; q.angle = refL.k[i].angle;
        q.min_level = (int16_t)(refL.k[i].octave - 1); q.max_level = (int16_t)(refL.k[i].octave + 1); q.flags = ORBM_Q_VALID | ORBM_Q_HAS_OBS;
    }
    MS.qdesc = &refL.d;
    MS.valid.assign(refL.k.size(), 1); MS.mp2.assign(refR.k.size(), 0);
    auto csr = [](orbslam3_hip::ORBmatcher::KeyFrameView& K, const Extracted& e, const std::vector<uint8_t>& mp) {
        K.N = (int)e.k.size(); K.keysUn = e.k.data(); K.descriptors = e.d.data(); K.hasMapPoint = mp.data();
        std::vector<std::vector<int32_t>> nodes(40);
        for (int i = 0; i < K.N; i++) nodes[((e.d[(size_t)i * 32] >> 4) * 7 + (e.d[(size_t)i * 32 + 9] >> 5) * 3) % 40].push_back(i);
        K.nodeStart.push_back(0);
        for (int n = 0; n < 40; n++)
            if (!nodes[n].empty()) { K.nodeId.push_back(n); K.featIdx.insert(K.featIdx.end(), nodes[n].begin(), nodes[n].end()); K.nodeStart.push_back((int32_t)K.featIdx.size()); }
    };
    csr(MS.K1, refL, MS.valid); csr(MS.K2, refR, MS.mp2);
    for (const orb_keypoint& k : refL.k) MS.angA.push_back(k.angle);
    for (const orb_keypoint& k : refR.k) MS.angB.push_back(k.angle);
    MatchResult refM;
    run_match(MS, refM);
    if (refM.nm < 20 || refM.nb < 5) { std::printf("FAIL: reference match counts %d / %d\n", refM.nm, refM.nb); return 1; }

    Window Wn;
    Wn.lmIts = argc > 5 ? std::atoi(argv[5]) : 5;     // (the ThreadSanitizer run of the emulated library pays ~30 s per LM iteration)
    Wn.build();
    int refIts = 0; double refChi = 0; std::vector<double> refPoses;
    Wn.run(refIts, refChi, refPoses);
    const bool withPose = argc > 6 ? std::atoi(argv[6]) != 0 : true;   // (40 LM iterations in one launch: 15 s per run under ThreadSanitizer)
    PoseScene PS;
    PS.build();
    double refPose[7] = {0, 0, 0, 0, 0, 0, 0}; std::vector<bool> refOutl;
    const int refGood = withPose ? PS.run(refPose, refOutl) : 100;
    if (refIts < 1 || refGood < 60) { std::printf("FAIL: reference optimisations (%d iterations, %d inliers)\n", refIts, refGood); return 1; }

    // ---- the same, concurrently
    auto extract_loop = [&](orbslam3_hip::ORBextractor& ex, const std::vector<uint8_t>& img, const Extracted& ref) {
        for (int it = 0; it < iters; it++) {
            Extracted e;
            e.mono = ex.extract(img.data(), W, H, W, e.k, e.d, lap);
            TCHECK(same(e, ref));
        }
    };
    std::thread t1([&] { extract_loop(exL, imgL, refL); });
    std::thread t2([&] { extract_loop(exR, imgR, refR); });
    std::thread t3([&] {
        for (int it = 0; it < iters; it++) {
            MatchResult r;
            run_match(MS, r);
            TCHECK(r.nm == refM.nm && r.nb == refM.nb && r.kpMatch == refM.kpMatch && r.qMatch == refM.qMatch && r.fm == refM.fm);
            orbx_config big{NF, 1.2f, 8, 20, 7};          // a failing orbx_create here AND on thread 5: each must read back its own message
            orbx_handle hb = nullptr;
            TCHECK(orbx_create(&big, 40, 30, 1, 0, &hb) != ORB_OK && hb == nullptr);      // too small for 8 levels
            TCHECK(std::string(orbx_last_error(nullptr)).find("image too small") == 0);
        }
    });
    std::thread t4([&] {
        for (int it = 0; it < iters; it++) {
            int its = 0; double chi = 0; std::vector<double> p;
            Wn.run(its, chi, p);
            TCHECK(its == refIts && std::fabs(chi - refChi) <= 1e-9 * std::fabs(refChi) && p.size() == refPoses.size());
            for (size_t i = 0; i < p.size(); i++) TCHECK(std::fabs(p[i] - refPoses[i]) <= 1e-12);
        }
    });
    std::thread t5([&] {
        const int W2 = W + 16, H2 = H + 8;
        const std::vector<uint8_t> img2 = make_image(W2, H2, 0, 0, 31u);
        Extracted first;
        for (int it = 0; it < iters; it++) {
            {   // a new extractor object: tables, a handle of another size, constant-table upload, destruction — next to live handles
                orbslam3_hip::ORBextractor ex2(NF / 2, 1.2f, 8, 20, 7);
                Extracted e;
                e.mono = ex2.extract(img2.data(), W2, H2, W2, e.k, e.d, lap);
                if (it == 0) first = e;
                TCHECK(same(e, first) && e.k.size() > 20);
            }
            orbx_config bad{NF, 1.2f, 0, 20, 7};          // nlevels = 0
            orbx_handle hb = nullptr;
            TCHECK(orbx_create(&bad, W, H, 1, 0, &hb) != ORB_OK && hb == nullptr);
            TCHECK(std::string(orbx_last_error(nullptr)) == "bad configuration");
            if (withPose) {
                double p[7]; std::vector<bool> outl;
                const int good = PS.run(p, outl);
                TCHECK(good == refGood && outl == refOutl && std::memcmp(p, refPose, sizeof(p)) == 0);
            }
        }
    });
    t1.join(); t2.join(); t3.join(); t4.join(); t5.join();
    if (g_fail.load()) { std::printf("threads_test FAILED (%d checks)\n", g_fail.load()); return 1; }
    std::printf("threads_test OK: 5 threads x %d turns at %dx%d: 2 x extract (%zu / %zu key points), SearchByProjection (%d) + SearchByBoW (%d), "
                "LbaLinearizer::optimize (%d iterations), create / destroy / failing create + PoseOptimizer (%d inliers) — all equal to the single-threaded results\n",
                iters, W, H, refL.k.size(), refR.k.size(), refM.nm, refM.nb, refIts, refGood);
    return 0;
}
