// Compile-and-run check of the reference-signature glue (integration/*.cc) against the mock declarations in tests/cpp/mock_orbslam3:
// every function is driven through the reference's own signature on mock Frame / KeyFrame / MapPoint objects and compared with the oracle's
// literal restatement of the reference loop on independently flattened inputs (matcher) or with the flattened LbaLinearizer path (LBA).
// Built by tests/test_glue.py against the emulated library (CPU tier) or the real liborbhip.so (GPU tier).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#include "ORBmatcher.h"
#include "Optimizer.h"
#include "G2oTypes.h"
#include <orbslam3_hip/Optimizer.h>

extern "C" {
void* oro_create(int, float, int, int, int);
void oro_destroy(void*);
int oro_extract(void*, const uint8_t*, int, int, int, int, int, void*, uint8_t*, int, int*);
int omo_search_by_projection(const void*, const uint8_t*, const float*, const uint8_t*, int, float, float, float, float, const void*,
                             const uint8_t*, int, int, int, float, int, int32_t*, int32_t*);
int omo_search_by_bow(const uint8_t*, const float*, const uint8_t*, const int32_t*, const int32_t*, const int32_t*, int, const uint8_t*, const float*, int,
                      const int32_t*, const int32_t*, const int32_t*, int, float, int, int32_t*, int);
int omo_search_for_initialization(const void* kps1, const uint8_t* desc1, int n1, const void* kps2, const uint8_t* desc2, int n2, float minX, float minY,
                                  float gwInv, float ghInv, float* prev, int windowSize, float nnratio, int checkOri, int32_t* matches12);
int omo_search_by_sim3(const void* kps1, const uint8_t* desc1, int n1, float minX1, float minY1, float gwInv1, float ghInv1, const void* kps2, const uint8_t* desc2,
                       int n2, float minX2, float minY2, float gwInv2, float ghInv2, const void* q12, const uint8_t* q12desc, const void* q21, const uint8_t* q21desc,
                       int32_t* matches12);
int opo_pose_optimize(const double* pose_in, const void* edges, int n_edges, const void* cams, double* pose_out, uint8_t* outlier);
int omo_search_by_bow_kf(const uint8_t*, const float*, const uint8_t*, const int32_t*, const int32_t*, const int32_t*, int, int, const uint8_t*, const float*,
                         const uint8_t*, const int32_t*, const int32_t*, const int32_t*, int, int, float, int, int32_t*);
}

std::mutex ORB_SLAM3::MapPoint::mGlobalMutex;

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

using namespace ORB_SLAM3;
// the parts of the reference's ORBmatcher.cc / Frame.cc that stay where they are
const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;
float ORBmatcher::RadiusByViewingCos(const float& viewCos) { return viewCos > 0.998 ? 2.5 : 4.0; }   // ORBmatcher.cc:260-266
float Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv;

static uint32_t rng_state = 4242u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static float frand(float a, float b) { return a + (b - a) * (float)(rnd() % 100000) / 100000.f; }

static std::vector<uint8_t> make_image(int W, int H, int dx, int dy) {
    std::vector<uint8_t> img((size_t)W * H, 110);
    uint32_t st = 777u;
    auto r = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
    for (int k = 0; k < 220; k++) {
        const int cx = r() % W + dx, cy = r() % H + dy, hw = 3 + r() % 25, hh = 3 + r() % 25, g = 20 + r() % 215;
        for (int y = cy - hh; y <= cy + hh; y++)
            for (int x = cx - hw; x <= cx + hw; x++)
                if (x >= 0 && x < W && y >= 0 && y < H) img[(size_t)y * W + x] = (uint8_t)g;
    }
    return img;
}
struct Query { float u, v, radius, uRight, angle; int16_t minLevel, maxLevel; uint32_t flags; };   // == orbm_query == the oracle's Query

static cv::Mat descMat(const std::vector<uint8_t>& d) { cv::Mat m((int)d.size() / 32, 32, CV_8U); std::memcpy(m.data, d.data(), d.size()); return m; }
static DBoW2::FeatureVector featVec(const std::vector<uint8_t>& d, int n, int dropMod) {
    DBoW2::FeatureVector fv;
    for (int i = 0; i < n; i++) {
        const unsigned node = ((d[(size_t)i * 32] >> 4) * 7 + (d[(size_t)i * 32 + 9] >> 5) * 3) % 40;
        if (dropMod && node % dropMod == 1) continue;
        fv[node].push_back((unsigned)i);
    }
    return fv;
}
static void csr(const DBoW2::FeatureVector& fv, std::vector<int32_t>& id, std::vector<int32_t>& st, std::vector<int32_t>& fe) {
    st.push_back(0);
    for (auto& kv : fv) { id.push_back((int32_t)kv.first); for (unsigned f : kv.second) fe.push_back((int32_t)f); st.push_back((int32_t)fe.size()); }
}

int main() {
    const int W = 480, H = 360, NF = 600;
    const float fx = 400.f, fy = 400.f, cx = 240.f, cy = 180.f;
    std::vector<float> sf(8), invSig2(8);
    sf[0] = 1.f; for (int i = 1; i < 8; i++) sf[i] = sf[i - 1] * 1.2f;
    for (int i = 0; i < 8; i++) invSig2[i] = 1.f / (sf[i] * sf[i]);
    // two views of one scene extracted by the oracle: A (the "last frame" / key frame) and B = A moved by (5,-3) (the current frame)
    std::vector<cv::KeyPoint> kA(4 * NF), kB(4 * NF);
    std::vector<uint8_t> dA((size_t)4 * NF * 32), dB((size_t)4 * NF * 32);
    int nA = 0, nB = 0;
    {
        void* o = oro_create(NF, 1.2f, 8, 20, 7);
        auto a = make_image(W, H, 0, 0), b = make_image(W, H, 5, -3);
        oro_extract(o, a.data(), W, H, W, 0, 0, kA.data(), dA.data(), 4 * NF, &nA);
        oro_extract(o, b.data(), W, H, W, 0, 0, kB.data(), dB.data(), 4 * NF, &nB);
        oro_destroy(o);
        kA.resize(nA); kB.resize(nB); dA.resize((size_t)nA * 32); dB.resize((size_t)nB * 32);
    }
    CHECK(nA > 400 && nB > 400);
    Frame::mnMinX = 0; Frame::mnMaxX = W; Frame::mnMinY = 0; Frame::mnMaxY = H;
    Frame::mfGridElementWidthInv = 64.f / W; Frame::mfGridElementHeightInv = 48.f / H;
    Pinhole cam(fx, fy, cx, cy);
    auto fresh_frame = [&]() {
        Frame F;
        F.N = nB; F.mvKeys = kB; F.mvKeysUn = kB; F.mvuRight.assign(nB, -1.f); F.mDescriptors = descMat(dB);
        F.mvpMapPoints.assign(nB, (MapPoint*)NULL); F.mvbOutlier.assign(nB, false); F.mvScaleFactors = sf; F.mpCamera = &cam;
        F.mTcw = cv::Mat::eye(4, 4, CV_32F); F.mbf = 40.f; F.mb = 0.1f; F.mfLogScaleFactor = std::log(1.2f);
        for (int i = 0; i < nB; i++) if (i % 3 == 0) F.mvuRight[i] = kB[i].pt.x - frand(2.f, 30.f);
        return F;
    };
    // map points = A's keypoints lifted to 3-D at the current frame's pose (identity), so that they project to A's position + (5,-3) in B
    std::vector<MapPoint> mps(nA);
    std::vector<MapPoint*> vpMPs(nA);
    for (int i = 0; i < nA; i++) {
        MapPoint& p = mps[i];
        const float u = kA[i].pt.x + 5.f, v = kA[i].pt.y - 3.f, z = frand(2.f, 8.f);
        p.mWorldPos = cv::Mat(3, 1, CV_32F);
        p.mWorldPos.at<float>(0) = (u - cx) / fx * z; p.mWorldPos.at<float>(1) = (v - cy) / fy * z; p.mWorldPos.at<float>(2) = z;
        p.mNormalVector = cv::Mat(3, 1, CV_32F);
        const float nrm = (float)cv::norm(p.mWorldPos);
        for (int k = 0; k < 3; k++) p.mNormalVector.at<float>(k) = p.mWorldPos.at<float>(k) / nrm;
        p.mDescriptor = cv::Mat(1, 32, CV_8U); std::memcpy(p.mDescriptor.data, &dA[(size_t)i * 32], 32);
        p.nObs = (i % 11 == 0) ? 0 : 3; p.mnId = i; p.mbBad = (i % 29 == 7);
        p.mfMaxDistance = nrm * sf[kA[i].octave] * 1.05f; p.mfMinDistance = p.mfMaxDistance / sf[7] * 0.5f;
        p.mbTrackInView = (i % 13 != 5); p.mTrackProjX = u; p.mTrackProjY = v; p.mTrackProjXR = u - 40.f / z; p.mTrackDepth = z;
        p.mnTrackScaleLevel = kA[i].octave; p.mTrackViewCos = (i % 3 == 0) ? 0.9990f : 0.95f;
        vpMPs[i] = &p;
    }
    const float grid[4] = {0.f, 0.f, 64.f / W, 48.f / H};
    auto oracle = [&](const Frame& F, const std::vector<uint8_t>& occ, bool useUR, const std::vector<Query>& q, const std::vector<uint8_t>& qd, int mode, int thDist,
                      float ratio, int ori, std::vector<int32_t>& km) {
        std::vector<int32_t> qm(q.size() + 1);
        km.assign(F.N + 1, -1);
        return omo_search_by_projection(F.mvKeysUn.data(), F.mDescriptors.data, useUR ? F.mvuRight.data() : nullptr, occ.data(), F.N, grid[0], grid[1], grid[2], grid[3],
                                        q.data(), qd.data(), (int)q.size(), mode, thDist, ratio, ori, qm.data(), km.data());
    };

    // ---- 1. local-map search (Tracking::SearchLocalPoints) ----
    for (float th : {1.f, 3.f}) {
        Frame F = fresh_frame();
        MapPoint pre; pre.nObs = 2;
        for (int i = 0; i < nB; i += 9) F.mvpMapPoints[i] = &pre;     // keypoints that already hold an observed point
        std::vector<uint8_t> occ(nB, 0);
        for (int i = 0; i < nB; i++) occ[i] = F.mvpMapPoints[i] != NULL;
        std::vector<Query> q; std::vector<uint8_t> qd; std::vector<int> own;
        for (int i = 0; i < nA; i++) {
            MapPoint& p = mps[i];
            if (!p.mbTrackInView || p.mbBad) continue;
            float r = (p.mTrackViewCos > 0.998 ? 2.5f : 4.0f); if (th != 1.f) r *= th;
            q.push_back(Query{p.mTrackProjX, p.mTrackProjY, r * sf[p.mnTrackScaleLevel], p.mTrackProjXR, 0.f, (int16_t)(p.mnTrackScaleLevel - 1), (int16_t)p.mnTrackScaleLevel,
                              1u | 2u | (p.nObs > 0 ? 4u : 0u)});
            qd.insert(qd.end(), &dA[(size_t)i * 32], &dA[(size_t)i * 32] + 32); own.push_back(i);
        }
        std::vector<int32_t> km;
        const int on = oracle(F, occ, true, q, qd, 0, 100, 0.8f, 1, km);
        ORBmatcher m(0.8f, true);
        const int n = m.SearchByProjection(F, vpMPs, th);
        CHECK(n == on && n > 100);
        for (int i = 0; i < nB; i++) CHECK(F.mvpMapPoints[i] == (km[i] >= 0 ? &mps[own[km[i]]] : (occ[i] ? &pre : (MapPoint*)NULL)));
        std::printf("glue local-map th=%.0f: %d matches\n", th, n);
    }
    // ---- 2. motion-model search (Tracking::TrackWithMotionModel): LastFrame = A with its map points ----
    {
        Frame Last;
        Last.N = nA; Last.mvKeys = kA; Last.mvKeysUn = kA; Last.mvpMapPoints = vpMPs; Last.mvbOutlier.assign(nA, false); Last.mTcw = cv::Mat::eye(4, 4, CV_32F);
        for (int i = 0; i < nA; i += 17) Last.mvbOutlier[i] = true;
        for (int i = 0; i < nA; i += 23) Last.mvpMapPoints[i] = NULL;
        for (int mono = 0; mono < 2; mono++) {
            Frame F = fresh_frame();
            std::vector<uint8_t> occ(nB, 0);
            std::vector<Query> q; std::vector<uint8_t> qd; std::vector<int> own;
            for (int i = 0; i < nA; i++) {
                if (!Last.mvpMapPoints[i] || Last.mvbOutlier[i]) continue;
                MapPoint& p = mps[i];
                const float X = p.mWorldPos.at<float>(0), Y = p.mWorldPos.at<float>(1), Z = p.mWorldPos.at<float>(2);
                const float invz = 1.0 / Z;
                const float u = fx * X / Z + cx, v = fy * Y / Z + cy;
                if (u < 0 || u > W || v < 0 || v > H) continue;
                const int o = kA[i].octave;
                q.push_back(Query{u, v, 7.f * sf[o], u - 40.f * invz, kA[i].angle, (int16_t)(o - 1), (int16_t)(o + 1), 1u | 2u | (p.nObs > 0 ? 4u : 0u)});
                qd.insert(qd.end(), &dA[(size_t)i * 32], &dA[(size_t)i * 32] + 32); own.push_back(i);
            }
            std::vector<int32_t> km;
            const int on = oracle(F, occ, true, q, qd, 1, 100, 0.9f, 1, km);
            ORBmatcher m(0.9f, true);
            const int n = m.SearchByProjection(F, Last, 7.f, mono != 0);
            CHECK(n == on && n > 100);
            for (int i = 0; i < nB; i++) CHECK(F.mvpMapPoints[i] == (km[i] >= 0 ? &mps[own[km[i]]] : (MapPoint*)NULL));
            std::printf("glue motion-model bMono=%d: %d matches\n", mono, n);
        }
    }
    // key frame A (for the relocalisation, Sim3 and BoW searches)
    KeyFrame KFa(fx, fy, cx, cy, 40.f, -1, 0, 0, W, H, 64.f / W, 48.f / H, sf, invSig2);
    KFa.N = nA; KFa.mvKeys = kA; KFa.mvKeysUn = kA; KFa.mvuRight.assign(nA, -1.f); KFa.mDescriptors = descMat(dA); KFa.mvpMapPoints = vpMPs;
    KFa.mpCamera = &cam; KFa.mfLogScaleFactor = std::log(1.2f); KFa.mFeatVec = featVec(dA, nA, 0);
    for (int i = 0; i < nA; i += 19) KFa.mvpMapPoints[i] = NULL;
    // ---- 3. relocalisation search ----
    {
        Frame F = fresh_frame();
        MapPoint pre; pre.nObs = 0;
        for (int i = 0; i < nB; i += 7) F.mvpMapPoints[i] = &pre;     // already-found points block regardless of their observations (:2586)
        std::set<MapPoint*> found;
        for (int i = 0; i < nA; i += 5) found.insert(&mps[i]);
        std::vector<uint8_t> occ(nB);
        for (int i = 0; i < nB; i++) occ[i] = F.mvpMapPoints[i] != NULL;
        std::vector<Query> q; std::vector<uint8_t> qd; std::vector<int> own;
        for (int i = 0; i < nA; i++) {
            MapPoint* p = KFa.mvpMapPoints[i];
            if (!p || p->mbBad || found.count(p)) continue;
            const float X = p->mWorldPos.at<float>(0), Y = p->mWorldPos.at<float>(1), Z = p->mWorldPos.at<float>(2);
            const float u = fx * X / Z + cx, v = fy * Y / Z + cy;
            if (u < 0 || u > W || v < 0 || v > H) continue;
            const float dist3D = (float)cv::norm(p->mWorldPos);
            if (dist3D < p->GetMinDistanceInvariance() || dist3D > p->GetMaxDistanceInvariance()) continue;
            const int L = p->PredictScale(dist3D, &F);
            q.push_back(Query{u, v, 10.f * sf[L], 0.f, kA[i].angle, (int16_t)(L - 1), (int16_t)(L + 1), 1u | 4u});
            qd.insert(qd.end(), &dA[(size_t)i * 32], &dA[(size_t)i * 32] + 32); own.push_back(i);
        }
        std::vector<int32_t> km;
        const int on = oracle(F, occ, false, q, qd, 1, 64, 0.9f, 1, km);
        ORBmatcher m(0.9f, true);
        const int n = m.SearchByProjection(F, &KFa, found, 10.f, 64);
        CHECK(n == on && n > 50);
        for (int i = 0; i < nB; i++) CHECK(F.mvpMapPoints[i] == (km[i] >= 0 ? &mps[own[km[i]]] : (occ[i] ? &pre : (MapPoint*)NULL)));
        std::printf("glue relocalisation: %d matches\n", n);
    }
    // ---- 4./5. Sim3 projection searches into key frame B ----
    {
        KeyFrame KFb(fx, fy, cx, cy, 40.f, -1, 0, 0, W, H, 64.f / W, 48.f / H, sf, invSig2);
        KFb.N = nB; KFb.mvKeys = kB; KFb.mvKeysUn = kB; KFb.mvuRight.assign(nB, -1.f); KFb.mDescriptors = descMat(dB); KFb.mpCamera = &cam;
        KFb.mfLogScaleFactor = std::log(1.2f);
        cv::Mat Scw = cv::Mat::eye(4, 4, CV_32F);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) Scw.at<float>(r, c) *= 1.5f;   // s = 1.5, R = I, t = 0: projections unchanged
        std::vector<MapPoint*> matched(nB, (MapPoint*)NULL), matched2;
        for (int i = 0; i < nB; i += 6) matched[i] = &mps[(i * 7) % nA];
        matched2 = matched;
        std::set<MapPoint*> already(matched.begin(), matched.end());
        already.erase((MapPoint*)NULL);
        std::vector<uint8_t> occ(nB);
        for (int i = 0; i < nB; i++) occ[i] = matched[i] != NULL;
        Frame Fb; Fb.N = nB; Fb.mvKeysUn = kB; Fb.mDescriptors = descMat(dB);
        std::vector<Query> q; std::vector<uint8_t> qd; std::vector<int> own;
        for (int i = 0; i < nA; i++) {
            MapPoint* p = &mps[i];
            if (p->mbBad || already.count(p)) continue;
            const float X = p->mWorldPos.at<float>(0), Y = p->mWorldPos.at<float>(1), Z = p->mWorldPos.at<float>(2);
            const float u = fx * X / Z + cx, v = fy * Y / Z + cy;
            if (!KFb.IsInImage(u, v)) continue;
            const float dist = (float)cv::norm(p->mWorldPos);
            if (dist < p->GetMinDistanceInvariance() || dist > p->GetMaxDistanceInvariance()) continue;
            const int L = p->PredictScale(dist, &KFb);
            q.push_back(Query{u, v, 8.f * sf[L], 0.f, 0.f, (int16_t)(L - 1), (int16_t)L, 1u | 4u});
            qd.insert(qd.end(), &dA[(size_t)i * 32], &dA[(size_t)i * 32] + 32); own.push_back(i);
        }
        std::vector<int32_t> km;
        const int on = oracle(Fb, occ, false, q, qd, 1, 37, 0.9f, 0, km);   // floor(50 * 0.75)
        ORBmatcher m(0.75f, true);
        const int n = m.SearchByProjection(&KFb, Scw, vpMPs, matched, 8, 0.75f);
        CHECK(n == on && n > 30);
        for (int i = 0; i < nB; i++) CHECK(matched[i] == (km[i] >= 0 ? &mps[own[km[i]]] : matched2[i]));
        std::vector<KeyFrame*> srcKF(nA, &KFa), matchedKF(nB, (KeyFrame*)NULL);
        std::vector<MapPoint*> matched3 = matched2;
        const int n2 = m.SearchByProjection(&KFb, Scw, vpMPs, srcKF, matched3, matchedKF, 8, 0.75f);
        CHECK(n2 == on);
        for (int i = 0; i < nB; i++) CHECK(matched3[i] == matched[i] && matchedKF[i] == (km[i] >= 0 ? &KFa : (KeyFrame*)NULL));
        std::printf("glue Sim3 searches: %d matches\n", n);
    }
    // ---- 6./7. SearchByBoW ----
    {
        Frame F = fresh_frame();
        F.mFeatVec = featVec(dB, nB, 7);
        std::vector<uint8_t> valid(nA);
        for (int i = 0; i < nA; i++) valid[i] = KFa.mvpMapPoints[i] && !KFa.mvpMapPoints[i]->mbBad;
        std::vector<int32_t> id1, st1, fe1, id2, st2, fe2;
        csr(KFa.mFeatVec, id1, st1, fe1); csr(F.mFeatVec, id2, st2, fe2);
        std::vector<float> angA(nA), angB(nB);
        for (int i = 0; i < nA; i++) angA[i] = kA[i].angle;
        for (int i = 0; i < nB; i++) angB[i] = kB[i].angle;
        std::vector<int32_t> ofm(nB);
        const int on = omo_search_by_bow(dA.data(), angA.data(), valid.data(), id1.data(), st1.data(), fe1.data(), (int)id1.size(), dB.data(), angB.data(), nB, id2.data(),
                                         st2.data(), fe2.data(), (int)id2.size(), 0.7f, 1, ofm.data(), -1);
        ORBmatcher m(0.7f, true);
        std::vector<MapPoint*> vpMatches;
        const int n = m.SearchByBoW(&KFa, F, vpMatches);
        CHECK(n == on && n > 10 && (int)vpMatches.size() == nB);
        for (int j = 0; j < nB; j++) CHECK(vpMatches[j] == (ofm[j] >= 0 ? KFa.mvpMapPoints[ofm[j]] : (MapPoint*)NULL));
        // key frame / key frame
        KeyFrame KFb(fx, fy, cx, cy, 40.f, -1, 0, 0, W, H, 64.f / W, 48.f / H, sf, invSig2);
        std::vector<MapPoint> mpsB(nB);
        KFb.N = nB; KFb.mvKeys = kB; KFb.mvKeysUn = kB; KFb.mDescriptors = descMat(dB); KFb.mFeatVec = F.mFeatVec; KFb.mvpMapPoints.assign(nB, (MapPoint*)NULL);
        std::vector<uint8_t> valid2(nB, 0);
        for (int i = 0; i < nB; i++) if (i % 5 != 4) { KFb.mvpMapPoints[i] = &mpsB[i]; mpsB[i].mbBad = (i % 31 == 3); valid2[i] = !mpsB[i].mbBad; }
        std::vector<int32_t> om12(nA);
        const int onk = omo_search_by_bow_kf(dA.data(), angA.data(), valid.data(), id1.data(), st1.data(), fe1.data(), (int)id1.size(), nA, dB.data(), angB.data(),
                                             valid2.data(), id2.data(), st2.data(), fe2.data(), (int)id2.size(), nB, 0.8f, 1, om12.data());
        ORBmatcher mk(0.8f, true);
        std::vector<MapPoint*> vp12;
        const int nk = mk.SearchByBoW(&KFa, &KFb, vp12);
        CHECK(nk == onk && nk > 10 && (int)vp12.size() == nA);
        for (int i = 0; i < nA; i++) CHECK(vp12[i] == (om12[i] >= 0 ? KFb.mvpMapPoints[om12[i]] : (MapPoint*)NULL));
        std::printf("glue SearchByBoW: KF-F %d, KF-KF %d matches\n", n, nk);
        // ---- 7b. SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse): the gather (epipole, R12 / t12, F12 or the
        //          fisheye pair record, key-frame views) against the flattened adapter driven by hand with the same quantities ----
        {
            auto pose = [](float ay, float tx, float ty, float tz) {
                cv::Mat T = cv::Mat::eye(4, 4, CV_32F);
                T.at<float>(0, 0) = std::cos(ay); T.at<float>(0, 2) = std::sin(ay); T.at<float>(2, 0) = -std::sin(ay); T.at<float>(2, 2) = std::cos(ay);
                T.at<float>(0, 3) = tx; T.at<float>(1, 3) = ty; T.at<float>(2, 3) = tz;
                return T;
            };
            Pinhole camP(fx, fy, cx, cy);
            KannalaBrandt8 camK(std::vector<float>{190.978f, 190.973f, 254.93f, 256.9f, 0.0034f, 0.0007f, -0.0020f, 0.0002f});
            std::vector<float> uRa(nA, -1.f), uRb(nB, -1.f);
            for (int i = 0; i < nA; i += 3) uRa[i] = kA[i].pt.x - 3.f;
            for (int i = 0; i < nB; i += 4) uRb[i] = kB[i].pt.x - 2.f;
            KFa.mvuRight = uRa; KFb.mvuRight = uRb;
            KFa.SetPose(pose(0.f, 0.f, 0.f, 0.f)); KFb.SetPose(pose(0.02f, -0.35f, 0.01f, 0.02f));
            std::vector<MapPoint*> keepA = KFa.mvpMapPoints, keepB = KFb.mvpMapPoints;
            GeometricCamera *camA0 = KFa.mpCamera, *camB0 = KFb.mpCamera;
            for (int i = 0; i < nA; i++) if (i % 4 != 0) KFa.mvpMapPoints[i] = NULL;     // most features are still untriangulated
            for (int i = 0; i < nB; i++) if (i % 5 != 0) KFb.mvpMapPoints[i] = NULL;
            orbslam3_hip::ORBmatcher dev(0.6f, false);
            orbslam3_hip::ORBmatcher::KeyFrameView Va, Vb;
            std::vector<uint8_t> ha(nA), hb(nB);
            for (int i = 0; i < nA; i++) ha[i] = KFa.mvpMapPoints[i] != NULL;
            for (int i = 0; i < nB; i++) hb[i] = KFb.mvpMapPoints[i] != NULL;
            Va.N = nA; Va.keysUn = (const orb_keypoint*)KFa.mvKeysUn.data(); Va.descriptors = KFa.mDescriptors.data; Va.uRight = uRa.data(); Va.hasMapPoint = ha.data();
            Vb.N = nB; Vb.keysUn = (const orb_keypoint*)KFb.mvKeysUn.data(); Vb.descriptors = KFb.mDescriptors.data; Vb.uRight = uRb.data(); Vb.hasMapPoint = hb.data();
            csr(KFa.mFeatVec, Va.nodeId, Va.nodeStart, Va.featIdx); csr(KFb.mFeatVec, Vb.nodeId, Vb.nodeStart, Vb.featIdx);
            cv::Mat R1w = KFa.GetRotation(), t1w = KFa.GetTranslation(), R2w = KFb.GetRotation(), t2w = KFb.GetTranslation();
            cv::Mat R12 = R1w * R2w.t(), t12 = -R1w * R2w.t() * t2w + t1w;
            cv::Mat C2 = R2w * KFa.GetCameraCenter() + t2w;
            ORBmatcher mt(0.6f, false);
            for (int fish = 0; fish < 2; fish++) {
                KFa.mpCamera = fish ? (GeometricCamera*)&camK : (GeometricCamera*)&camP;
                KFb.mpCamera = KFa.mpCamera;
                const cv::Point2f ep = KFb.mpCamera->project(C2);
                const float epf[2] = {ep.x, ep.y};
                std::vector<std::pair<size_t, size_t>> want, got;
                int nw;
                if (!fish) {
                    cv::Mat t12x(3, 3, CV_32F);
                    const float x = t12.at<float>(0), y = t12.at<float>(1), z = t12.at<float>(2), sk[9] = {0, -z, y, z, 0, -x, -y, x, 0};
                    for (int i = 0; i < 9; i++) t12x.at<float>(i / 3, i % 3) = sk[i];
                    cv::Mat Fm = camP.toK().t().inv() * t12x * R12 * camP.toK().inv();
                    float Ff[9];
                    for (int i = 0; i < 9; i++) Ff[i] = Fm.at<float>(i / 3, i % 3);
                    nw = dev.SearchForTriangulation(Va, Vb, Ff, epf, KFb.mvLevelSigma2.data(), sf.data(), 8, want, false, false);
                } else {
                    orbm_tri_kb8_pair P{};
                    P.n_cams = 1;
                    for (int i = 0; i < 8; i++) { P.k1[0][i] = camK.getParameter(i); P.k2[0][i] = camK.getParameter(i); }
                    for (int i = 0; i < 9; i++) P.R12[0][i] = R12.at<float>(i / 3, i % 3);
                    for (int i = 0; i < 3; i++) P.t12[0][i] = t12.at<float>(i);
                    P.ep[0] = ep.x; P.ep[1] = ep.y;
                    for (int i = 0; i < 8; i++) { P.level_sigma2_1[i] = KFa.mvLevelSigma2[i]; P.level_sigma2_2[i] = KFb.mvLevelSigma2[i]; P.scale_factors_2[i] = sf[i]; }
                    Va.uRight = Vb.uRight = nullptr;
                    nw = dev.SearchForTriangulationKB8(Va, -1, Vb, -1, P, want, false, false);
                }
                const int ng = mt.SearchForTriangulation(&KFa, &KFb, cv::Mat(), got, false, false);
                CHECK(ng == nw && got == want);
                if (!fish) CHECK(ng > 5);
                std::printf("glue SearchForTriangulation (%s): %d pairs\n", fish ? "fisheye" : "pinhole", ng);
            }
            // a fisheye rig: [mvKeys | mvKeysRight], mpCamera2, the four left / right combinations of (R12, t12)
            {
                const int nlA = nA / 2, nlB = nB / 2;
                KeyFrame Ra(fx, fy, cx, cy, 40.f, nlA, 0, 0, W, H, 64.f / W, 48.f / H, sf, invSig2), Rb(fx, fy, cx, cy, 40.f, nlB, 0, 0, W, H, 64.f / W, 48.f / H, sf, invSig2);
                KeyFrame* R[2] = {&Ra, &Rb};
                const std::vector<cv::KeyPoint>* ks[2] = {&kA, &kB};
                const int nl[2] = {nlA, nlB}, nn[2] = {nA, nB};
                std::vector<uint8_t> hr[2];
                orbslam3_hip::ORBmatcher::KeyFrameView Vr[2];
                for (int q = 0; q < 2; q++) {
                    R[q]->N = nn[q];
                    R[q]->mvKeys.assign(ks[q]->begin(), ks[q]->begin() + nl[q]);
                    R[q]->mvKeysRight.assign(ks[q]->begin() + nl[q], ks[q]->end());
                    R[q]->mDescriptors = (q ? KFb : KFa).mDescriptors; R[q]->mFeatVec = (q ? KFb : KFa).mFeatVec;
                    R[q]->mvpMapPoints = (q ? KFb : KFa).mvpMapPoints;
                    R[q]->mpCamera = &camK; R[q]->mpCamera2 = &camK;
                    R[q]->mTlr = pose(0.01f, 0.1f, 0.f, 0.f);
                    R[q]->SetPose((q ? KFb : KFa).GetPose());
                    hr[q].resize(nn[q]);
                    for (int i = 0; i < nn[q]; i++) hr[q][i] = R[q]->mvpMapPoints[i] != NULL;
                    Vr[q].N = nn[q]; Vr[q].keysUn = (const orb_keypoint*)ks[q]->data(); Vr[q].descriptors = R[q]->mDescriptors.data; Vr[q].hasMapPoint = hr[q].data();
                    csr(R[q]->mFeatVec, Vr[q].nodeId, Vr[q].nodeStart, Vr[q].featIdx);
                }
                orbm_tri_kb8_pair P{};
                P.n_cams = 2;
                for (int c = 0; c < 2; c++) for (int i = 0; i < 8; i++) { P.k1[c][i] = camK.getParameter(i); P.k2[c][i] = camK.getParameter(i); }
                cv::Mat Rl[2] = {Ra.GetRotation(), Rb.GetRotation()}, Rr[2] = {Ra.GetRightRotation(), Rb.GetRightRotation()};
                cv::Mat tl[2] = {Ra.GetTranslation(), Rb.GetTranslation()}, tr[2] = {Ra.GetRightTranslation(), Rb.GetRightTranslation()};
                for (int r1 = 0; r1 < 2; r1++)
                    for (int r2 = 0; r2 < 2; r2++) {
                        const cv::Mat& A1 = r1 ? Rr[0] : Rl[0]; const cv::Mat& A2 = r2 ? Rr[1] : Rl[1];
                        const cv::Mat& b1 = r1 ? tr[0] : tl[0]; const cv::Mat& b2 = r2 ? tr[1] : tl[1];
                        cv::Mat Rc = A1 * A2.t(), tc = A1 * (-A2.t() * b2) + b1;
                        for (int i = 0; i < 9; i++) P.R12[r1 * 2 + r2][i] = Rc.at<float>(i / 3, i % 3);
                        for (int i = 0; i < 3; i++) P.t12[r1 * 2 + r2][i] = tc.at<float>(i);
                    }
                const cv::Point2f ep = camK.project(Rb.GetRotation() * Ra.GetCameraCenter() + Rb.GetTranslation());
                P.ep[0] = ep.x; P.ep[1] = ep.y;
                for (int i = 0; i < 8; i++) { P.level_sigma2_1[i] = Ra.mvLevelSigma2[i]; P.level_sigma2_2[i] = Rb.mvLevelSigma2[i]; P.scale_factors_2[i] = sf[i]; }
                std::vector<std::pair<size_t, size_t>> want, got;
                const int nw = dev.SearchForTriangulationKB8(Vr[0], nlA, Vr[1], nlB, P, want, false, false);
                const int ng = mt.SearchForTriangulation(&Ra, &Rb, cv::Mat(), got, false, false);
                CHECK(ng == nw && got == want);
                std::printf("glue SearchForTriangulation (fisheye rig): %d pairs\n", ng);
            }
            KFa.mvpMapPoints = keepA; KFb.mvpMapPoints = keepB; KFa.mpCamera = camA0; KFb.mpCamera = camB0;
        }
    }
    // ---- 7a'. SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) against the oracle's restatement of :838-979 ----
    {
        Frame F1 = fresh_frame(), F2 = fresh_frame();
        F1.N = nA; F1.mvKeys = kA; F1.mvKeysUn = kA; F1.mDescriptors = descMat(dA); F1.mvuRight.assign(nA, -1.f); F1.mvpMapPoints.assign(nA, (MapPoint*)NULL);
        std::vector<cv::Point2f> prev(nA);
        std::vector<float> oprev(2 * (size_t)nA);
        for (int i = 0; i < nA; i++) { prev[i] = cv::Point2f(kA[i].pt.x + 5.f, kA[i].pt.y - 3.f); oprev[2 * i] = prev[i].x; oprev[2 * i + 1] = prev[i].y; }
        std::vector<int32_t> om(nA);
        const int on = omo_search_for_initialization(kA.data(), dA.data(), nA, kB.data(), dB.data(), nB, Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv,
                                                     Frame::mfGridElementHeightInv, oprev.data(), 30, 0.9f, 1, om.data());
        ORBmatcher mi(0.9f, true);
        std::vector<int> m12;
        const int n = mi.SearchForInitialization(F1, F2, prev, m12, 30);
        CHECK(n == on && n > 20 && (int)m12.size() == nA);
        for (int i = 0; i < nA; i++) {
            CHECK(m12[i] == om[i]);
            CHECK(prev[i].x == oprev[2 * i] && prev[i].y == oprev[2 * i + 1]);
        }
        std::printf("glue SearchForInitialization: %d matches\n", n);
    }
    // ---- 7b'. SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th): two key frames looking at the same points under a similarity; the
    //           window queries are rebuilt here from the scene's construction and given to the oracle's restatement of :2044-2220 ----
    {
        KeyFrame K1(fx, fy, cx, cy, 40.f, -1, 0, 0, W, H, 64.f / W, 48.f / H, sf, invSig2), K2(fx, fy, cx, cy, 40.f, -1, 0, 0, W, H, 64.f / W, 48.f / H, sf, invSig2);
        K1.N = nA; K1.mvKeys = kA; K1.mvKeysUn = kA; K1.mDescriptors = descMat(dA); K1.mfLogScaleFactor = std::log(1.2f);
        K2.N = nB; K2.mvKeys = kB; K2.mvKeysUn = kB; K2.mDescriptors = descMat(dB); K2.mfLogScaleFactor = std::log(1.2f);
        cv::Mat I4 = cv::Mat::eye(4, 4, CV_32F);
        K1.SetPose(I4); K2.SetPose(I4);
        // Sim3 from camera 2 to camera 1: identity rotation, scale 1, translation 0 — every point projects to its own pixel in the other view, so the
        // queries are known in closed form (and the frames are the shifted pair A / B: windows find the true partners)
        const float s12 = 1.0f;
        cv::Mat R12 = cv::Mat::eye(3, 3, CV_32F), t12(3, 1, CV_32F);
        std::vector<MapPoint> p1(nA), p2(nB);
        K1.mvpMapPoints.assign(nA, (MapPoint*)NULL); K2.mvpMapPoints.assign(nB, (MapPoint*)NULL);
        struct Q { float u, v, radius, ur, angle; int16_t lo, hi; uint32_t flags; };
        std::vector<Q> q12(nA), q21(nB);
        std::vector<uint8_t> d12((size_t)nA * 32, 0), d21((size_t)nB * 32, 0);
        auto lift = [&](MapPoint& p, const cv::KeyPoint& k, float du, float dv, const uint8_t* desc, Q& q, uint8_t* qd, KeyFrame* into) {
            const float z = 4.f, u = k.pt.x + du, v = k.pt.y + dv;     // lands at (u, v) in the other view
            p.mWorldPos = cv::Mat(3, 1, CV_32F);
            p.mWorldPos.at<float>(0) = (u - cx) / fx * z; p.mWorldPos.at<float>(1) = (v - cy) / fy * z; p.mWorldPos.at<float>(2) = z;
            const float d = (float)cv::norm(p.mWorldPos);
            p.mfMaxDistance = d * sf[k.octave] * 0.999f; p.mfMinDistance = 0.01f;
            p.mDescriptor = cv::Mat(1, 32, CV_8U);
            std::memcpy(p.mDescriptor.data, desc, 32);
            // the reference's projection of this point (:2056-2062): u = fx * (X / Z) + cx
            const float invz = 1.0 / p.mWorldPos.at<float>(2), xx = p.mWorldPos.at<float>(0) * invz, yy = p.mWorldPos.at<float>(1) * invz;
            const float uu = fx * xx + cx, vv = fy * yy + cy;
            if (!into->IsInImage(uu, vv)) return;
            const int L = p.PredictScale(d, into);
            q = Q{uu, vv, 7.5f * sf[L], 0.f, 0.f, (int16_t)(L - 1), (int16_t)L, 1u};
            std::memcpy(qd, desc, 32);
        };
        for (int i = 0; i < nA; i++) { q12[i] = Q{}; if (i % 4 != 3) { lift(p1[i], kA[i], 6.f, -4.f, &dA[(size_t)i * 32], q12[i], &d12[(size_t)i * 32], &K2); K1.mvpMapPoints[i] = &p1[i]; } }
        for (int j = 0; j < nB; j++) { q21[j] = Q{}; if (j % 5 != 4) { lift(p2[j], kB[j], -6.f, 4.f, &dB[(size_t)j * 32], q21[j], &d21[(size_t)j * 32], &K1); K2.mvpMapPoints[j] = &p2[j]; } }
        p1[8].mbBad = true; q12[8] = Q{}; std::memset(&d12[8 * 32], 0, 32);                       // a bad point takes no part
        std::vector<MapPoint*> m12(nA, (MapPoint*)NULL);
        m12[12] = &p2[20]; p2[20].mObservations[&K2] = std::make_tuple(20, -1);                     // an earlier match: both ends are taken
        q12[12] = Q{}; std::memset(&d12[12 * 32], 0, 32); q21[20] = Q{}; std::memset(&d21[20 * 32], 0, 32);
        std::vector<int32_t> om(nA);
        const int on = omo_search_by_sim3(kA.data(), dA.data(), nA, 0.f, 0.f, 64.f / W, 48.f / H, kB.data(), dB.data(), nB, 0.f, 0.f, 64.f / W, 48.f / H, q12.data(),
                                          d12.data(), q21.data(), d21.data(), om.data());
        ORBmatcher ms(0.75f, true);
        const int n = ms.SearchBySim3(&K1, &K2, m12, s12, R12, t12, 7.5f);
        CHECK(n == on && n > 20);
        for (int i = 0; i < nA; i++) CHECK(m12[i] == (i == 12 ? &p2[20] : om[i] >= 0 ? K2.mvpMapPoints[om[i]] : (MapPoint*)NULL));
        std::printf("glue SearchBySim3: %d mutual matches\n", n);
    }
    // ---- 7c. Fuse(pKF, vpMapPoints, th, bRight = false): map points built to land on chosen features of a key frame; every branch of the serial
    //          scatter (:1828-1855) and its order dependence — two points on one feature, a point already in the key frame, bad / null points ----
    {
        KeyFrame KF(fx, fy, cx, cy, 40.f, -1, 0, 0, W, H, 64.f / W, 48.f / H, sf, invSig2);
        Pinhole camP(fx, fy, cx, cy);
        KF.mpCamera = &camP; KF.N = nB; KF.mvKeys = kB; KF.mvKeysUn = kB; KF.mDescriptors = descMat(dB); KF.mvuRight.assign(nB, -1.f);
        KF.mfLogScaleFactor = std::log(1.2f); KF.mnScaleLevels = 8;
        KF.mvpMapPoints.assign(nB, (MapPoint*)NULL);
        cv::Mat I4 = cv::Mat::eye(4, 4, CV_32F);
        KF.SetPose(I4);
        std::vector<MapPoint> inKF(nB);                 // map points the key frame already has on some features
        std::vector<int> target;                        // feature index each fused candidate is built for
        std::vector<MapPoint> cand;
        cand.reserve(400);
        std::vector<MapPoint*> vp;
        auto make = [&](int j, float depth, int nobs) {
            MapPoint p;
            const float u = kB[j].pt.x, v = kB[j].pt.y;
            p.mWorldPos = cv::Mat(3, 1, CV_32F);
            p.mWorldPos.at<float>(0) = (u - cx) * depth / fx; p.mWorldPos.at<float>(1) = (v - cy) * depth / fy; p.mWorldPos.at<float>(2) = depth;
            const float d = (float)cv::norm(p.mWorldPos);
            p.mNormalVector = p.mWorldPos / d;
            p.mfMaxDistance = d * sf[kB[j].octave] * 0.999f / 1.2f;   // GetMaxDistanceInvariance() = 1.2 * this -> PredictScale == the feature's octave
            p.mfMinDistance = 0.01f;
            p.mDescriptor = cv::Mat(1, 32, CV_8U);
            std::memcpy(p.mDescriptor.data, &dB[(size_t)j * 32], 32);
            p.mDescriptor.data[j % 32] ^= 0x11;                      // two bits off the feature's own descriptor
            p.nObs = nobs;
            return p;
        };
        // PredictScale uses mfMaxDistance (not the 1.2x getter): ratio = mfMaxDistance / dist
        int expectFused = 0;
        for (int j = 5; j < nB && (int)cand.size() < 300; j += 3) {
            if (kB[j].octave < 1) continue;                          // level window [oct-1, oct] must contain the feature
            const int kind = (int)cand.size() % 6;
            MapPoint p = make(j, 3.f + (j % 7), 3);
            p.mfMaxDistance = (float)cv::norm(p.mWorldPos) * sf[kB[j].octave] * 0.999f;
            if (kind == 1) { inKF[j].nObs = 9; KF.mvpMapPoints[j] = &inKF[j]; }          // feature's point has more observations: candidate is replaced by it
            if (kind == 2) { inKF[j].nObs = 1; KF.mvpMapPoints[j] = &inKF[j]; }          // fewer: the feature's point is replaced by the candidate
            if (kind == 3) p.mbBad = true;                                               // skipped
            cand.push_back(p); target.push_back(j);
        }
        for (size_t i = 0; i < cand.size(); i++) {
            vp.push_back(&cand[i]);
            if (i % 6 == 4) { cand[i].mObservations[&KF] = std::make_tuple(0, -1); }     // already observed in this key frame: skipped
            if (i % 6 == 5 && i + 6 < cand.size()) {                                     // a second candidate for the SAME feature right behind it
                MapPoint p2 = make(target[i], 3.f + (target[i] % 7), 7);
                p2.mfMaxDistance = (float)cv::norm(p2.mWorldPos) * sf[kB[target[i]].octave] * 0.999f;
                cand.push_back(p2); target.push_back(target[i]);
            }
        }
        vp.clear();
        for (size_t i = 0; i < cand.size(); i++) vp.push_back(&cand[i]);
        vp.push_back(NULL);
        // expectation by construction, applied serially
        std::vector<MapPoint*> expMP = KF.mvpMapPoints;
        std::vector<int> expBad(cand.size(), 0), expObs(cand.size());
        std::vector<int> inBad(nB, 0);
        for (size_t i = 0; i < cand.size(); i++) expObs[i] = cand[i].nObs, expBad[i] = cand[i].mbBad;
        auto idxOf = [&](MapPoint* p) { return (int)(p - &cand[0]); };
        for (size_t i = 0; i < cand.size(); i++) {
            if (expBad[i] || cand[i].mObservations.count(&KF)) continue;
            const int j = target[i];
            MapPoint* in = expMP[j];
            if (in) {
                const bool isCand = in >= &cand[0] && in < &cand[0] + cand.size();
                const int inObs = isCand ? expObs[idxOf(in)] : in->nObs;
                const bool inIsBad = isCand ? expBad[idxOf(in)] : inBad[j];
                if (!inIsBad) {
                    if (inObs > expObs[i]) { expBad[i] = 1; if (isCand) expObs[idxOf(in)] += expObs[i]; }
                    else { if (isCand) expBad[idxOf(in)] = 1; else inBad[j] = 1; expObs[i] += inObs; }
                }
            } else { expMP[j] = &cand[i]; expObs[i]++; }
            expectFused++;
        }
        ORBmatcher mf(0.6f, true);
        const int nf = mf.Fuse(&KF, vp, 3.0f, false);
        CHECK(nf == expectFused && nf > 100);
        for (int j = 0; j < nB; j++) CHECK(KF.mvpMapPoints[j] == expMP[j]);
        for (size_t i = 0; i < cand.size(); i++) CHECK(cand[i].mbBad == (expBad[i] != 0) && cand[i].nObs == expObs[i]);
        for (int j = 0; j < nB; j++) CHECK(inKF[j].mbBad == (inBad[j] != 0));
        std::printf("glue Fuse: %zu candidates, %d fused\n", cand.size(), nf);
        // the Sim3 overload on what is left: fresh points on features that are still free or already taken (-> vpReplacePoint), Scw = identity
        {
            std::vector<MapPoint> more;
            more.reserve(200);
            std::vector<int> tgt;
            for (int j = 4; j < nB && (int)more.size() < 150; j += 5) {
                if (kB[j].octave < 1) continue;
                MapPoint p = make(j, 5.f, 2);
                p.mfMaxDistance = (float)cv::norm(p.mWorldPos) * sf[kB[j].octave] * 0.999f;
                if (more.size() % 7 == 3) p.mbBad = true;
                more.push_back(p); tgt.push_back(j);
            }
            std::vector<MapPoint*> vq, repl(more.size(), (MapPoint*)NULL), expRepl(more.size(), (MapPoint*)NULL);
            for (size_t i = 0; i < more.size(); i++) vq.push_back(&more[i]);
            vq[5] = KF.mvpMapPoints[target[0]] ? KF.mvpMapPoints[target[0]] : vq[5];   // a point the key frame already has: skipped (spAlreadyFound)
            std::vector<MapPoint*> expMP2 = KF.mvpMapPoints;
            const std::set<MapPoint*> have = KF.GetMapPoints();
            int expN = 0;
            for (size_t i = 0; i < vq.size(); i++) {
                if (vq[i]->mbBad || have.count(vq[i])) continue;
                const int j = tgt[i];
                if (expMP2[j]) { if (!expMP2[j]->mbBad) expRepl[i] = expMP2[j]; }
                else expMP2[j] = vq[i];
                expN++;
            }
            const int n2 = mf.Fuse(&KF, I4, vq, 3.0f, repl);
            CHECK(n2 == expN && n2 > 60);
            for (size_t i = 0; i < vq.size(); i++) CHECK(repl[i] == expRepl[i]);
            for (int j = 0; j < nB; j++) CHECK(KF.mvpMapPoints[j] == expMP2[j]);
            std::printf("glue Fuse (Sim3): %zu points, %d fused\n", vq.size(), n2);
        }
    }
    // ---- 8. Optimizer::LocalBundleAdjustment on a small mock map, against the flattened LbaLinearizer path driven by hand ----
    {
        Map map; map.mnInitKFid = 0;
        const int NK = 9, NP = 260;
        std::vector<KeyFrame*> kfs;
        std::vector<MapPoint> pts(NP);
        std::vector<cv::Mat> Xtrue(NP);
        for (int j = 0; j < NP; j++) {
            Xtrue[j] = cv::Mat(3, 1, CV_32F);
            Xtrue[j].at<float>(0) = frand(-3.f, 3.f); Xtrue[j].at<float>(1) = frand(-2.f, 2.f); Xtrue[j].at<float>(2) = frand(4.f, 9.f);
            pts[j].mnId = j; pts[j].mpMap = &map; pts[j].mWorldPos = Xtrue[j].clone();
            for (int k = 0; k < 3; k++) pts[j].mWorldPos.at<float>(k) += frand(-0.03f, 0.03f);
        }
        for (int k = 0; k < NK; k++) {
            KeyFrame* kf = new KeyFrame(fx, fy, cx, cy, 40.f, -1, 0, 0, W, H, 64.f / W, 48.f / H, sf, invSig2);
            kf->mnId = k; kf->mpMap = &map; kf->mpCamera = &cam;
            cv::Mat T = cv::Mat::eye(4, 4, CV_32F);
            const float a = 0.02f * k;
            T.at<float>(0, 0) = std::cos(a); T.at<float>(0, 2) = std::sin(a); T.at<float>(2, 0) = -std::sin(a); T.at<float>(2, 2) = std::cos(a);
            T.at<float>(0, 3) = -0.15f * k; T.at<float>(1, 3) = 0.01f * k;
            // observations from the TRUE pose; the stored pose is perturbed
            for (int j = 0; j < NP; j++) {
                if ((j + 3 * k) % 4 == 0) continue;
                cv::Mat Xc = T.rowRange(0, 3).colRange(0, 3) * Xtrue[j] + T.rowRange(0, 3).col(3);
                const float z = Xc.at<float>(2), u = fx * Xc.at<float>(0) / z + cx, v = fy * Xc.at<float>(1) / z + cy;
                if (z < 0.5f || u < 5 || u > W - 5 || v < 5 || v > H - 5) continue;
                cv::KeyPoint kp; kp.pt = cv::Point2f(u + frand(-0.5f, 0.5f), v + frand(-0.5f, 0.5f)); kp.octave = (int)(rnd() % 4);
                if ((j * 7 + k) % 41 == 0) kp.pt.x += 35.f;   // a gross outlier observation
                const int idx = (int)kf->mvKeysUn.size();
                kf->mvKeysUn.push_back(kp); kf->mvKeys.push_back(kp);
                kf->mvuRight.push_back((j % 3 == 0) ? kp.pt.x - 40.f / z : -1.f);
                kf->mvpMapPoints.push_back(&pts[j]);
                pts[j].mObservations[kf] = std::make_tuple(idx, -1);
            }
            kf->N = (int)kf->mvKeysUn.size();
            if (k > 0) { T.at<float>(0, 3) += frand(-0.02f, 0.02f); T.at<float>(2, 3) += frand(-0.02f, 0.02f); }
            kf->Tcw = T;
            kfs.push_back(kf);
        }
        KeyFrame* cur = kfs[NK - 1];
        for (int k = NK - 2; k >= 3; k--) cur->mvpOrderedConnectedKeyFrames.push_back(kfs[k]);   // key frames 0..2 only see local points: fixed cameras
        // the flattened path, driven by hand in the order the glue must produce (poses by ascending id; edges landmark-major in GetObservations order)
        orbslam3_hip::LbaLinearizer L;
        std::map<KeyFrame*, int> pidx;
        lba_camera c{}; c.model = LBA_CAM_PINHOLE; c.p[0] = fx; c.p[1] = fy; c.p[2] = cx; c.p[3] = cy; c.bf = 40.0; c.trl_q[3] = 1.0;
        const int camId = L.addCamera(c);
        for (int k = 0; k < NK; k++) { double p7[7]; orbslam3_hip::LbaLinearizer::poseFromTcw(kfs[k]->Tcw.ptr<float>(), 4, p7); pidx[kfs[k]] = L.addPose(p7, k < 3); }
        // local map points in the glue's discovery order: key frames cur, NK-2 .. 3, their map-point vectors in index order
        std::vector<MapPoint*> order; std::set<MapPoint*> seen;
        std::vector<KeyFrame*> locals = {cur};
        for (KeyFrame* k : cur->mvpOrderedConnectedKeyFrames) locals.push_back(k);
        for (KeyFrame* k : locals) for (MapPoint* p : k->mvpMapPoints) if (p && !seen.count(p)) { seen.insert(p); order.push_back(p); }
        std::vector<std::pair<KeyFrame*, MapPoint*>> eref; std::vector<bool> est;
        std::map<MapPoint*, int> lidx;
        for (MapPoint* p : order) {
            const double X[3] = {p->mWorldPos.at<float>(0), p->mWorldPos.at<float>(1), p->mWorldPos.at<float>(2)};
            const int l = L.addPoint(X); lidx[p] = l;
            for (auto& ob : p->mObservations) {
                KeyFrame* k = ob.first; const int i = std::get<0>(ob.second);
                const cv::KeyPoint& kp = k->mvKeysUn[i];
                const bool st = k->mvuRight[i] >= 0;
                L.addEdge(pidx[k], l, st ? LBA_EDGE_STEREO : LBA_EDGE_MONO, camId, kp.pt.x, kp.pt.y, st ? k->mvuRight[i] : 0.f, invSig2[kp.octave]);
                eref.push_back(std::make_pair(k, p)); est.push_back(st);
            }
        }
        orbslam3_hip::LbaHostSystem S0; L.computeErrors(S0);
        L.optimize(5); L.optimize(10);
        orbslam3_hip::LbaHostSystem S; L.computeErrors(S);
        CHECK(S.robustChi2 < 0.7 * S0.robustChi2);   // it really optimised
        int nOut = 0;
        for (size_t i = 0; i < eref.size(); i++) if (S.chi2[i] > (est[i] ? 7.815 : 5.991) || !(S.depth[i] > 0)) nOut++;
        // the glue
        int numFixed = -1; bool stop = false;
        Optimizer::LocalBundleAdjustment(cur, &stop, &map, numFixed);
        CHECK(numFixed == 3);
        int erased = 0;
        for (KeyFrame* k : kfs) erased += k->nErased;
        CHECK(erased == nOut && nOut > 3 && nOut < (int)eref.size() / 4);
        for (int k = 0; k < NK; k++) {
            const double* p7 = L.pose(pidx[kfs[k]]);
            if (k < 3) { CHECK(kfs[k]->nPoseSets == 0); continue; }
            CHECK(kfs[k]->nPoseSets == 1);
            for (int r = 0; r < 3; r++) CHECK(kfs[k]->Tcw.at<float>(r, 3) == (float)p7[r]);   // identical to the flattened path, bit for bit
        }
        for (MapPoint* p : order) {
            const double* X = L.point(lidx[p]);
            for (int r = 0; r < 3; r++) CHECK(p->mWorldPos.at<float>(r) == (float)X[r]);
            CHECK(p->nNormalUpdates == 1);
        }
        CHECK(order.size() > 200);
        CHECK(map.mnMapChange == 1);
        // pbStopFlag raised before the call: nothing is touched
        stop = true; const int before = kfs[5]->nPoseSets;
        Optimizer::LocalBundleAdjustment(cur, &stop, &map, numFixed);
        CHECK(kfs[5]->nPoseSets == before);
        std::printf("glue LocalBundleAdjustment: %zu edges, %d erased observations, %d fixed key frames\n", eref.size(), nOut, numFixed);
        for (KeyFrame* k : kfs) delete k;
    }
    // ---- 9. Optimizer::PoseOptimization(Frame*) on a mock frame (monocular + stereo observations, features without a map point, gross
    //         outliers), against the oracle's restatement of the g2o schedule on the same edges ----
    {
        const float fx = 458.654f, fy = 457.296f, cx = 367.215f, cy = 248.375f, bf = 47.9f;
        Pinhole cam(fx, fy, cx, cy);
        Frame F;
        F.mpCamera = &cam; F.mbf = bf; F.fx = fx; F.fy = fy; F.cx = cx; F.cy = cy;
        F.N = 420;
        F.mvInvLevelSigma2.resize(8);
        for (int l = 0; l < 8; l++) F.mvInvLevelSigma2[l] = 1.0f / std::pow(1.2f, 2.0f * l);
        F.mvKeysUn.resize(F.N); F.mvuRight.assign(F.N, -1.f); F.mvpMapPoints.assign(F.N, (MapPoint*)NULL); F.mvbOutlier.assign(F.N, true);
        std::vector<MapPoint> mps(F.N);
        // true pose: small rotation about y, translation; the frame starts from a perturbed pose
        const float ang = 0.05f, Rt[9] = {std::cos(ang), 0, std::sin(ang), 0, 1, 0, -std::sin(ang), 0, std::cos(ang)}, tt[3] = {0.1f, -0.05f, 0.2f};
        std::vector<pose_edge> flat;
        for (int i = 0; i < F.N; i++) {
            const float X[3] = {(float)(rnd() % 8000) / 1000.f - 4.f, (float)(rnd() % 4000) / 1000.f - 2.f, 4.f + (float)(rnd() % 6000) / 1000.f};
            float Xc[3];
            for (int r = 0; r < 3; r++) Xc[r] = Rt[r * 3] * X[0] + Rt[r * 3 + 1] * X[1] + Rt[r * 3 + 2] * X[2] + tt[r];
            const int oct = (int)(rnd() % 8);
            float u = fx * Xc[0] / Xc[2] + cx + ((float)(rnd() % 2000) / 1000.f - 1.f) * 0.6f, v = fy * Xc[1] / Xc[2] + cy + ((float)(rnd() % 2000) / 1000.f - 1.f) * 0.6f;
            if (i % 11 == 3) { u += 25.f; v -= 18.f; }                         // gross outlier
            F.mvKeysUn[i].pt = cv::Point2f(u, v); F.mvKeysUn[i].octave = oct;
            if (i % 3 == 1) F.mvuRight[i] = u - bf / Xc[2] + ((float)(rnd() % 2000) / 1000.f - 1.f) * 0.4f;   // stereo observation
            if (i % 7 == 5) continue;                                          // a feature without a map point
            mps[i].mWorldPos = cv::Mat(3, 1, CV_32F);
            for (int r = 0; r < 3; r++) mps[i].mWorldPos.at<float>(r) = X[r];
            F.mvpMapPoints[i] = &mps[i];
            flat.push_back(pose_edge{{X[0], X[1], X[2]}, {u, v, F.mvuRight[i] < 0 ? 0.f : F.mvuRight[i]}, F.mvInvLevelSigma2[oct],
                                     (int16_t)(F.mvuRight[i] < 0 ? LBA_EDGE_MONO : LBA_EDGE_STEREO), 0});
        }
        const float a0 = 0.03f;
        F.mTcw = cv::Mat::eye(4, 4, CV_32F);
        F.mTcw.at<float>(0, 0) = std::cos(a0); F.mTcw.at<float>(0, 2) = std::sin(a0); F.mTcw.at<float>(2, 0) = -std::sin(a0); F.mTcw.at<float>(2, 2) = std::cos(a0);
        F.mTcw.at<float>(0, 3) = 0.05f; F.mTcw.at<float>(1, 3) = 0.f; F.mTcw.at<float>(2, 3) = 0.1f;
        double p7[7], pref[7];
        orbslam3_hip::LbaLinearizer::poseFromTcw(F.mTcw.ptr<float>(), 4, p7);
        lba_camera lc{};
        lc.model = LBA_CAM_PINHOLE; lc.p[0] = fx; lc.p[1] = fy; lc.p[2] = cx; lc.p[3] = cy; lc.bf = bf; lc.trl_q[3] = 1.0;
        std::vector<uint8_t> oref(flat.size());
        const int gref = opo_pose_optimize(p7, flat.data(), (int)flat.size(), &lc, pref, oref.data());
        const int good = Optimizer::PoseOptimization(&F);
        CHECK(good == gref && good > 250 && good < (int)flat.size());
        size_t k = 0;
        int nOutl = 0;
        for (int i = 0; i < F.N; i++) {
            if (!F.mvpMapPoints[i]) { CHECK(F.mvbOutlier[i] == true); continue; }      // untouched
            CHECK(F.mvbOutlier[i] == (oref[k] != 0));
            nOutl += oref[k] != 0;
            k++;
        }
        CHECK(k == flat.size() && nOutl >= 30);
        double q7[7];
        orbslam3_hip::LbaLinearizer::poseFromTcw(F.mTcw.ptr<float>(), 4, q7);         // the pose SetPose received (float matrix)
        for (int j = 0; j < 7; j++) CHECK(std::fabs(q7[j] - pref[j]) < 2e-6);
        for (int r = 0; r < 3; r++) CHECK(std::fabs(F.mTcw.at<float>(r, 3) - tt[r]) < 0.02f);   // and it found the true pose
        std::printf("glue PoseOptimization: %zu observations, %d inliers, %d outliers\n", flat.size(), good, nOutl);
    }
    // ---- 10. Optimizer::LocalInertialBA(pKF, pbStopFlag, pMap, bLarge, bRecInit) on a mock inertial map (a temporal chain of key frames with
    //          preintegrations, monocular + stereo observations, gross outliers, near and far points), against InertialBA driven by hand in the
    //          order the glue must produce.  Scene A: the chain is longer than the window (the key frame before it is fixed with its V / G / A
    //          vertices, an older observer is fixed by the observation walk).  Scene B: the chain ends inside the window (its oldest key frame
    //          is popped and fixed), bLarge + bRecInit. ----
    KannalaBrandt8 fishL(std::vector<float>{190.f, 190.f, 240.f, 180.f, 0.003f, 0.0007f, -0.002f, 0.0002f}), fishR(std::vector<float>{191.f, 189.f, 238.f, 181.f, 0.002f, 0.0009f, -0.001f, 0.0001f});
    for (int scene = 0; scene < 3; scene++) {
        const bool bLarge = scene == 1, bRecInit = scene == 1, rigScene = scene == 2;   // scene C: scene A's window on a fisheye rig (right-camera edges)
        const int NK = scene == 0 ? 8 : (scene == 1 ? 5 : 6), NP = 140;
        cv::Mat Trl = cv::Mat::eye(4, 4, CV_32F);   // x_right = Rrl x_left + trl
        Trl.at<float>(0, 0) = std::cos(0.02f); Trl.at<float>(0, 2) = std::sin(0.02f); Trl.at<float>(2, 0) = -std::sin(0.02f); Trl.at<float>(2, 2) = std::cos(0.02f);
        Trl.at<float>(0, 3) = -0.1f; Trl.at<float>(1, 3) = 0.001f; Trl.at<float>(2, 3) = 0.002f;
        const float dtk = 0.3f;
        Map map; map.mnInitKFid = 0; map.mbIsInertial = true; map.nKeyFrames = scene == 1 ? 20 : NK;
        std::vector<KeyFrame*> kfs;
        std::vector<IMU::Preintegrated*> pre;
        std::vector<MapPoint> pts(NP);
        std::vector<cv::Mat> Xtrue(NP);
        for (int j = 0; j < NP; j++) {
            Xtrue[j] = cv::Mat(3, 1, CV_32F);
            Xtrue[j].at<float>(0) = frand(-2.f, 4.f); Xtrue[j].at<float>(1) = frand(-2.f, 2.f); Xtrue[j].at<float>(2) = frand(4.f, 14.f);
            pts[j].mnId = j; pts[j].mpMap = &map; pts[j].mWorldPos = Xtrue[j].clone();
            for (int k = 0; k < 3; k++) pts[j].mWorldPos.at<float>(k) += frand(-0.03f, 0.03f);
            pts[j].mTrackDepth = Xtrue[j].at<float>(2);   // near (< 10: the 1.5x chi2 rule) and far points
        }
        cv::Mat Tcb = cv::Mat::eye(4, 4, CV_32F), Tbc = cv::Mat::eye(4, 4, CV_32F);
        Tcb.at<float>(0, 3) = 0.05f; Tbc.at<float>(0, 3) = -0.05f;
        for (int k = 0; k < NK; k++) {
            cv::Mat T = cv::Mat::eye(4, 4, CV_32F);
            T.at<float>(0, 3) = 0.05f - 0.3f * k;              // truth: body at (0.3 k, 0, 0), Rwb = I, camera 5 cm beside it
            std::vector<cv::KeyPoint> keysL, keysR;
            std::vector<float> uR;
            std::vector<MapPoint*> mpL, mpR;
            std::vector<int> jl, jr;
            for (int j = 0; j < NP; j++) {
                if ((j + 2 * k) % 5 == 0) continue;
                cv::Mat Xc = T.rowRange(0, 3).colRange(0, 3) * Xtrue[j] + T.rowRange(0, 3).col(3);
                const float z = Xc.at<float>(2);
                const cv::Point2f pl = rigScene ? fishL.project(Xc) : cam.project(Xc);
                const float u = pl.x, v = pl.y;
                const bool inL = !(u < 5 || u > W - 5 || v < 5 || v > H - 5) && !(rigScene && j % 7 == 3);   // rig: some points are seen by the right camera only
                if (inL) {
                    cv::KeyPoint kp; kp.pt = cv::Point2f(u + frand(-0.5f, 0.5f), v + frand(-0.5f, 0.5f)); kp.octave = (int)(rnd() % 3);
                    if ((j * 5 + k) % 37 == 0) kp.pt.y += 30.f;   // a gross outlier observation
                    else if ((j * 3 + k) % 7 == 0 && j % 2 == 0) { kp.pt.y += 2.75f; kp.octave = 0; }   // a monocular residual between chi2Mono2 and 1.5 x chi2Mono2
                    keysL.push_back(kp); uR.push_back((j % 2 && !rigScene) ? kp.pt.x - 40.f / z : -1.f); mpL.push_back(&pts[j]); jl.push_back(j);
                }
                if (rigScene && j % 3 != 1) {
                    cv::Mat Xr = Trl.rowRange(0, 3).colRange(0, 3) * Xc + Trl.rowRange(0, 3).col(3);
                    const cv::Point2f pr = fishR.project(Xr);
                    if (pr.x < 5 || pr.x > W - 5 || pr.y < 5 || pr.y > H - 5) continue;
                    cv::KeyPoint kp; kp.pt = cv::Point2f(pr.x + frand(-0.5f, 0.5f), pr.y + frand(-0.5f, 0.5f)); kp.octave = (int)(rnd() % 3);
                    if ((j * 3 + k) % 41 == 0) kp.pt.x -= 25.f;
                    keysR.push_back(kp); mpR.push_back(&pts[j]); jr.push_back(j);
                }
            }
            KeyFrame* kf = new KeyFrame(fx, fy, cx, cy, 40.f, rigScene ? (int)keysL.size() : -1, 0, 0, W, H, 64.f / W, 48.f / H, sf, invSig2);
            kf->mnId = k; kf->mpMap = &map; kf->bImu = true;
            kf->mpCamera = rigScene ? (GeometricCamera*)&fishL : (GeometricCamera*)&cam;
            if (rigScene) { kf->mpCamera2 = &fishR; kf->mTrl = Trl; }
            kf->mImuCalib.Tcb = Tcb; kf->mImuCalib.Tbc = Tbc;
            kf->mvKeys = keysL; kf->mvKeysUn = keysL; kf->mvuRight = uR; kf->mvKeysRight = keysR; kf->mvpMapPoints = mpL;
            for (MapPoint* p : mpR) kf->mvpMapPoints.push_back(p);
            for (size_t i = 0; i < jl.size(); i++) pts[jl[i]].mObservations[kf] = std::make_tuple((int)i, -1);
            for (size_t i = 0; i < jr.size(); i++) {
                auto it = pts[jr[i]].mObservations.find(kf);
                const int li = it == pts[jr[i]].mObservations.end() ? -1 : std::get<0>(it->second);
                pts[jr[i]].mObservations[kf] = std::make_tuple(li, (int)(keysL.size() + i));
            }
            kf->N = (int)kf->mvKeysUn.size();
            const bool moved = scene == 1 ? k >= 1 : k >= 2;   // the key frames that will be optimised start off the truth
            if (moved) { T.at<float>(0, 3) += frand(-0.02f, 0.02f); T.at<float>(1, 3) += frand(-0.02f, 0.02f); T.at<float>(2, 3) += frand(-0.02f, 0.02f); }
            kf->Tcw = T;
            kf->Vw = cv::Mat(3, 1, CV_32F);
            kf->Vw.at<float>(0) = 1.f + (moved ? frand(-0.05f, 0.05f) : 0.f); kf->Vw.at<float>(1) = moved ? frand(-0.05f, 0.05f) : 0.f; kf->Vw.at<float>(2) = 0.f;
            kf->mImuBias = IMU::Bias(frand(-0.01f, 0.01f), frand(-0.01f, 0.01f), frand(-0.01f, 0.01f), frand(-0.001f, 0.001f), frand(-0.001f, 0.001f), frand(-0.001f, 0.001f));
            if (k > 0) {
                kf->mPrevKF = kfs[k - 1]; kfs[k - 1]->mNextKF = kf;
                IMU::Preintegrated* pi = new IMU::Preintegrated();
                pi->dT = dtk;
                pi->dR = cv::Mat::eye(3, 3, CV_32F); pi->dV = cv::Mat(3, 1, CV_32F); pi->dP = cv::Mat(3, 1, CV_32F);
                pi->dV.at<float>(2) = 9.81f * dtk; pi->dP.at<float>(2) = 0.5f * 9.81f * dtk * dtk;   // the exact deltas of the constant-velocity truth
                pi->JRg = -(double)dtk * cv::Mat::eye(3, 3, CV_32F); pi->JVa = -(double)dtk * cv::Mat::eye(3, 3, CV_32F);
                pi->JPa = -0.5 * dtk * dtk * cv::Mat::eye(3, 3, CV_32F);
                pi->JVg = cv::Mat(3, 3, CV_32F); pi->JPg = cv::Mat(3, 3, CV_32F);
                pi->JVg.at<float>(0, 1) = 0.01f; pi->JPg.at<float>(1, 0) = -0.002f;
                pi->C = cv::Mat(15, 15, CV_32F);
                for (int q = 0; q < 15; q++) pi->C.at<float>(q, q) = q < 3 ? 1.f / 3e4f : (q < 6 ? 1.f / 2e3f : (q < 9 ? 1.f / 8e3f : (q < 12 ? 1.f / 4e5f : 1.f / 2e3f)));
                pi->C.at<float>(0, 3) = pi->C.at<float>(3, 0) = 2e-5f; pi->C.at<float>(4, 7) = pi->C.at<float>(7, 4) = -3e-5f;
                pi->C.at<float>(9, 10) = pi->C.at<float>(10, 9) = 4e-7f;
                pi->b = kfs[k - 1]->mImuBias;   // integrated with the previous key frame's bias
                kf->mpImuPreintegrated = pi;
                pre.push_back(pi);
            }
            kfs.push_back(kf);
        }
        KeyFrame* cur = kfs[NK - 1];
        // ---- by hand ----
        const int firstFree = scene == 1 ? 1 : 2;   // A: Nd = min(8 - 2, 10) = 6 -> key frames 7..2, 1 fixed (before the chain), 0 fixed (observer);
                                                    // B: Nd = 18 > chain -> 4..0 collected, 0 has no mPrevKF: popped and fixed
        orbslam3_hip::InertialBA IB2;
        liba_rig rig{};
        rig.n_cams = 1; rig.bf = 40.0; rig.model[0] = LBA_CAM_PINHOLE;
        for (int i = 0; i < 3; i++) { rig.Rcb[0][i * 4] = 1; rig.Rbc[0][i * 4] = 1; }
        rig.tcb[0][0] = (double)0.05f; rig.tbc[0][0] = (double)-0.05f;
        rig.p[0][0] = fx; rig.p[0][1] = fy; rig.p[0][2] = cx; rig.p[0][3] = cy;
        double Rrl[3][3], trl[3];
        for (int a = 0; a < 3; a++) { for (int c = 0; c < 3; c++) Rrl[a][c] = (double)Trl.at<float>(a, c); trl[a] = (double)Trl.at<float>(a, 3); }
        if (rigScene) {   // ImuCamPose's second camera (G2oTypes.cc:55-66): Rcb1 = Rrl Rcb0 = Rrl, tcb1 = Rrl tcb0 + trl, Rbc1 = Rcb1^T, tbc1 = -Rbc1 tcb1
            rig.n_cams = 2; rig.model[0] = rig.model[1] = LBA_CAM_KB8;
            for (int i = 0; i < 8; i++) { rig.p[0][i] = (double)fishL.mvParameters[i]; rig.p[1][i] = (double)fishR.mvParameters[i]; }
            for (int a = 0; a < 3; a++) {
                rig.tcb[1][a] = Rrl[a][0] * rig.tcb[0][0] + Rrl[a][1] * rig.tcb[0][1] + Rrl[a][2] * rig.tcb[0][2] + trl[a];
                for (int c = 0; c < 3; c++) { rig.Rcb[1][a * 3 + c] = Rrl[a][c]; rig.Rbc[1][c * 3 + a] = Rrl[a][c]; }
            }
            for (int a = 0; a < 3; a++) rig.tbc[1][a] = -(rig.Rbc[1][a * 3] * rig.tcb[1][0] + rig.Rbc[1][a * 3 + 1] * rig.tcb[1][1] + rig.Rbc[1][a * 3 + 2] * rig.tcb[1][2]);
        }
        IB2.setRig(rig);
        auto w = [](const cv::Mat& m, double* o) { for (int i = 0; i < m.rows * m.cols; i++) o[i] = (double)m.at<float>(i); };
        for (int k = 0; k < NK; k++) {
            double Rwb[9], twb[3], Rcw[9], tcw[3], v[3], bg[3], ba[3];
            w(kfs[k]->GetImuRotation(), Rwb); w(kfs[k]->GetImuPosition(), twb); w(kfs[k]->GetRotation(), Rcw); w(kfs[k]->GetTranslation(), tcw);
            w(kfs[k]->Vw, v); w(kfs[k]->GetGyroBias(), bg); w(kfs[k]->GetAccBias(), ba);
            double Rcw1[9], tcw1[3];
            for (int a = 0; a < 3; a++) {
                tcw1[a] = Rrl[a][0] * tcw[0] + Rrl[a][1] * tcw[1] + Rrl[a][2] * tcw[2] + trl[a];
                for (int c = 0; c < 3; c++) Rcw1[a * 3 + c] = Rrl[a][0] * Rcw[c] + Rrl[a][1] * Rcw[3 + c] + Rrl[a][2] * Rcw[6 + c];
            }
            CHECK(IB2.addKeyFrame(Rwb, twb, Rcw, tcw, rigScene ? Rcw1 : nullptr, rigScene ? tcw1 : nullptr, v, bg, ba, k < firstFree, true, k < firstFree) == k);
        }
        for (int k = NK - 1; k >= firstFree; k--) {   // newest first
            IMU::Preintegrated* pi = kfs[k]->mpImuPreintegrated;
            liba_imu_edge e{};
            e.kf1 = k - 1; e.kf2 = k; e.dT = pi->dT;
            std::memcpy(e.dR, pi->dR.data, 36); std::memcpy(e.dV, pi->dV.data, 12); std::memcpy(e.dP, pi->dP.data, 12);
            std::memcpy(e.JRg, pi->JRg.data, 36); std::memcpy(e.JVg, pi->JVg.data, 36); std::memcpy(e.JVa, pi->JVa.data, 36);
            std::memcpy(e.JPg, pi->JPg.data, 36); std::memcpy(e.JPa, pi->JPa.data, 36);
            const float bb[6] = {pi->b.bax, pi->b.bay, pi->b.baz, pi->b.bwx, pi->b.bwy, pi->b.bwz};
            std::memcpy(e.b, bb, 24);
            EdgeInertial ei(pi);
            const bool last = k == firstFree;
            for (int q = 0; q < 81; q++) e.info[q] = ei.information().v[q] * (last ? 1e-2 : 1.0);
            e.huber = (last || bRecInit) ? std::sqrt(16.92) : 0.0;
            w(pi->C.rowRange(9, 12).colRange(9, 12).inv(cv::DECOMP_SVD), e.info_g); w(pi->C.rowRange(12, 15).colRange(12, 15).inv(cv::DECOMP_SVD), e.info_a);
            CHECK(e.info[0] > 1e2 && e.info_g[0] > 1e5 && e.info_g[1] != 0.0);
            IB2.addInertial(e);
        }
        std::vector<MapPoint*> order; std::set<MapPoint*> seen;
        for (int k = NK - 1; k >= (scene == 1 ? 0 : firstFree); k--)   // B: the popped key frame's points were collected before it was popped
            for (MapPoint* p : kfs[k]->mvpMapPoints) if (p && !seen.count(p)) { seen.insert(p); order.push_back(p); }
        struct ER { KeyFrame* kf; MapPoint* mp; bool st; };
        std::vector<ER> eref;
        int nRightEdges = 0;
        for (size_t l = 0; l < order.size(); l++) {
            MapPoint* p = order[l];
            const float X[3] = {p->mWorldPos.at<float>(0), p->mWorldPos.at<float>(1), p->mWorldPos.at<float>(2)};
            CHECK(IB2.addPoint(X) == (int)l);
            for (auto& ob : p->mObservations) {
                KeyFrame* k = ob.first; const int i = std::get<0>(ob.second), ir = std::get<1>(ob.second);
                int leftOctave = 0;   // the reference weights the right-camera edge with the LEFT key point's octave (0 without a left observation)
                if (i >= 0) {
                    const cv::KeyPoint& kp = k->mvKeysUn[i];
                    leftOctave = kp.octave;
                    if (k->mvuRight[i] >= 0) IB2.addStereo((int)k->mnId, (int)l, kp.pt.x, kp.pt.y, k->mvuRight[i], invSig2[kp.octave]);
                    else IB2.addMono((int)k->mnId, (int)l, kp.pt.x, kp.pt.y, invSig2[kp.octave]);
                    eref.push_back(ER{k, p, k->mvuRight[i] >= 0});
                }
                if (ir >= 0) {
                    const cv::KeyPoint& kp = k->mvKeysRight[ir - k->NLeft];
                    IB2.addMono((int)k->mnId, (int)l, kp.pt.x, kp.pt.y, invSig2[leftOctave], 1);
                    eref.push_back(ER{k, p, false});
                    nRightEdges++;
                }
            }
        }
        double err = 0, errEnd = 0;
        const int its = IB2.optimize(bLarge ? 1e-2 : 1.0, bLarge ? 4 : 10, &err, &errEnd);
        CHECK(its >= 2 && errEnd < 0.9 * err);   // it really optimised (the gross outliers keep their Huber cost)
        int nOut = 0, nClose = 0;
        for (size_t i = 0; i < eref.size(); i++) {
            const double c2 = IB2.visualChi2((int)i);
            const bool close = eref[i].mp->mTrackDepth < 10.f;
            nClose += close && !eref[i].st && c2 > 5.991f && !(c2 > 1.5f * 5.991f);   // kept by the near-point rule only
            if (eref[i].st ? c2 > 7.815f : ((c2 > 5.991f && !close) || (c2 > 1.5f * 5.991f && close) || !IB2.depthPositive((int)i))) nOut++;
        }
        // ---- the glue ----
        Optimizer::LocalInertialBA(cur, nullptr, &map, bLarge, bRecInit);
        int erased = 0;
        for (KeyFrame* k : kfs) erased += k->nErased;
        CHECK(erased == nOut && nOut > 3 && nOut < (int)eref.size() / 5 && (scene == 1 || nClose > 0));
        for (int k = 0; k < NK; k++) {
            const liba_keyframe& r = IB2.keyFrame(k);
            CHECK(kfs[k]->mnBALocalForKF == 0 && kfs[k]->mnBAFixedForKF == 0);
            if (k < firstFree) { CHECK(kfs[k]->nPoseSets == 0 && kfs[k]->nVelSets == 0 && kfs[k]->nBiasSets == 0); continue; }
            CHECK(kfs[k]->nPoseSets == 1 && kfs[k]->nVelSets == 1 && kfs[k]->nBiasSets == 1);
            for (int a = 0; a < 3; a++) {   // identical to the flattened path, bit for bit
                CHECK(kfs[k]->Tcw.at<float>(a, 3) == (float)r.tcw[0][a]);
                for (int c = 0; c < 3; c++) CHECK(kfs[k]->Tcw.at<float>(a, c) == (float)r.Rcw[0][a * 3 + c]);
                CHECK(kfs[k]->Vw.at<float>(a) == (float)r.v[a]);
            }
            CHECK(kfs[k]->mImuBias.bwx == (float)r.bg[0] && kfs[k]->mImuBias.bwz == (float)r.bg[2] && kfs[k]->mImuBias.bax == (float)r.ba[0] && kfs[k]->mImuBias.bay == (float)r.ba[1]);
            CHECK(kfs[k]->mpImuPreintegrated->nSetNewBias == 2);   // SetNewBias(prev bias) before the edge (:4972) + KeyFrame::SetNewBias of the write-back
            CHECK(std::fabs(r.twb[0] - 0.3 * k) < 0.04 && std::fabs(r.twb[1]) < 0.04 && std::fabs(r.v[0] - 1.0) < 0.1);   // pulled towards the truth
        }
        for (size_t l = 0; l < order.size(); l++) {
            const double* X = IB2.point((int)l);
            for (int a = 0; a < 3; a++) CHECK(order[l]->mWorldPos.at<float>(a) == (float)X[a]);
            CHECK(order[l]->nNormalUpdates == 1);
        }
        CHECK(order.size() > 100 && map.mnMapChange == 1 && (nRightEdges > 150) == rigScene);
        std::printf("glue LocalInertialBA (%s): %d iterations, %zu edges, %d erased (%d kept by the near-point rule), chi2 %.1f -> %.1f\n",
                    scene == 0 ? "window inside the chain" : (scene == 1 ? "short chain, bLarge + bRecInit" : "fisheye rig"), its, eref.size(), nOut, nClose, err, errEnd);
        for (KeyFrame* k : kfs) delete k;
        for (IMU::Preintegrated* q : pre) delete q;
    }
    std::printf("glue_test OK\n");
    return 0;
}
