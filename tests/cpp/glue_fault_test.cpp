// Failure policy of the reference-signature glue (include/orbslam3_hip/GlueGuard.h): the functions integration/*.cc define replace bodies
// that never throw, on threads that catch nothing.  This program interposes ONE C-ABI entry point every adapter goes through
// (orb_memcpy_h2d: the executable's definition wins over the library's for the header-only adapters compiled into it) and makes it fail on
// demand.  Each glue function is first run normally (it must do real work), then with the fault armed: it must return quietly with the
// reference's "found nothing" value, leave the frame / map exactly as they were, and count the failure.  Also here: the LocalBundleAdjustment
// window without a single observation edge (the reference leaves through `vToErase.size() >= 0`, Optimizer.cc:2349), which is no failure.
// Built and run by tests/test_glue.py (emulated library in the CPU tier, the real liborbhip.so in the GPU tier).
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#include "ORBmatcher.h"
#include "Optimizer.h"
#include "G2oTypes.h"
#include <orbslam3_hip/GlueGuard.h>
#include <orbslam3_hip/Optimizer.h>

static bool g_fault = false;
extern "C" int orb_memcpy_h2d(void* d, const void* h, size_t n, void* stream) {
    typedef int (*fn_t)(void*, const void*, size_t, void*);
    static fn_t real = (fn_t)dlsym(RTLD_NEXT, "orb_memcpy_h2d");
    if (g_fault || !real) return ORB_E_HIP;
    return real(d, h, n, stream);
}

std::mutex ORB_SLAM3::MapPoint::mGlobalMutex;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

using namespace ORB_SLAM3;
const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;
float ORBmatcher::RadiusByViewingCos(const float& viewCos) { return viewCos > 0.998 ? 2.5 : 4.0; }
float Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv;

static uint32_t rng_state = 977u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static float frand(float a, float b) { return a + (b - a) * (float)(rnd() % 100000) / 100000.f; }

int main() {
    const int W = 480, H = 360;
    const float fx = 400.f, fy = 400.f, cx = 240.f, cy = 180.f;
    std::vector<float> sf(8), invSig2(8);
    sf[0] = 1.f; for (int i = 1; i < 8; i++) sf[i] = sf[i - 1] * 1.2f;
    for (int i = 0; i < 8; i++) invSig2[i] = 1.f / (sf[i] * sf[i]);
    Frame::mnMinX = 0; Frame::mnMaxX = W; Frame::mnMinY = 0; Frame::mnMaxY = H;
    Frame::mfGridElementWidthInv = 64.f / W; Frame::mfGridElementHeightInv = 48.f / H;
    Pinhole cam(fx, fy, cx, cy);
    const unsigned long f0 = orbslam3_hip::glue_failures();

    // ---- 1. ORBmatcher::SearchByProjection(Frame&, vpMapPoints, th): every key point has a map point projecting onto it with its descriptor ----
    {
        const int N = 300;
        Frame F;
        F.N = N; F.mvKeys.resize(N); F.mvuRight.assign(N, -1.f); F.mDescriptors = cv::Mat(N, 32, CV_8U);
        F.mvpMapPoints.assign(N, (MapPoint*)NULL); F.mvbOutlier.assign(N, false); F.mvScaleFactors = sf; F.mpCamera = &cam;
        F.mTcw = cv::Mat::eye(4, 4, CV_32F); F.mbf = 40.f; F.mb = 0.1f;
        for (int i = 0; i < N; i++) {
            F.mvKeys[i].pt = cv::Point2f(frand(20.f, W - 20.f), frand(20.f, H - 20.f)); F.mvKeys[i].octave = (int)(rnd() % 4); F.mvKeys[i].angle = frand(0.f, 359.f);
            for (int b = 0; b < 32; b++) F.mDescriptors.data[i * 32 + b] = (uint8_t)rnd();
        }
        F.mvKeysUn = F.mvKeys;
        std::vector<MapPoint> mps(N);
        std::vector<MapPoint*> vp(N);
        for (int i = 0; i < N; i++) {
            MapPoint& p = mps[i];
            p.mDescriptor = cv::Mat(1, 32, CV_8U); std::memcpy(p.mDescriptor.data, F.mDescriptors.data + i * 32, 32);
            p.nObs = 3; p.mbTrackInView = true; p.mTrackProjX = F.mvKeys[i].pt.x; p.mTrackProjY = F.mvKeys[i].pt.y; p.mTrackProjXR = -1.f; p.mTrackDepth = 5.f;
            p.mnTrackScaleLevel = F.mvKeys[i].octave; p.mTrackViewCos = 0.9f;
            vp[i] = &p;
        }
        ORBmatcher m(0.8f, true);
        Frame G = F;
        CHECK(m.SearchByProjection(G, vp, 1.f) > N / 2);                     // the un-faulted call matches
        g_fault = true;
        Frame Q = F;
        const int n = m.SearchByProjection(Q, vp, 1.f);
        g_fault = false;
        CHECK(n == 0);
        for (int i = 0; i < N; i++) CHECK(Q.mvpMapPoints[i] == NULL);        // nothing was scattered
        CHECK(orbslam3_hip::glue_failures() == f0 + 1);
        // motion-model overload on the same data
        Frame Last = F;
        for (int i = 0; i < N; i++) {
            mps[i].mWorldPos = cv::Mat(3, 1, CV_32F);
            const float z = 4.f;
            mps[i].mWorldPos.at<float>(0) = (F.mvKeys[i].pt.x - cx) / fx * z; mps[i].mWorldPos.at<float>(1) = (F.mvKeys[i].pt.y - cy) / fy * z; mps[i].mWorldPos.at<float>(2) = z;
        }
        Last.mvpMapPoints = vp;
        Frame C1 = F, C2 = F;
        CHECK(m.SearchByProjection(C1, Last, 7.f, true) > N / 2);
        g_fault = true;
        CHECK(m.SearchByProjection(C2, Last, 7.f, true) == 0);
        g_fault = false;
        for (int i = 0; i < N; i++) CHECK(C2.mvpMapPoints[i] == NULL);
        CHECK(orbslam3_hip::glue_failures() == f0 + 2);
        std::printf("glue fault: SearchByProjection x2 dropped quietly, frame untouched\n");
    }

    // ---- 2. Optimizer::LocalBundleAdjustment: a faulted call moves nothing; a window without edges returns through the reference's own exit ----
    {
        Map map; map.mnInitKFid = 0;
        const int NK = 6, NP = 120;
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
            T.at<float>(0, 3) = -0.15f * k;
            for (int j = 0; j < NP; j++) {
                cv::Mat Xc = T.rowRange(0, 3).colRange(0, 3) * Xtrue[j] + T.rowRange(0, 3).col(3);
                const float z = Xc.at<float>(2), u = fx * Xc.at<float>(0) / z + cx, v = fy * Xc.at<float>(1) / z + cy;
                if (u < 5 || u > W - 5 || v < 5 || v > H - 5) continue;
                cv::KeyPoint kp; kp.pt = cv::Point2f(u + frand(-0.5f, 0.5f), v + frand(-0.5f, 0.5f)); kp.octave = (int)(rnd() % 4);
                const int idx = (int)kf->mvKeysUn.size();
                kf->mvKeysUn.push_back(kp); kf->mvKeys.push_back(kp); kf->mvuRight.push_back(-1.f); kf->mvpMapPoints.push_back(&pts[j]);
                pts[j].mObservations[kf] = std::make_tuple(idx, -1);
            }
            kf->N = (int)kf->mvKeysUn.size();
            if (k > 0) T.at<float>(0, 3) += frand(-0.02f, 0.02f);
            kf->Tcw = T;
            kfs.push_back(kf);
        }
        KeyFrame* cur = kfs[NK - 1];
        for (int k = NK - 2; k >= 2; k--) cur->mvpOrderedConnectedKeyFrames.push_back(kfs[k]);
        std::vector<cv::Mat> T0, X0;
        for (KeyFrame* k : kfs) T0.push_back(k->Tcw.clone());
        for (MapPoint& p : pts) X0.push_back(p.mWorldPos.clone());
        int numFixed = -1; bool stop = false;
        const unsigned long fb = orbslam3_hip::glue_failures();
        g_fault = true;
        Optimizer::LocalBundleAdjustment(cur, &stop, &map, numFixed);      // must not throw
        g_fault = false;
        CHECK(orbslam3_hip::glue_failures() == fb + 1);
        CHECK(map.mnMapChange == 0);
        for (int k = 0; k < NK; k++) { CHECK(kfs[k]->nPoseSets == 0 && kfs[k]->nErased == 0); CHECK(std::memcmp(kfs[k]->Tcw.data, T0[k].data, 64) == 0); }
        for (int j = 0; j < NP; j++) { CHECK(pts[j].nNormalUpdates == 0 && pts[j].nErased == 0); CHECK(std::memcmp(pts[j].mWorldPos.data, X0[j].data, 12) == 0); }
        // (the window-selection stamps are keyed by the key frame id, which is new for every call in a running system)
        for (MapPoint& p : pts) p.mnBALocalForKF = 0;
        for (KeyFrame* k : kfs) k->mnBALocalForKF = k->mnBAFixedForKF = 0;
        Optimizer::LocalBundleAdjustment(cur, &stop, &map, numFixed);      // and the same map optimises once the device answers again
        CHECK(orbslam3_hip::glue_failures() == fb + 1 && map.mnMapChange == 1 && kfs[NK - 1]->nPoseSets == 1);
        // a window whose map points are only observed by bad key frames: local points exist, no edge does
        Map map2; map2.mnInitKFid = 100;
        KeyFrame lone(fx, fy, cx, cy, 40.f, -1, 0, 0, W, H, 64.f / W, 48.f / H, sf, invSig2), bad(fx, fy, cx, cy, 40.f, -1, 0, 0, W, H, 64.f / W, 48.f / H, sf, invSig2);
        lone.mnId = 7; lone.mpMap = &map2; lone.mpCamera = &cam; lone.Tcw = cv::Mat::eye(4, 4, CV_32F);
        bad.mnId = 3; bad.mpMap = &map2; bad.mpCamera = &cam; bad.Tcw = cv::Mat::eye(4, 4, CV_32F); bad.mbBad = true;
        std::vector<MapPoint> orphans(5);
        for (int j = 0; j < 5; j++) {
            orphans[j].mnId = 1000 + j; orphans[j].mpMap = &map2; orphans[j].mWorldPos = Xtrue[j].clone();
            cv::KeyPoint kp; kp.pt = cv::Point2f(100.f + j, 100.f); kp.octave = 0;
            bad.mvKeysUn.push_back(kp); bad.mvKeys.push_back(kp); bad.mvuRight.push_back(-1.f); bad.mvpMapPoints.push_back(&orphans[j]);
            orphans[j].mObservations[&bad] = std::make_tuple(j, -1);
            lone.mvpMapPoints.push_back(&orphans[j]);   // the current key frame lists them but holds no observation record (e.g. just culled)
        }
        const unsigned long fc = orbslam3_hip::glue_failures();
        Optimizer::LocalBundleAdjustment(&lone, &stop, &map2, numFixed);
        CHECK(orbslam3_hip::glue_failures() == fc);                        // not a failure: the reference's quiet exit
        CHECK(map2.mnMapChange == 0 && lone.nPoseSets == 0);
        for (MapPoint& p : orphans) CHECK(p.nNormalUpdates == 0);
        std::printf("glue fault: LocalBundleAdjustment dropped quietly, map untouched; empty window leaves through the reference's exit\n");
        for (KeyFrame* k : kfs) delete k;
    }

    // ---- 3. Optimizer::PoseOptimization(Frame*): a faulted call returns 0 and leaves pose and outlier flags alone ----
    {
        Frame F;
        F.mpCamera = &cam; F.mbf = 40.f; F.fx = fx; F.fy = fy; F.cx = cx; F.cy = cy;
        F.N = 200;
        F.mvInvLevelSigma2 = invSig2;
        F.mvKeysUn.resize(F.N); F.mvuRight.assign(F.N, -1.f); F.mvpMapPoints.assign(F.N, (MapPoint*)NULL); F.mvbOutlier.assign(F.N, true);
        std::vector<MapPoint> mps(F.N);
        for (int i = 0; i < F.N; i++) {
            const float X[3] = {frand(-3.f, 3.f), frand(-2.f, 2.f), frand(4.f, 9.f)};
            F.mvKeysUn[i].pt = cv::Point2f(fx * X[0] / X[2] + cx + frand(-0.5f, 0.5f), fy * X[1] / X[2] + cy + frand(-0.5f, 0.5f)); F.mvKeysUn[i].octave = (int)(rnd() % 8);
            mps[i].mWorldPos = cv::Mat(3, 1, CV_32F);
            for (int r = 0; r < 3; r++) mps[i].mWorldPos.at<float>(r) = X[r];
            F.mvpMapPoints[i] = &mps[i];
        }
        F.mTcw = cv::Mat::eye(4, 4, CV_32F);
        F.mTcw.at<float>(0, 3) = 0.04f;
        const cv::Mat T0 = F.mTcw.clone();
        const unsigned long fb = orbslam3_hip::glue_failures();
        g_fault = true;
        const int good = Optimizer::PoseOptimization(&F);
        g_fault = false;
        CHECK(good == 0 && orbslam3_hip::glue_failures() == fb + 1);
        CHECK(std::memcmp(F.mTcw.data, T0.data, 64) == 0);
        for (int i = 0; i < F.N; i++) CHECK(F.mvbOutlier[i] == true);
        CHECK(Optimizer::PoseOptimization(&F) > 150);                        // the same frame optimises once the device answers again
        CHECK(std::fabs(F.mTcw.at<float>(0, 3)) < 0.01f);
        std::printf("glue fault: PoseOptimization dropped quietly, frame untouched\n");
    }
    std::printf("glue_fault_test OK (%lu faults injected)\n", orbslam3_hip::glue_failures() - f0);
    return 0;
}
