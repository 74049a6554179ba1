// integration/Optimizer_hip.cc — Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*, int&) (reference include/Optimizer.h:58,
// src/Optimizer.cc:1811-2523), Optimizer::PoseOptimization(Frame*) (include/Optimizer.h:53, src/Optimizer.cc:907-1273) and
// Optimizer::LocalInertialBA(KeyFrame*, bool*, Map*, bool, bool) (include/Optimizer.h:99, src/Optimizer.cc:4753-5365) over liborbhip.so.
//
// Drop-in for the same-named function of src/Optimizer.cc: compile this file INSIDE the ORB-SLAM3 tree instead of that body
// (integration/README.md), with -DORBHIP_WITH_ORBSLAM3.  LocalMapping calls it unchanged (LocalMapping.cc:236).
//   * window selection (:1816-1945), the outlier passes (:2229-2283, :2300-2344), the erase loop (:2375-2401) and the write-back
//     (:2425-2515) are the reference's statements, kept verbatim;
//   * the g2o graph build (:1957-2190) becomes calls on orbslam3_hip::LbaLinearizer in the same order (vertices first, then the edges
//     landmark-major in GetObservations() order); optimizer.optimize(5) / optimize(10) (:2205, :2290) are LbaLinearizer::optimize — the
//     Levenberg-Marquardt loop with Schur complement and dense Cholesky on the device, polling pbStopFlag between lambda trials like g2o's
//     terminate(); e->chi2() / e->isDepthPositive() come from LbaLinearizer::computeErrors.
// Not taken over: the Verbose / file-dump diagnostics of the reference (bRedrawError is dead code there: the function returns before it).
#ifdef ORBHIP_WITH_ORBSLAM3
#include "Optimizer.h"
#include "G2oTypes.h"

#include <algorithm>
#include <cmath>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include <orbslam3_hip/GlueGuard.h>
#include <orbslam3_hip/Optimizer.h>

namespace ORB_SLAM3 {

namespace {
// GeometricCamera -> lba_camera: Pinhole (CameraModels/Pinhole.cpp) or KannalaBrandt8 parameters widened to double; mTrl for the body edges
lba_camera make_camera(GeometricCamera* cam, double bf, const cv::Mat& Trl) {
    lba_camera c{};
    c.model = cam->GetType() == cam->CAM_FISHEYE ? LBA_CAM_KB8 : LBA_CAM_PINHOLE;
    for (size_t i = 0; i < cam->size() && i < 8; i++) c.p[i] = (double)cam->getParameter((int)i);
    c.bf = bf;
    c.trl_q[3] = 1.0;
    if (!Trl.empty()) {   // Converter::toSE3Quat(pKFi->mTrl) (:2168)
        double p7[7];
        orbslam3_hip::LbaLinearizer::poseFromTcw(Trl.ptr<float>(), Trl.cols, p7);
        for (int i = 0; i < 3; i++) c.trl_t[i] = p7[i];
        for (int i = 0; i < 4; i++) c.trl_q[i] = p7[3 + i];
    }
    return c;
}
// Converter::toCvMat(g2o::SE3Quat) (Converter.cc:60-65 via to_homogeneous_matrix): 4x4 CV_32F from (t, q)
cv::Mat pose_to_cvmat(const double* p7) {
    const double x = p7[3], y = p7[4], z = p7[5], w = p7[6];
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    cv::Mat T = cv::Mat::eye(4, 4, CV_32F);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T.at<float>(r, c) = (float)R[r * 3 + c]; T.at<float>(r, 3) = (float)p7[r]; }
    return T;
}
}  // namespace

void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF) try {
    // ---- Local KeyFrames: First Breath Search from Current Keyframe (:1813-1829) ----
    std::list<KeyFrame*> lLocalKeyFrames;
    lLocalKeyFrames.push_back(pKF);
    pKF->mnBALocalForKF = pKF->mnId;
    Map* pCurrentMap = pKF->GetMap();
    const std::vector<KeyFrame*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
    for (int i = 0, iend = vNeighKFs.size(); i < iend; i++) {
        KeyFrame* pKFi = vNeighKFs[i];
        pKFi->mnBALocalForKF = pKF->mnId;
        if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lLocalKeyFrames.push_back(pKFi);
    }
    // ---- Local MapPoints seen in Local KeyFrames (:1831-1862) ----
    num_fixedKF = 0;
    std::list<MapPoint*> lLocalMapPoints;
    for (std::list<KeyFrame*>::iterator lit = lLocalKeyFrames.begin(), lend = lLocalKeyFrames.end(); lit != lend; lit++) {
        KeyFrame* pKFi = *lit;
        if (pKFi->mnId == pMap->GetInitKFid()) num_fixedKF = 1;
        std::vector<MapPoint*> vpMPs = pKFi->GetMapPointMatches();
        for (std::vector<MapPoint*>::iterator vit = vpMPs.begin(), vend = vpMPs.end(); vit != vend; vit++) {
            MapPoint* pMP = *vit;
            if (pMP)
                if (!pMP->isBad() && pMP->GetMap() == pCurrentMap)
                    if (pMP->mnBALocalForKF != pKF->mnId) { lLocalMapPoints.push_back(pMP); pMP->mnBALocalForKF = pKF->mnId; }
        }
    }
    // ---- Fixed Keyframes: see Local MapPoints but are not Local Keyframes (:1864-1882) ----
    std::list<KeyFrame*> lFixedCameras;
    for (std::list<MapPoint*>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++) {
        std::map<KeyFrame*, std::tuple<int, int>> observations = (*lit)->GetObservations();
        for (std::map<KeyFrame*, std::tuple<int, int>>::iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
            KeyFrame* pKFi = mit->first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                pKFi->mnBAFixedForKF = pKF->mnId;
                if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lFixedCameras.push_back(pKFi);
            }
        }
    }
    num_fixedKF = lFixedCameras.size() + num_fixedKF;
    if (num_fixedKF < 2) {   // :1885-1931
        std::list<KeyFrame*>::iterator lit = lLocalKeyFrames.begin();
        int lowerId = pKF->mnId;
        KeyFrame* pLowerKf = NULL;
        int secondLowerId = pKF->mnId;
        KeyFrame* pSecondLowerKF = NULL;
        for (; lit != lLocalKeyFrames.end(); lit++) {
            KeyFrame* pKFi = *lit;
            if (pKFi == pKF || pKFi->mnId == pMap->GetInitKFid()) continue;
            if ((int)pKFi->mnId < lowerId) { lowerId = pKFi->mnId; pLowerKf = pKFi; }
            else if ((int)pKFi->mnId < secondLowerId) { secondLowerId = pKFi->mnId; pSecondLowerKF = pKFi; }
        }
        if (pLowerKf) {   // (the reference dereferences an uninitialised pointer when no candidate exists; a window that small is not optimised there either)
            lFixedCameras.push_back(pLowerKf);
            lLocalKeyFrames.remove(pLowerKf);
            num_fixedKF++;
            if (num_fixedKF < 2 && pSecondLowerKF) {
                lFixedCameras.push_back(pSecondLowerKF);
                lLocalKeyFrames.remove(pSecondLowerKF);
                num_fixedKF++;
            }
        }
    }

    // ---- the graph (:1957-2190) as the flattened window of orbslam3_hip::LbaLinearizer ----
    orbslam3_hip::LbaLinearizer L;
    // vertex ids are the key frames' mnId and g2o orders the Hessian blocks by id: poses are added in ascending mnId
    std::map<long unsigned int, std::pair<KeyFrame*, bool>> kfById;   // id -> (key frame, fixed)
    for (KeyFrame* pKFi : lLocalKeyFrames) kfById[pKFi->mnId] = std::make_pair(pKFi, pKFi->mnId == pMap->GetInitKFid());   // :1982-1990
    for (KeyFrame* pKFi : lFixedCameras) kfById[pKFi->mnId] = std::make_pair(pKFi, true);                                  // :1996-2005
    std::map<KeyFrame*, int> poseIdx, camLeft, camRight;
    std::map<std::pair<GeometricCamera*, float>, int> camIds;   // one lba_camera per distinct (camera model object, bf)
    for (auto& kv : kfById) {
        KeyFrame* pKFi = kv.second.first;
        const cv::Mat Tcw = pKFi->GetPose();
        double p7[7];
        orbslam3_hip::LbaLinearizer::poseFromTcw(Tcw.ptr<float>(), Tcw.cols, p7);   // Converter::toSE3Quat
        poseIdx[pKFi] = L.addPose(p7, kv.second.second);
        const std::pair<GeometricCamera*, float> kl(pKFi->mpCamera, pKFi->mbf);
        if (!camIds.count(kl)) camIds[kl] = L.addCamera(make_camera(pKFi->mpCamera, (double)pKFi->mbf, cv::Mat()));
        camLeft[pKFi] = camIds[kl];
        if (pKFi->mpCamera2) {
            const std::pair<GeometricCamera*, float> kr(pKFi->mpCamera2, 0.f);
            if (!camIds.count(kr)) camIds[kr] = L.addCamera(make_camera(pKFi->mpCamera2, 0.0, pKFi->mTrl));
            camRight[pKFi] = camIds[kr];
        }
    }
    struct EdgeRef { KeyFrame* kf; MapPoint* mp; bool stereo; };
    std::vector<EdgeRef> edges;   // in LbaLinearizer edge order
    std::map<MapPoint*, int> pointIdx;
    for (std::list<MapPoint*>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++) {
        MapPoint* pMP = *lit;
        const cv::Mat Xw = pMP->GetWorldPos();
        const double X[3] = {(double)Xw.at<float>(0), (double)Xw.at<float>(1), (double)Xw.at<float>(2)};   // Converter::toVector3d
        const int l = L.addPoint(X);
        pointIdx[pMP] = l;
        const std::map<KeyFrame*, std::tuple<int, int>> observations = pMP->GetObservations();
        for (std::map<KeyFrame*, std::tuple<int, int>>::const_iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
            KeyFrame* pKFi = mit->first;
            if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;
            const int leftIndex = std::get<0>(mit->second);
            if (leftIndex != -1 && pKFi->mvuRight[leftIndex] < 0) {   // Monocular observation (:2088-2113)
                const cv::KeyPoint& kpUn = pKFi->mvKeysUn[leftIndex];
                L.addEdge(poseIdx[pKFi], l, LBA_EDGE_MONO, camLeft[pKFi], kpUn.pt.x, kpUn.pt.y, 0.f, pKFi->mvInvLevelSigma2[kpUn.octave]);
                edges.push_back(EdgeRef{pKFi, pMP, false});
            } else if (leftIndex != -1 && pKFi->mvuRight[leftIndex] >= 0) {   // Stereo observation (:2114-2147)
                const cv::KeyPoint& kpUn = pKFi->mvKeysUn[leftIndex];
                L.addEdge(poseIdx[pKFi], l, LBA_EDGE_STEREO, camLeft[pKFi], kpUn.pt.x, kpUn.pt.y, pKFi->mvuRight[leftIndex], pKFi->mvInvLevelSigma2[kpUn.octave]);
                edges.push_back(EdgeRef{pKFi, pMP, true});
            }
            if (pKFi->mpCamera2) {   // right camera of a rig (:2149-2186)
                int rightIndex = std::get<1>(mit->second);
                if (rightIndex != -1) {
                    rightIndex -= pKFi->NLeft;
                    const cv::KeyPoint kp = pKFi->mvKeysRight[rightIndex];
                    L.addEdge(poseIdx[pKFi], l, LBA_EDGE_BODY, camRight[pKFi], kp.pt.x, kp.pt.y, 0.f, pKFi->mvInvLevelSigma2[kp.octave]);
                    edges.push_back(EdgeRef{pKFi, pMP, false});
                }
            }
        }
    }
    // (the Huber deltas are LbaLinearizer's: sqrt(5.991) / sqrt(7.815) as floats, :2052-2053)

    if (pbStopFlag)
        if (*pbStopFlag) return;
    // a window without a single observation edge: the reference's optimize() calls do nothing on the empty graph and the function then leaves
    // through `vToErase.size() >= 0` (:2349-2352) with the map untouched
    if (edges.empty()) return;

    L.optimize(5, nullptr, nullptr, pbStopFlag);   // optimizer.optimize(5) :2205; pbStopFlag is polled between lambda trials like g2o's terminate()

    bool bDoMore = true;
    if (pbStopFlag)
        if (*pbStopFlag) bDoMore = false;
    if (bDoMore) {
        // the reference's first outlier pass only COUNTS bad observations (:2214-2283: no setLevel, no kernel change), then re-optimises
        L.optimize(10, nullptr, nullptr, pbStopFlag);   // optimizer.initializeOptimization(0); optimizer.optimize(10) :2289-2290
    }

    // ---- outlier edges (:2293-2344): e->chi2() > 5.991 (mono, body) / 7.815 (stereo) or !e->isDepthPositive() ----
    orbslam3_hip::LbaHostSystem S;
    L.computeErrors(S);
    std::vector<std::pair<KeyFrame*, MapPoint*>> vToErase;
    vToErase.reserve(edges.size());
    size_t nMonoStereo = 0;
    for (int pass = 0; pass < 3; pass++)   // the reference walks vpEdgesMono, then vpEdgesBody, then vpEdgesStereo
        for (size_t i = 0; i < edges.size(); i++) {
            const int kind = L.edgeKind((int)i);
            if ((pass == 0 && kind != LBA_EDGE_MONO) || (pass == 1 && kind != LBA_EDGE_BODY) || (pass == 2 && kind != LBA_EDGE_STEREO)) continue;
            if (pass != 1) nMonoStereo++;
            MapPoint* pMP = edges[i].mp;
            if (pMP->isBad()) continue;
            if (S.chi2[i] > (edges[i].stereo ? 7.815 : 5.991) || !(S.depth[i] > 0.0)) vToErase.push_back(std::make_pair(edges[i].kf, pMP));
        }
    if (vToErase.size() >= nMonoStereo * 0.5) return;   // "MOST OF THE POINTS HAS BECOME OUTLIERS" :2349-2352

    // Get Map Mutex
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
    if (!vToErase.empty())
        for (size_t i = 0; i < vToErase.size(); i++) {   // :2390-2401
            KeyFrame* pKFi = vToErase[i].first;
            MapPoint* pMPi = vToErase[i].second;
            pKFi->EraseMapPointMatch(pMPi);
            pMPi->EraseObservation(pKFi);
        }
    // Recover optimized data: Keyframes (:2425-2433), then Points (:2499-2506)
    for (std::list<KeyFrame*>::iterator lit = lLocalKeyFrames.begin(), lend = lLocalKeyFrames.end(); lit != lend; lit++) {
        KeyFrame* pKFi = *lit;
        pKFi->SetPose(pose_to_cvmat(L.pose(poseIdx[pKFi])));
    }
    for (std::list<MapPoint*>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++) {
        MapPoint* pMP = *lit;
        const double* X = L.point(pointIdx[pMP]);
        cv::Mat Xw(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) Xw.at<float>(i) = (float)X[i];   // Converter::toCvMat(Eigen::Vector3d)
        pMP->SetWorldPos(Xw);
        pMP->UpdateNormalAndDepth();
    }
    pMap->IncreaseChangeIndex();
} ORBHIP_GLUE_CATCH("Optimizer::LocalBundleAdjustment", return;)

// Optimizer::PoseOptimization(Frame*) (reference include/Optimizer.h:53, src/Optimizer.cc:907-1273; called after every matcher call in Tracking:
// Tracking.cc:2210, 2395, 2468).  The observation walk (:966-1127) is the reference's, one PoseOptimizer observation per g2o edge in the same
// order; the four optimise / classify rounds (:1133-1252) are the single call PoseOptimizer::optimize (one launch on the device); pose
// recovery and return value as :1255-1270.
int Optimizer::PoseOptimization(Frame* pFrame) try {
    thread_local orbslam3_hip::PoseOptimizer PO;     // device buffers are reused from frame to frame
    PO.reset();
    int nInitialCorrespondences = 0;
    const int N = pFrame->N;
    std::vector<size_t> vnIndexEdge;                 // feature index of every observation, in edge order
    vnIndexEdge.reserve(N);
    const int camL = PO.addCamera(make_camera(pFrame->mpCamera, (double)pFrame->mbf, cv::Mat()));
    const int camR = pFrame->mpCamera2 ? PO.addCamera(make_camera(pFrame->mpCamera2, (double)pFrame->mbf, pFrame->mTrl)) : -1;
    {
        std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);
        for (int i = 0; i < N; i++) {
            MapPoint* pMP = pFrame->mvpMapPoints[i];
            if (!pMP) continue;
            const cv::Mat Xwm = pMP->GetWorldPos();
            const float Xw[3] = {Xwm.at<float>(0), Xwm.at<float>(1), Xwm.at<float>(2)};
            if (!pFrame->mpCamera2) {                // Conventional SLAM (:971-1050)
                nInitialCorrespondences++;
                const cv::KeyPoint& kpUn = pFrame->mvKeysUn[i];
                const float invSigma2 = pFrame->mvInvLevelSigma2[kpUn.octave];
                if (pFrame->mvuRight[i] < 0) PO.addMono(Xw, kpUn.pt.x, kpUn.pt.y, invSigma2, camL);                      // :979-1011
                else PO.addStereo(Xw, kpUn.pt.x, kpUn.pt.y, pFrame->mvuRight[i], invSigma2, camL);                        // :1013-1049
            } else {                                 // fisheye rig (:1052-1125): left camera on mvKeys, right camera through mTrl
                nInitialCorrespondences++;
                if (i < pFrame->Nleft) {
                    const cv::KeyPoint kpUn = pFrame->mvKeys[i];
                    PO.addMono(Xw, kpUn.pt.x, kpUn.pt.y, pFrame->mvInvLevelSigma2[kpUn.octave], camL);
                } else {
                    const cv::KeyPoint kpUn = pFrame->mvKeysRight[i - pFrame->Nleft];
                    PO.addBody(Xw, kpUn.pt.x, kpUn.pt.y, pFrame->mvInvLevelSigma2[kpUn.octave], camR);
                }
            }
            vnIndexEdge.push_back((size_t)i);
        }
    }
    // pFrame->mvbOutlier[i] = false of the graph build (:977, :1057) is applied after the device call (the classification overwrites every
    // one of these entries anyway), so that a failing call leaves the frame as it was
    if (nInitialCorrespondences < 3) {               // :1130-1131
        for (size_t k = 0; k < vnIndexEdge.size(); k++) pFrame->mvbOutlier[vnIndexEdge[k]] = false;
        return 0;
    }
    double p7[7];
    orbslam3_hip::LbaLinearizer::poseFromTcw(pFrame->mTcw.ptr<float>(), pFrame->mTcw.cols, p7);   // Converter::toSE3Quat(pFrame->mTcw)
    std::vector<bool> outl;
    const int nGood = PO.optimize(p7, outl);         // 4 x optimizer.optimize(10) + chi2 classification, Huber dropped for the last round
    for (size_t k = 0; k < vnIndexEdge.size(); k++) pFrame->mvbOutlier[vnIndexEdge[k]] = outl[k];
    pFrame->SetPose(pose_to_cvmat(p7));              // :1255-1258
    return nGood;                                    // nInitialCorrespondences - nBad
} ORBHIP_GLUE_CATCH("Optimizer::PoseOptimization", return 0;)

// Optimizer::LocalInertialBA(KeyFrame*, bool*, Map*, bool bLarge, bool bRecInit) (reference include/Optimizer.h:99, src/Optimizer.cc:4753-5365;
// LocalMapping.cc:196 calls it in the inertial modes instead of LocalBundleAdjustment).
//   * the temporal window (:4769-4787), its map points (:4791-4811), the fixed key frames (:4813-4876), the outlier pass (:5237-5275), the
//     rejection test (:5280-5285), the erase loop and the write-back (:5289-5350) are the reference's statements;
//   * the g2o graph (:4899-5215) becomes one window of orbslam3_hip::InertialBA: key frames in ascending mnId (g2o's Hessian block order), one
//     liba_imu_edge per EdgeInertial + EdgeGyroRW + EdgeAccRW triple in the reference's insertion order, visual edges landmark-major in
//     GetObservations() order; optimizer.optimize(opt_it) with setUserLambdaInit is InertialBA::optimize (one launch on the device).
//   * EdgeInertial's information matrix is taken from the reference's own EdgeInertial constructor (G2oTypes.cc:706-725), so the eigenvalue
//     clean-up is the reference's code, not a restatement.
// pbStopFlag: the reference hands it to the optimizer only AFTER optimize() returned (:5228-5229), so it has no effect there either.
void Optimizer::LocalInertialBA(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, bool bLarge, bool bRecInit) try {
    (void)pbStopFlag;
    Map* pCurrentMap = pKF->GetMap();
    int maxOpt = 10, opt_it = 10;
    if (bLarge) { maxOpt = 25; opt_it = 4; }
    const int Nd = std::min((int)pCurrentMap->KeyFramesInMap() - 2, maxOpt);
    // ---- optimisable key frames: the temporal chain behind pKF (:4769-4787) ----
    std::vector<KeyFrame*> vpOptimizableKFs;
    const std::vector<KeyFrame*> vpNeighsKFs = pKF->GetVectorCovisibleKeyFrames();
    std::list<KeyFrame*> lpOptVisKFs;
    vpOptimizableKFs.reserve(Nd);
    vpOptimizableKFs.push_back(pKF);
    pKF->mnBALocalForKF = pKF->mnId;
    for (int i = 1; i < Nd; i++) {
        if (vpOptimizableKFs.back()->mPrevKF) {
            vpOptimizableKFs.push_back(vpOptimizableKFs.back()->mPrevKF);
            vpOptimizableKFs.back()->mnBALocalForKF = pKF->mnId;
        } else
            break;
    }
    int N = vpOptimizableKFs.size();
    // ---- their map points (:4791-4811) ----
    std::list<MapPoint*> lLocalMapPoints;
    for (int i = 0; i < N; i++) {
        std::vector<MapPoint*> vpMPs = vpOptimizableKFs[i]->GetMapPointMatches();
        for (std::vector<MapPoint*>::iterator vit = vpMPs.begin(), vend = vpMPs.end(); vit != vend; vit++) {
            MapPoint* pMP = *vit;
            if (pMP)
                if (!pMP->isBad())
                    if (pMP->mnBALocalForKF != pKF->mnId) { lLocalMapPoints.push_back(pMP); pMP->mnBALocalForKF = pKF->mnId; }
        }
    }
    // ---- the key frame before the chain is fixed; without one the oldest of the chain is (:4813-4826) ----
    std::list<KeyFrame*> lFixedKeyFrames;
    if (vpOptimizableKFs.back()->mPrevKF) {
        lFixedKeyFrames.push_back(vpOptimizableKFs.back()->mPrevKF);
        vpOptimizableKFs.back()->mPrevKF->mnBAFixedForKF = pKF->mnId;
    } else {
        vpOptimizableKFs.back()->mnBALocalForKF = 0;
        vpOptimizableKFs.back()->mnBAFixedForKF = pKF->mnId;
        lFixedKeyFrames.push_back(vpOptimizableKFs.back());
        vpOptimizableKFs.pop_back();
    }
    // ---- optimisable covisible key frames (maxCovKF = 0 in the reference: the list stays empty, :4829-4860) ----
    const size_t maxCovKF = 0;
    for (int i = 0, iend = vpNeighsKFs.size(); i < iend; i++) {
        if (lpOptVisKFs.size() >= maxCovKF) break;
        KeyFrame* pKFi = vpNeighsKFs[i];
        if (pKFi->mnBALocalForKF == pKF->mnId || pKFi->mnBAFixedForKF == pKF->mnId) continue;
        pKFi->mnBALocalForKF = pKF->mnId;
        if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) {
            lpOptVisKFs.push_back(pKFi);
            std::vector<MapPoint*> vpMPs = pKFi->GetMapPointMatches();
            for (std::vector<MapPoint*>::iterator vit = vpMPs.begin(), vend = vpMPs.end(); vit != vend; vit++) {
                MapPoint* pMP = *vit;
                if (pMP)
                    if (!pMP->isBad())
                        if (pMP->mnBALocalForKF != pKF->mnId) { lLocalMapPoints.push_back(pMP); pMP->mnBALocalForKF = pKF->mnId; }
            }
        }
    }
    // ---- fixed key frames: the first unmarked observer of each local map point, at most 200 (:4863-4882) ----
    const size_t maxFixKF = 200;
    for (std::list<MapPoint*>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++) {
        std::map<KeyFrame*, std::tuple<int, int>> observations = (*lit)->GetObservations();
        for (std::map<KeyFrame*, std::tuple<int, int>>::iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
            KeyFrame* pKFi = mit->first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                pKFi->mnBAFixedForKF = pKF->mnId;
                if (!pKFi->isBad()) { lFixedKeyFrames.push_back(pKFi); break; }
            }
        }
        if (lFixedKeyFrames.size() >= maxFixKF) break;
    }
    N = vpOptimizableKFs.size();

    // ---- the graph (:4899-5215) as one window of orbslam3_hip::InertialBA ----
    thread_local orbslam3_hip::InertialBA IB;   // device buffers are reused from call to call
    IB.clear();
    auto widen = [](const cv::Mat& m, double* out) { for (int r = 0; r < m.rows; r++) for (int c = 0; c < m.cols; c++) out[r * m.cols + c] = (double)m.at<float>(r, c); };
    auto mul33 = [](const double* A, const double* B, double* C) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c]; };
    auto mul31 = [](const double* A, const double* x, const double* t, double* y) { for (int r = 0; r < 3; r++) y[r] = A[r * 3] * x[0] + A[r * 3 + 1] * x[1] + A[r * 3 + 2] * x[2] + (t ? t[r] : 0.0); };
    double Rrl[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, trl[3] = {0, 0, 0};
    const bool bRig = pKF->mpCamera2 != nullptr;
    if (bRig) { widen(pKF->mTrl.rowRange(0, 3).colRange(0, 3), Rrl); widen(pKF->mTrl.rowRange(0, 3).col(3), trl); }   // Converter::toMatrix4d(pKF->mTrl)
    {   // the calibration members of ImuCamPose (G2oTypes.cc:44-66): the same for every key frame of the map
        liba_rig rig{};
        rig.n_cams = bRig ? 2 : 1;
        widen(pKF->mImuCalib.Tcb.rowRange(0, 3).colRange(0, 3), rig.Rcb[0]);
        widen(pKF->mImuCalib.Tcb.rowRange(0, 3).col(3), rig.tcb[0]);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rig.Rbc[0][r * 3 + c] = rig.Rcb[0][c * 3 + r];
        widen(pKF->mImuCalib.Tbc.rowRange(0, 3).col(3), rig.tbc[0]);
        rig.bf = (double)pKF->mbf;
        GeometricCamera* cams[2] = {pKF->mpCamera, pKF->mpCamera2};
        for (int k = 0; k < rig.n_cams; k++) {
            rig.model[k] = cams[k]->GetType() == cams[k]->CAM_FISHEYE ? LBA_CAM_KB8 : LBA_CAM_PINHOLE;
            for (size_t i = 0; i < cams[k]->size() && i < 8; i++) rig.p[k][i] = (double)cams[k]->getParameter((int)i);
        }
        if (bRig) {
            mul31(Rrl, rig.tcb[0], trl, rig.tcb[1]);
            mul33(Rrl, rig.Rcb[0], rig.Rcb[1]);
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rig.Rbc[1][r * 3 + c] = rig.Rcb[1][c * 3 + r];
            double nt[3] = {-rig.tcb[1][0], -rig.tcb[1][1], -rig.tcb[1][2]};
            mul31(rig.Rbc[1], nt, nullptr, rig.tbc[1]);
        }
        IB.setRig(rig);
    }
    // vertices: VertexPose id = mnId, V / G / A ids above every pose id -> key frames in ascending mnId
    struct KfRole { KeyFrame* kf; bool poseFixed, hasImu, imuFixed; };
    std::map<long unsigned int, KfRole> kfById;
    for (int i = 0; i < N; i++) kfById[vpOptimizableKFs[i]->mnId] = KfRole{vpOptimizableKFs[i], false, vpOptimizableKFs[i]->bImu, false};              // :4899-4925
    for (KeyFrame* pKFi : lpOptVisKFs) kfById[pKFi->mnId] = KfRole{pKFi, false, false, false};                                                        // :4928-4935
    for (KeyFrame* pKFi : lFixedKeyFrames) kfById[pKFi->mnId] = KfRole{pKFi, true, pKFi->bImu, true};                                                 // :4938-4962
    std::map<KeyFrame*, int> kfIdx;
    for (auto& kv : kfById) {
        KeyFrame* pKFi = kv.second.kf;
        double Rwb[9], twb[3], Rcw0[9], tcw0[3], Rcw1[9], tcw1[3], v[3] = {0, 0, 0}, bg[3] = {0, 0, 0}, ba[3] = {0, 0, 0};
        widen(pKFi->GetImuRotation(), Rwb); widen(pKFi->GetImuPosition(), twb);      // ImuCamPose(KeyFrame*) (G2oTypes.cc:24-70)
        widen(pKFi->GetRotation(), Rcw0); widen(pKFi->GetTranslation(), tcw0);
        if (bRig) { mul33(Rrl, Rcw0, Rcw1); mul31(Rrl, tcw0, trl, tcw1); }
        if (kv.second.hasImu) { widen(pKFi->GetVelocity(), v); widen(pKFi->GetGyroBias(), bg); widen(pKFi->GetAccBias(), ba); }   // G2oTypes.cc:671-696
        kfIdx[pKFi] = IB.addKeyFrame(Rwb, twb, Rcw0, tcw0, bRig ? Rcw1 : nullptr, bRig ? tcw1 : nullptr, v, bg, ba, kv.second.poseFixed, kv.second.hasImu, kv.second.imuFixed);
    }
    // inertial edges, newest first (:4964-5062)
    for (int i = 0; i < N; i++) {
        KeyFrame* pKFi = vpOptimizableKFs[i];
        if (!pKFi->mPrevKF) continue;   // "NOT INERTIAL LINK TO PREVIOUS FRAME"
        if (pKFi->bImu && pKFi->mPrevKF->bImu && pKFi->mpImuPreintegrated) {
            IMU::Preintegrated* pInt = pKFi->mpImuPreintegrated;
            pInt->SetNewBias(pKFi->mPrevKF->GetImuBias());
            if (!kfIdx.count(pKFi->mPrevKF)) continue;   // a vertex of the previous key frame is missing (:4982-4987)
            liba_imu_edge e{};
            e.kf1 = kfIdx[pKFi->mPrevKF]; e.kf2 = kfIdx[pKFi];
            auto narrow = [](const cv::Mat& m, float* out) { for (int r = 0; r < m.rows; r++) for (int c = 0; c < m.cols; c++) out[r * m.cols + c] = m.at<float>(r, c); };
            narrow(pInt->dR, e.dR); narrow(pInt->dV, e.dV); narrow(pInt->dP, e.dP);
            narrow(pInt->JRg, e.JRg); narrow(pInt->JVg, e.JVg); narrow(pInt->JVa, e.JVa); narrow(pInt->JPg, e.JPg); narrow(pInt->JPa, e.JPa);
            e.b[0] = pInt->b.bax; e.b[1] = pInt->b.bay; e.b[2] = pInt->b.baz; e.b[3] = pInt->b.bwx; e.b[4] = pInt->b.bwy; e.b[5] = pInt->b.bwz;
            e.dT = pInt->dT;
            EdgeInertial ei(pInt);   // the reference's constructor builds the information matrix
            const double scale = (i == N - 1) ? 1e-2 : 1.0;                       // vei[i]->setInformation(vei[i]->information()*1e-2) (:5009)
            for (int r = 0; r < 9; r++) for (int c = 0; c < 9; c++) e.info[r * 9 + c] = ei.information()(r, c) * scale;
            e.huber = (i == N - 1 || bRecInit) ? std::sqrt(16.92) : 0.0;          // rki->setDelta(sqrt(16.92)) (:5005-5012)
            const cv::Mat cvInfoG = pInt->C.rowRange(9, 12).colRange(9, 12).inv(cv::DECOMP_SVD);     // :5024-5029
            const cv::Mat cvInfoA = pInt->C.rowRange(12, 15).colRange(12, 15).inv(cv::DECOMP_SVD);   // :5036-5041
            widen(cvInfoG, e.info_g); widen(cvInfoA, e.info_a);
            IB.addInertial(e);
        }
    }
    // map points and their observations (:5078-5215)
    struct VisRef { KeyFrame* kf; MapPoint* mp; bool stereo; };
    std::vector<VisRef> vis;   // in InertialBA edge order
    std::vector<MapPoint*> vpPoints;
    for (std::list<MapPoint*>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++) {
        MapPoint* pMP = *lit;
        const cv::Mat Xwm = pMP->GetWorldPos();
        const float Xw[3] = {Xwm.at<float>(0), Xwm.at<float>(1), Xwm.at<float>(2)};
        const int l = IB.addPoint(Xw);
        vpPoints.push_back(pMP);
        const std::map<KeyFrame*, std::tuple<int, int>> observations = pMP->GetObservations();
        for (std::map<KeyFrame*, std::tuple<int, int>>::const_iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
            KeyFrame* pKFi = mit->first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) continue;
            if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) {
                if (!kfIdx.count(pKFi)) continue;   // marked but without a vertex (the reference would dereference a null vertex here)
                const int leftIndex = std::get<0>(mit->second);
                cv::KeyPoint kpUn;
                Eigen::Matrix<double, 2, 1> obs;
                if (leftIndex != -1 && pKFi->mvuRight[leftIndex] < 0) {   // Monocular observation (:5103-5126)
                    kpUn = pKFi->mvKeysUn[leftIndex];
                    obs(0, 0) = kpUn.pt.x; obs(1, 0) = kpUn.pt.y;
                    const float unc2 = pKFi->mpCamera->uncertainty2(obs);
                    IB.addMono(kfIdx[pKFi], l, kpUn.pt.x, kpUn.pt.y, pKFi->mvInvLevelSigma2[kpUn.octave] / unc2, 0);
                    vis.push_back(VisRef{pKFi, pMP, false});
                } else if (leftIndex != -1) {                             // Stereo observation (:5128-5156)
                    kpUn = pKFi->mvKeysUn[leftIndex];
                    obs(0, 0) = kpUn.pt.x; obs(1, 0) = kpUn.pt.y;
                    const float unc2 = pKFi->mpCamera->uncertainty2(obs);
                    IB.addStereo(kfIdx[pKFi], l, kpUn.pt.x, kpUn.pt.y, pKFi->mvuRight[leftIndex], pKFi->mvInvLevelSigma2[kpUn.octave] / unc2);
                    vis.push_back(VisRef{pKFi, pMP, true});
                }
                if (pKFi->mpCamera2) {                                    // right camera of the rig (:5159-5192)
                    int rightIndex = std::get<1>(mit->second);
                    if (rightIndex != -1) {
                        rightIndex -= pKFi->NLeft;
                        const cv::KeyPoint kp = pKFi->mvKeysRight[rightIndex];
                        obs(0, 0) = kp.pt.x; obs(1, 0) = kp.pt.y;
                        const float unc2 = pKFi->mpCamera->uncertainty2(obs);
                        // the reference weights this edge with the LEFT key point's octave (kpUn; octave 0 when there is no left observation)
                        IB.addMono(kfIdx[pKFi], l, kp.pt.x, kp.pt.y, pKFi->mvInvLevelSigma2[kpUn.octave] / unc2, 1);
                        vis.push_back(VisRef{pKFi, pMP, false});
                    }
                }
            }
        }
    }

    if (vis.empty()) return;   // no visual edge: nothing to erase and nothing the optimisation could move against
    double dErr = 0, dErrEnd = 0;
    const int its = IB.optimize(bLarge ? 1e-2 : 1e0, opt_it, &dErr, &dErrEnd);   // setUserLambdaInit (:4884-4896), optimizer.optimize(opt_it) (:5225)
    if (its < 0) return;   // the window was rejected (more than LIBA_MAX_FREE optimisable key frames)
    const float err = (float)dErr, err_end = (float)dErrEnd;

    // ---- outlier observations (:5237-5275): monocular edges (left and right camera) first, then stereo ----
    const float chi2Mono2 = 5.991f, chi2Stereo2 = 7.815f;
    std::vector<std::pair<KeyFrame*, MapPoint*>> vToErase;
    vToErase.reserve(vis.size());
    for (int pass = 0; pass < 2; pass++)
        for (size_t i = 0; i < vis.size(); i++) {
            if (vis[i].stereo != (pass == 1)) continue;
            MapPoint* pMP = vis[i].mp;
            if (pMP->isBad()) continue;
            const double chi2 = IB.visualChi2((int)i);
            if (pass == 0) {
                const bool bClose = pMP->mTrackDepth < 10.f;
                if ((chi2 > chi2Mono2 && !bClose) || (chi2 > 1.5f * chi2Mono2 && bClose) || !IB.depthPositive((int)i)) vToErase.push_back(std::make_pair(vis[i].kf, pMP));
            } else if (chi2 > chi2Stereo2)
                vToErase.push_back(std::make_pair(vis[i].kf, pMP));
        }

    // Get Map Mutex and erase outliers
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
    if ((2 * err < err_end || std::isnan(err) || std::isnan(err_end)) && !bLarge) return;   // "FAIL LOCAL-INERTIAL BA" (:5280-5285)
    if (!vToErase.empty())
        for (size_t i = 0; i < vToErase.size(); i++) {
            KeyFrame* pKFi = vToErase[i].first;
            MapPoint* pMPi = vToErase[i].second;
            pKFi->EraseMapPointMatch(pMPi);
            pMPi->EraseObservation(pKFi);
        }
    for (std::list<KeyFrame*>::iterator lit = lFixedKeyFrames.begin(), lend = lFixedKeyFrames.end(); lit != lend; lit++) (*lit)->mnBAFixedForKF = 0;

    // ---- recover optimized data (:5306-5350) ----
    auto toCvSE3 = [](const liba_keyframe& k) {   // Converter::toCvSE3(VP->estimate().Rcw[0], VP->estimate().tcw[0])
        cv::Mat T = cv::Mat::eye(4, 4, CV_32F);
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T.at<float>(r, c) = (float)k.Rcw[0][r * 3 + c]; T.at<float>(r, 3) = (float)k.tcw[0][r]; }
        return T;
    };
    N = vpOptimizableKFs.size();
    for (int i = 0; i < N; i++) {
        KeyFrame* pKFi = vpOptimizableKFs[i];
        const liba_keyframe& k = IB.keyFrame(kfIdx[pKFi]);
        pKFi->SetPose(toCvSE3(k));
        pKFi->mnBALocalForKF = 0;
        if (pKFi->bImu) {
            cv::Mat Vw(3, 1, CV_32F);
            for (int r = 0; r < 3; r++) Vw.at<float>(r) = (float)k.v[r];   // Converter::toCvMat(VV->estimate())
            pKFi->SetVelocity(Vw);
            pKFi->SetNewBias(IMU::Bias(k.ba[0], k.ba[1], k.ba[2], k.bg[0], k.bg[1], k.bg[2]));   // IMU::Bias(b[3],b[4],b[5],b[0],b[1],b[2]) with b = (bg, ba)
        }
    }
    for (std::list<KeyFrame*>::iterator it = lpOptVisKFs.begin(), itEnd = lpOptVisKFs.end(); it != itEnd; it++) {
        KeyFrame* pKFi = *it;
        pKFi->SetPose(toCvSE3(IB.keyFrame(kfIdx[pKFi])));
        pKFi->mnBALocalForKF = 0;
    }
    for (size_t l = 0; l < vpPoints.size(); l++) {
        MapPoint* pMP = vpPoints[l];
        const double* X = IB.point((int)l);
        cv::Mat Xw(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) Xw.at<float>(i) = (float)X[i];
        pMP->SetWorldPos(Xw);
        pMP->UpdateNormalAndDepth();
    }
    pMap->IncreaseChangeIndex();
} ORBHIP_GLUE_CATCH("Optimizer::LocalInertialBA", return;)

}  // namespace ORB_SLAM3
#endif  // ORBHIP_WITH_ORBSLAM3
