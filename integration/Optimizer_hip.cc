// integration/Optimizer_hip.cc — Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*, int&) (reference include/Optimizer.h:58,
// src/Optimizer.cc:1811-2523) and Optimizer::PoseOptimization(Frame*) (include/Optimizer.h:53, src/Optimizer.cc:907-1273) over liborbhip.so.
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

#include <cmath>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <vector>

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

void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF) {
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
}

// Optimizer::PoseOptimization(Frame*) (reference include/Optimizer.h:53, src/Optimizer.cc:907-1273; called after every matcher call in Tracking:
// Tracking.cc:2210, 2395, 2468).  The observation walk (:966-1127) is the reference's, one PoseOptimizer observation per g2o edge in the same
// order; the four optimise / classify rounds (:1133-1252) are the single call PoseOptimizer::optimize (one launch on the device); pose
// recovery and return value as :1255-1270.
int Optimizer::PoseOptimization(Frame* pFrame) {
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
                pFrame->mvbOutlier[i] = false;
                const cv::KeyPoint& kpUn = pFrame->mvKeysUn[i];
                const float invSigma2 = pFrame->mvInvLevelSigma2[kpUn.octave];
                if (pFrame->mvuRight[i] < 0) PO.addMono(Xw, kpUn.pt.x, kpUn.pt.y, invSigma2, camL);                      // :979-1011
                else PO.addStereo(Xw, kpUn.pt.x, kpUn.pt.y, pFrame->mvuRight[i], invSigma2, camL);                        // :1013-1049
            } else {                                 // fisheye rig (:1052-1125): left camera on mvKeys, right camera through mTrl
                nInitialCorrespondences++;
                pFrame->mvbOutlier[i] = false;
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
    if (nInitialCorrespondences < 3) return 0;       // :1130-1131
    double p7[7];
    orbslam3_hip::LbaLinearizer::poseFromTcw(pFrame->mTcw.ptr<float>(), pFrame->mTcw.cols, p7);   // Converter::toSE3Quat(pFrame->mTcw)
    std::vector<bool> outl;
    const int nGood = PO.optimize(p7, outl);         // 4 x optimizer.optimize(10) + chi2 classification, Huber dropped for the last round
    for (size_t k = 0; k < vnIndexEdge.size(); k++) pFrame->mvbOutlier[vnIndexEdge[k]] = outl[k];
    pFrame->SetPose(pose_to_cvmat(p7));              // :1255-1258
    return nGood;                                    // nInitialCorrespondences - nBad
}

}  // namespace ORB_SLAM3
#endif  // ORBHIP_WITH_ORBSLAM3
