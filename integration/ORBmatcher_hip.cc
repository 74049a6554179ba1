// integration/ORBmatcher_hip.cc — the reference-signature bodies of ORB_SLAM3::ORBmatcher's searches over liborbhip.so.
//
// Drop-in for the same-named member functions of src/ORBmatcher.cc (reference include/ORBmatcher.h:46-68): compile this file INSIDE the
// ORB-SLAM3 tree instead of those bodies (integration/README.md), with -DORBHIP_WITH_ORBSLAM3 and <orbhip>/include on the include path.
// Tracking / LocalMapping / LoopClosing call it unchanged.  Every function is gather -> one adapter call -> scatter:
//   * the gather loop is the reference's own loop header and skip conditions (cited per block) with the inner window search removed —
//     all cv::Mat expressions are the reference's, so the projected values are bit-identical to the reference's by construction;
//   * the adapter (include/orbslam3_hip/ORBmatcher.h) does one packed upload, the device search and one packed download on persistent
//     device buffers (one thread-local adapter per calling thread — Tracking, LocalMapping and LoopClosing each own one);
//   * the scatter applies the result in the reference's serial order (mvpMapPoints[idx] = pMP, vpMatched[idx] = pMP, ...).
// tests/test_glue.py compiles this file against minimal mock declarations of the classes (tests/cpp/mock_orbslam3) and checks every
// function against the oracle's literal restatement of the reference loop.
#ifdef ORBHIP_WITH_ORBSLAM3
#include "ORBmatcher.h"

#include <cmath>
#include <cstring>
#include <set>
#include <vector>

#include <orbslam3_hip/GlueGuard.h>
#include <orbslam3_hip/ORBmatcher.h>

namespace ORB_SLAM3 {

namespace {
orbslam3_hip::ORBmatcher& device_matcher(float nnratio, bool checkOri) {
    static thread_local orbslam3_hip::ORBmatcher m;   // persistent device buffers, grown on demand
    m.mfNNratio = nnratio; m.mbCheckOrientation = checkOri;
    return m;
}
static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint is the 28-byte record of orbhip.h");

// What the searches read from a Frame (Nleft == -1: mvKeysUn; fisheye rig: the concatenation [mvKeys | mvKeysRight] the reference indexes
// with idx / idx - Nleft, and mvLeftToRightMatch / mvRightToLeftMatch as global partner links).
struct FrameGather {
    std::vector<orb_keypoint> keys;       // rig only
    std::vector<uint8_t> occ;
    std::vector<int32_t> link;
    orbslam3_hip::FrameView V;
    // occupied(i): the skip rule of the search at hand on F.mvpMapPoints[i] (Observations() > 0, or != NULL for the relocalisation search)
    template <class OccFn> FrameGather(const Frame& F, OccFn occupied, bool useURight) {
        V.N = F.N;
        V.descriptors = F.mDescriptors.data;
        V.grid = orbm_grid_params{Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv};
        occ.resize(F.N);
        for (int i = 0; i < F.N; i++) occ[i] = occupied(F.mvpMapPoints[i]) ? 1 : 0;
        V.occupied = occ.data();
        if (F.Nleft == -1) {
            V.keysUn = (const orb_keypoint*)F.mvKeysUn.data();
            V.uRight = useURight ? F.mvuRight.data() : nullptr;
        } else {
            keys.resize(F.N);
            std::memcpy(keys.data(), F.mvKeys.data(), (size_t)F.Nleft * sizeof(orb_keypoint));
            std::memcpy(keys.data() + F.Nleft, F.mvKeysRight.data(), (size_t)(F.N - F.Nleft) * sizeof(orb_keypoint));
            V.keysUn = keys.data();
            V.Nleft = F.Nleft;
            link.assign(F.N, -1);
            for (int i = 0; i < F.Nleft; i++) if (F.mvLeftToRightMatch[i] != -1) link[i] = F.mvLeftToRightMatch[i] + F.Nleft;
            for (int i = F.Nleft; i < F.N; i++) link[i] = F.mvRightToLeftMatch[i - F.Nleft];
            V.kpLink = link.data();
        }
    }
};

// DBoW2::FeatureVector (std::map<NodeId, vector<unsigned>>) -> the CSR of orbm_bow_side: map order = ascending node id
void flatten_featvec(const DBoW2::FeatureVector& fv, orbslam3_hip::ORBmatcher::KeyFrameView& K) {
    K.nodeStart.push_back(0);
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
        K.nodeId.push_back((int32_t)it->first);
        K.featIdx.insert(K.featIdx.end(), it->second.begin(), it->second.end());
        K.nodeStart.push_back((int32_t)K.featIdx.size());
    }
}
}  // namespace

// ---- SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)   ORBmatcher.cc:59-255, Tracking.cc:2964 ------
int ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th, const bool bFarPoints, const float thFarPoints) try {
    FrameGather G(F, [](MapPoint* p) { return p && p->Observations() > 0; }, true);   // :125-127 (left), :212-214 (right)
    const bool rig = F.Nleft != -1;
    std::vector<orbm_query> q;
    std::vector<uint8_t> qd;
    std::vector<MapPoint*> owner;   // query -> map point
    q.reserve(vpMapPoints.size() * (rig ? 2 : 1)); owner.reserve(q.capacity());
    const bool bFactor = th != 1.0;
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
        MapPoint* pMP = vpMapPoints[iMP];
        if (!pMP->mbTrackInView && !pMP->mbTrackInViewR) continue;   // :73-74
        if (bFarPoints && pMP->mTrackDepth > thFarPoints) continue;  // :77-78
        if (pMP->isBad()) continue;                                  // :81-82
        const uint32_t obs = pMP->Observations() > 0 ? ORBM_Q_HAS_OBS : 0u;
        orbm_query ql{};                                             // left camera, :85-181
        if (pMP->mbTrackInView) {
            const int& nPredictedLevel = pMP->mnTrackScaleLevel;
            float r = RadiusByViewingCos(pMP->mTrackViewCos);
            if (bFactor) r *= th;
            ql = orbm_query{pMP->mTrackProjX, pMP->mTrackProjY, r * F.mvScaleFactors[nPredictedLevel], pMP->mTrackProjXR, 0.f,
                            (int16_t)(nPredictedLevel - 1), (int16_t)nPredictedLevel, ORBM_Q_VALID | obs | (rig ? 0u : ORBM_Q_STEREO)};
        }
        const bool right = rig && pMP->mbTrackInViewR && pMP->mnTrackScaleLevelR != -1;   // :184-187
        if (!pMP->mbTrackInView && !right) continue;
        const cv::Mat MPdescriptor = pMP->GetDescriptor();
        // the left query — or, for a point only the right camera sees, a placeholder with flags = 0 that keeps the (left, right-twin) pairing aligned
        q.push_back(ql); owner.push_back(pMP);
        qd.insert(qd.end(), MPdescriptor.data, MPdescriptor.data + 32);
        if (right) {
            const int& nPredictedLevel = pMP->mnTrackScaleLevelR;
            const float r = RadiusByViewingCos(pMP->mTrackViewCosR);   // not multiplied by th (:190)
            q.push_back(orbm_query{pMP->mTrackProjXR, pMP->mTrackProjYR, r * F.mvScaleFactors[nPredictedLevel], 0.f, 0.f, (int16_t)(nPredictedLevel - 1),
                                   (int16_t)nPredictedLevel, ORBM_Q_VALID | obs | ORBM_Q_RIGHT | ORBM_Q_TWIN});
            owner.push_back(pMP);
            qd.insert(qd.end(), MPdescriptor.data, MPdescriptor.data + 32);
        }
    }
    std::vector<int> kpMatch, qMatch;
    const int nmatches = device_matcher(mfNNratio, mbCheckOrientation).SearchByProjection(G.V, q, qd, ORBM_MODE_LOCAL_MAP, TH_HIGH, kpMatch, qMatch);
    for (int idx = 0; idx < F.N; idx++)   // F.mvpMapPoints[bestIdx] = pMP (:171, :174, :241, :246): the holder after the serial loop
        if (kpMatch[idx] >= 0) F.mvpMapPoints[idx] = owner[kpMatch[idx]];
    return nmatches;
} ORBHIP_GLUE_CATCH("ORBmatcher::SearchByProjection", return 0;)

// ---- SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)   ORBmatcher.cc:2244-2509, Tracking.cc:2363-2378 --------
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) try {
    FrameGather G(CurrentFrame, [](MapPoint* p) { return p && p->Observations() > 0; }, true);   // :2347-2349
    const bool rig = CurrentFrame.Nleft != -1;
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat twc = -Rcw.t() * tcw;
    const cv::Mat Rlw = LastFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tlw = LastFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat tlc = Rlw * twc + tlw;
    const bool bForward = tlc.at<float>(2) > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc.at<float>(2) > CurrentFrame.mb && !bMono;
    std::vector<orbm_query> q;
    std::vector<uint8_t> qd;
    std::vector<MapPoint*> owner;
    for (int i = 0; i < LastFrame.N; i++) {
        MapPoint* pMP = LastFrame.mvpMapPoints[i];
        if (!pMP) continue;
        if (LastFrame.mvbOutlier[i]) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float invzc = 1.0 / x3Dc.at<float>(2);
        if (invzc < 0) continue;
        cv::Point2f uv = CurrentFrame.mpCamera->project(x3Dc);
        if (uv.x < CurrentFrame.mnMinX || uv.x > CurrentFrame.mnMaxX) continue;
        if (uv.y < CurrentFrame.mnMinY || uv.y > CurrentFrame.mnMaxY) continue;
        const int nLastOctave = (LastFrame.Nleft == -1 || i < LastFrame.Nleft) ? LastFrame.mvKeys[i].octave : LastFrame.mvKeysRight[i - LastFrame.Nleft].octave;
        const float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
        // GetFeaturesInArea level window: forward [nLastOctave, -1], backward [0, nLastOctave], else [nLastOctave-1, nLastOctave+1] (:2310-2326)
        const int16_t lo = bForward ? nLastOctave : (bBackward ? 0 : nLastOctave - 1), hi = bForward ? -1 : (bBackward ? nLastOctave : nLastOctave + 1);
        const cv::KeyPoint& kpLF = (LastFrame.Nleft == -1) ? LastFrame.mvKeysUn[i] : (i < LastFrame.Nleft) ? LastFrame.mvKeys[i] : LastFrame.mvKeysRight[i - LastFrame.Nleft];
        // a keypoint taken earlier in this call is skipped by later points iff its new holder has Observations() > 0 (:2347-2349); the temporal
        // points Tracking::UpdateLastFrame creates have none and may be overwritten
        const uint32_t flagsObs = pMP->Observations() > 0 ? ORBM_Q_HAS_OBS : 0u;
        const cv::Mat dMP = pMP->GetDescriptor();
        q.push_back(orbm_query{uv.x, uv.y, radius, uv.x - CurrentFrame.mbf * invzc, kpLF.angle, lo, hi, ORBM_Q_VALID | flagsObs | (rig ? 0u : ORBM_Q_STEREO)});
        owner.push_back(pMP);
        qd.insert(qd.end(), dMP.data, dMP.data + 32);
        if (rig) {   // :2403-2460: the same point through the right camera
            cv::Mat x3Dr = CurrentFrame.mTrl.colRange(0, 3).rowRange(0, 3) * x3Dc + CurrentFrame.mTrl.col(3);
            cv::Point2f uvr = CurrentFrame.mpCamera->project(x3Dr);
            q.push_back(orbm_query{uvr.x, uvr.y, radius, 0.f, kpLF.angle, lo, hi, ORBM_Q_VALID | flagsObs | ORBM_Q_RIGHT | ORBM_Q_TWIN});
            owner.push_back(pMP);
            qd.insert(qd.end(), dMP.data, dMP.data + 32);
        }
    }
    std::vector<int> kpMatch, qMatch;
    const int nmatches = device_matcher(mfNNratio, mbCheckOrientation).SearchByProjection(G.V, q, qd, ORBM_MODE_BEST_ONLY, TH_HIGH, kpMatch, qMatch);
    // scatter: a keypoint claimed during the call holds its final claimant (:2372, :2432), or NULL if the orientation cull removed it (:2499,
    // kpMatch == -2); untouched keypoints (-1) keep what they held
    for (int idx = 0; idx < CurrentFrame.N; idx++) {
        if (kpMatch[idx] >= 0) CurrentFrame.mvpMapPoints[idx] = owner[kpMatch[idx]];
        else if (kpMatch[idx] == -2) CurrentFrame.mvpMapPoints[idx] = static_cast<MapPoint*>(NULL);
    }
    return nmatches;
} ORBHIP_GLUE_CATCH("ORBmatcher::SearchByProjection", return 0;)

// ---- SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)   ORBmatcher.cc:2520-2652, Tracking.cc:3403,3417 (relocalisation) ----
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist) try {
    FrameGather G(CurrentFrame, [](MapPoint* p) { return p != NULL; }, false);   // `if(CurrentFrame.mvpMapPoints[i2]) continue;` :2586
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -Rcw.t() * tcw;
    const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
    std::vector<orbm_query> q;
    std::vector<uint8_t> qd;
    std::vector<MapPoint*> owner;
    for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
        MapPoint* pMP = vpMPs[i];
        if (!pMP) continue;
        if (pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const cv::Point2f uv = CurrentFrame.mpCamera->project(x3Dc);
        if (uv.x < CurrentFrame.mnMinX || uv.x > CurrentFrame.mnMaxX) continue;
        if (uv.y < CurrentFrame.mnMinY || uv.y > CurrentFrame.mnMaxY) continue;
        cv::Mat PO = x3Dw - Ow;
        float dist3D = cv::norm(PO);
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
        const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
        const cv::Mat dMP = pMP->GetDescriptor();
        q.push_back(orbm_query{uv.x, uv.y, radius, 0.f, pKF->mvKeysUn[i].angle, (int16_t)(nPredictedLevel - 1), (int16_t)(nPredictedLevel + 1),
                               ORBM_Q_VALID | ORBM_Q_HAS_OBS});   // any point placed during the call blocks its keypoint (mvpMapPoints[i2] != NULL)
        owner.push_back(pMP);
        qd.insert(qd.end(), dMP.data, dMP.data + 32);
    }
    std::vector<int> kpMatch, qMatch;
    const int nmatches = device_matcher(mfNNratio, mbCheckOrientation).SearchByProjection(G.V, q, qd, ORBM_MODE_BEST_ONLY, ORBdist, kpMatch, qMatch);
    for (int idx = 0; idx < CurrentFrame.N; idx++) {
        if (kpMatch[idx] >= 0) CurrentFrame.mvpMapPoints[idx] = owner[kpMatch[idx]];
        else if (kpMatch[idx] == -2) CurrentFrame.mvpMapPoints[idx] = NULL;   // :2637
    }
    return nmatches;
} ORBHIP_GLUE_CATCH("ORBmatcher::SearchByProjection", return 0;)

// ---- the two Sim3 projection searches of loop closing / merging   ORBmatcher.cc:593-706 and :708-824 ------------------------------------
namespace {
struct Sim3Gather {
    std::vector<orbm_query> q;
    std::vector<uint8_t> qd;
    std::vector<int> src;   // query -> index into vpPoints
};
// the projection loop both overloads share (:600-652 / :716-768), verbatim conditions
Sim3Gather gather_sim3(KeyFrame* pKF, const cv::Mat& Scw, const std::vector<MapPoint*>& vpPoints, const std::vector<MapPoint*>& vpMatched, int th,
                       float thFuse = -1.f) {   // thFuse >= 0: the float `th` of Fuse(pKF, Scw, ...) (:1881) instead of the searches' int
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
    Sim3Gather S;
    for (int iMP = 0, iendMP = vpPoints.size(); iMP < iendMP; iMP++) {
        MapPoint* pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.at<float>(2) < 0.0) continue;
        const float x = p3Dc.at<float>(0), y = p3Dc.at<float>(1), z = p3Dc.at<float>(2);
        const cv::Point2f uv = pKF->mpCamera->project(cv::Point3f(x, y, z));
        if (!pKF->IsInImage(uv.x, uv.y)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        cv::Mat PO = p3Dw - Ow;
        const float dist = cv::norm(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        cv::Mat Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist) continue;
        int nPredictedLevel = pMP->PredictScale(dist, pKF);
        const float radius = thFuse >= 0.f ? thFuse * pKF->mvScaleFactors[nPredictedLevel] : th * pKF->mvScaleFactors[nPredictedLevel];
        const cv::Mat dMP = pMP->GetDescriptor();
        // KeyFrame::GetFeaturesInArea has no level filter; the loop keeps [nPredictedLevel-1, nPredictedLevel] (:671-674): same candidates, same order
        S.q.push_back(orbm_query{uv.x, uv.y, radius, 0.f, 0.f, (int16_t)(nPredictedLevel - 1), (int16_t)nPredictedLevel, ORBM_Q_VALID | ORBM_Q_HAS_OBS});
        S.src.push_back(iMP);
        S.qd.insert(S.qd.end(), dMP.data, dMP.data + 32);
    }
    return S;
}
orbslam3_hip::FrameView keyframe_view(KeyFrame* pKF, std::vector<uint8_t>& occ, const std::vector<MapPoint*>& vpMatched) {
    orbslam3_hip::FrameView V;
    V.N = (int)pKF->mvKeysUn.size();
    V.keysUn = (const orb_keypoint*)pKF->mvKeysUn.data();
    V.descriptors = pKF->mDescriptors.data;
    V.grid = orbm_grid_params{(float)pKF->mnMinX, (float)pKF->mnMinY, pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv};
    occ.resize(V.N);
    for (int i = 0; i < V.N; i++) occ[i] = vpMatched[i] ? 1 : 0;   // `if(vpMatched[idx]) continue;` :668
    V.occupied = occ.data();
    return V;
}
}  // namespace

int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th, float ratioHamming) try {
    const Sim3Gather S = gather_sim3(pKF, Scw, vpPoints, vpMatched, th);
    std::vector<uint8_t> occ;
    const orbslam3_hip::FrameView V = keyframe_view(pKF, occ, vpMatched);
    std::vector<int> kpMatch, qMatch;
    // bestDist <= TH_LOW*ratioHamming (:686) with an integer distance <=> bestDist <= floor(TH_LOW*ratioHamming); no orientation check in this search
    const int thDist = (int)std::floor((float)TH_LOW * ratioHamming);
    const int nmatches = device_matcher(mfNNratio, false).SearchByProjection(V, S.q, S.qd, ORBM_MODE_BEST_ONLY, thDist, kpMatch, qMatch);
    for (int idx = 0; idx < V.N; idx++)
        if (kpMatch[idx] >= 0) vpMatched[idx] = vpPoints[S.src[kpMatch[idx]]];   // :688
    return nmatches;
} ORBHIP_GLUE_CATCH("ORBmatcher::SearchByProjection", return 0;)

int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, const std::vector<KeyFrame*>& vpPointsKFs,
                                   std::vector<MapPoint*>& vpMatched, std::vector<KeyFrame*>& vpMatchedKF, int th, float ratioHamming) try {
    const Sim3Gather S = gather_sim3(pKF, Scw, vpPoints, vpMatched, th);
    std::vector<uint8_t> occ;
    const orbslam3_hip::FrameView V = keyframe_view(pKF, occ, vpMatched);
    std::vector<int> kpMatch, qMatch;
    const int thDist = (int)std::floor((float)TH_LOW * ratioHamming);
    const int nmatches = device_matcher(mfNNratio, false).SearchByProjection(V, S.q, S.qd, ORBM_MODE_BEST_ONLY, thDist, kpMatch, qMatch);
    for (int idx = 0; idx < V.N; idx++)
        if (kpMatch[idx] >= 0) { vpMatched[idx] = vpPoints[S.src[kpMatch[idx]]]; vpMatchedKF[idx] = vpPointsKFs[S.src[kpMatch[idx]]]; }   // :806-807
    return nmatches;
} ORBHIP_GLUE_CATCH("ORBmatcher::SearchByProjection", return 0;)

// ---- SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches)   ORBmatcher.cc:323-587, Tracking.cc:2185 / :3340 ------------------------------------
int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) try {
    const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
    const int nKF = (int)vpMapPointsKF.size();
    orbslam3_hip::ORBmatcher::KeyFrameView K, Fv;
    std::vector<uint8_t> valid(nKF);
    for (int i = 0; i < nKF; i++) valid[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();   // :360-366
    std::vector<float> angK(nKF), angF(F.N);
    for (int i = 0; i < nKF; i++)   // kp of :481-484
        angK[i] = (!pKF->mpCamera2) ? pKF->mvKeysUn[i].angle : (i >= pKF->NLeft) ? pKF->mvKeysRight[i - pKF->NLeft].angle : pKF->mvKeys[i].angle;
    for (int i = 0; i < F.N; i++)     // Fkp of :493-496 / :527-530
        angF[i] = (F.Nleft == -1 || i < F.Nleft) ? F.mvKeys[i].angle : F.mvKeysRight[i - F.Nleft].angle;
    K.N = nKF; K.descriptors = pKF->mDescriptors.data; K.hasMapPoint = valid.data();
    Fv.N = F.N; Fv.descriptors = F.mDescriptors.data;
    flatten_featvec(pKF->mFeatVec, K);
    flatten_featvec(F.mFeatVec, Fv);
    std::vector<int> fMatch;
    const int nmatches = device_matcher(mfNNratio, mbCheckOrientation).SearchByBoW(K, angK.data(), Fv, angF.data(), F.Nleft, fMatch);
    for (int j = 0; j < F.N; j++)
        if (fMatch[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[fMatch[j]];   // :473, :513
    return nmatches;
} ORBHIP_GLUE_CATCH("ORBmatcher::SearchByBoW", return 0;)

// ---- SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)   ORBmatcher.cc:984-1124, LoopClosing.cc:697 --------------------------------------------
int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) try {
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));
    KeyFrame* kf[2] = {pKF1, pKF2};
    const std::vector<MapPoint*>* mps[2] = {&vpMapPoints1, &vpMapPoints2};
    orbslam3_hip::ORBmatcher::KeyFrameView K[2];
    std::vector<uint8_t> valid[2];
    std::vector<float> ang[2];
    for (int s = 0; s < 2; s++) {
        const int n = (int)mps[s]->size(), nUn = (int)kf[s]->mvKeysUn.size();
        valid[s].resize(n); ang[s].assign(n, 0.f);
        for (int i = 0; i < n; i++) {
            MapPoint* p = (*mps[s])[i];
            // `NLeft != -1 && idx >= mvKeysUn.size()` (:1020-1022, :1043-1045) and `!pMP || pMP->isBad()` (:1025-1028, :1049-1053)
            valid[s][i] = !(kf[s]->NLeft != -1 && i >= nUn) && p && !p->isBad();
            if (i < nUn) ang[s][i] = kf[s]->mvKeysUn[i].angle;   // vKeysUn1[idx1].angle / vKeysUn2[bestIdx2].angle :1082
        }
        K[s].N = n; K[s].descriptors = kf[s]->mDescriptors.data; K[s].hasMapPoint = valid[s].data();
        flatten_featvec(kf[s]->mFeatVec, K[s]);
    }
    std::vector<int> m12;
    const int nmatches = device_matcher(mfNNratio, mbCheckOrientation).SearchByBoW(K[0], ang[0].data(), K[1], ang[1].data(), m12);
    for (size_t i = 0; i < m12.size(); i++)
        if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];   // :1076
    return nmatches;
} ORBHIP_GLUE_CATCH("ORBmatcher::SearchByBoW", return 0;)

// ---- SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)   ORBmatcher.cc:838-979, Tracking.cc (monocular initialisation) ----
int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize) try {
    vnMatches12 = std::vector<int>(F1.mvKeysUn.size(), -1);   // :842 (sized before the device call: the caller indexes it whatever happens)
    FrameGather G1(F1, [](MapPoint*) { return false; }, false), G2(F2, [](MapPoint*) { return false; }, false);
    G1.V.N = (int)F1.mvKeysUn.size(); G2.V.N = (int)F2.mvKeysUn.size();   // the function indexes mvKeysUn (:844, :856-859)
    std::vector<float> prev(2 * (size_t)G1.V.N);
    for (int i = 0; i < G1.V.N; i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
    const int nmatches = device_matcher(mfNNratio, mbCheckOrientation).SearchForInitialization(G1.V, G2.V, prev, vnMatches12, windowSize);
    for (size_t i1 = 0, iend1 = vnMatches12.size(); i1 < iend1; i1++)   // Update prev matched (:972-975)
        if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt;
    return nmatches;
} ORBHIP_GLUE_CATCH("ORBmatcher::SearchForInitialization", return 0;)

// ---- SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse)   ORBmatcher.cc:1138-1428, LocalMapping.cc:628 ------
// The statements before the loops (:1144-1193: epipole, R12 / t12 or the four left / right combinations of a rig) are the reference's cv::Mat
// expressions; the vocabulary-node walk with the epipolar gate runs on the device.  `F12` is not read — as in the reference, whose
// GeometricCamera::epipolarConstrain recomputes what it needs from R12, t12 and the calibrations (Pinhole.cpp:155-160, KannalaBrandt8.cpp:235-330).
int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                                       const bool bOnlyStereo, const bool bCoarse) try {
    (void)F12;
    // Compute epipole in second image (:1144-1152)
    cv::Mat Cw = pKF1->GetCameraCenter();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat C2 = R2w * Cw + t2w;
    const cv::Point2f ep = pKF2->mpCamera->project(C2);
    cv::Mat R1w = pKF1->GetRotation();
    cv::Mat t1w = pKF1->GetTranslation();
    GeometricCamera *pCamera1 = pKF1->mpCamera, *pCamera2 = pKF2->mpCamera;
    const bool rig = pKF1->mpCamera2 || pKF2->mpCamera2;
    const bool fisheye = pCamera1->GetType() == pCamera1->CAM_FISHEYE;

    // what the loops read from a key frame: mvKeysUn (a rig: [mvKeys | mvKeysRight], indexed idx / idx - NLeft :1252-1262), descriptors, mvuRight,
    // GetMapPoint(idx) != NULL (:1229-1234), mFeatVec
    KeyFrame* kf[2] = {pKF1, pKF2};
    orbslam3_hip::ORBmatcher::KeyFrameView K[2];
    std::vector<orb_keypoint> keys[2];
    std::vector<uint8_t> has[2];
    for (int s = 0; s < 2; s++) {
        const int n = kf[s]->N;
        K[s].N = n;
        if (kf[s]->NLeft == -1) K[s].keysUn = (const orb_keypoint*)kf[s]->mvKeysUn.data();
        else {
            keys[s].resize(n);
            std::memcpy(keys[s].data(), kf[s]->mvKeys.data(), (size_t)kf[s]->NLeft * sizeof(orb_keypoint));
            std::memcpy(keys[s].data() + kf[s]->NLeft, kf[s]->mvKeysRight.data(), (size_t)(n - kf[s]->NLeft) * sizeof(orb_keypoint));
            K[s].keysUn = keys[s].data();
        }
        K[s].descriptors = kf[s]->mDescriptors.data;
        K[s].uRight = (!rig && !fisheye) ? kf[s]->mvuRight.data() : nullptr;
        has[s].resize(n);
        for (int i = 0; i < n; i++) has[s][i] = kf[s]->GetMapPoint(i) ? 1 : 0;
        K[s].hasMapPoint = has[s].data();
        flatten_featvec(kf[s]->mFeatVec, K[s]);
    }
    orbslam3_hip::ORBmatcher& M = device_matcher(mfNNratio, mbCheckOrientation);
    const int nLevels = (int)pKF2->mvScaleFactors.size();
    if (!rig && !fisheye) {
        // :1176-1177, then Pinhole::epipolarConstrain's matrix (Pinhole.cpp:157-160), evaluated once per pair instead of once per candidate
        cv::Mat R12 = R1w * R2w.t();
        cv::Mat t12 = -R1w * R2w.t() * t2w + t1w;
        cv::Mat t12x = cv::Mat(3, 3, CV_32F);
        {   // Pinhole::SkewSymmetricMatrix (Pinhole.cpp:195-200)
            const float x = t12.at<float>(0), y = t12.at<float>(1), z = t12.at<float>(2);
            const float sk[9] = {0, -z, y, z, 0, -x, -y, x, 0};
            for (int i = 0; i < 9; i++) t12x.at<float>(i / 3, i % 3) = sk[i];
        }
        cv::Mat K1 = pCamera1->toK();
        cv::Mat K2 = pCamera2->toK();
        cv::Mat F = K1.t().inv() * t12x * R12 * K2.inv();
        float Ff[9];
        for (int i = 0; i < 9; i++) Ff[i] = F.at<float>(i / 3, i % 3);
        const float epf[2] = {ep.x, ep.y};
        return M.SearchForTriangulation(K[0], K[1], Ff, epf, pKF2->mvLevelSigma2.data(), pKF2->mvScaleFactors.data(), nLevels, vMatchedPairs,
                                        bOnlyStereo, bCoarse);
    }
    // KannalaBrandt8 key frames — one fisheye camera, or a rig with the four left / right combinations (:1181-1193)
    orbm_tri_kb8_pair P{};
    P.n_cams = rig ? 2 : 1;
    GeometricCamera* cams[2][2] = {{pKF1->mpCamera, pKF1->mpCamera2}, {pKF2->mpCamera, pKF2->mpCamera2}};
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < 8; i++) {
            if (cams[0][c] && i < (int)cams[0][c]->size()) P.k1[c][i] = cams[0][c]->getParameter(i);
            if (cams[1][c] && i < (int)cams[1][c]->size()) P.k2[c][i] = cams[1][c]->getParameter(i);
        }
    auto put = [&](int idx, const cv::Mat& R, const cv::Mat& t) {
        for (int i = 0; i < 9; i++) P.R12[idx][i] = R.at<float>(i / 3, i % 3);
        for (int i = 0; i < 3; i++) P.t12[idx][i] = t.at<float>(i);
    };
    if (!rig) put(0, R1w * R2w.t(), -R1w * R2w.t() * t2w + t1w);
    else {
        put(0, pKF1->GetRotation() * pKF2->GetRotation().t(), pKF1->GetRotation() * (-pKF2->GetRotation().t() * pKF2->GetTranslation()) + pKF1->GetTranslation());
        put(1, pKF1->GetRotation() * pKF2->GetRightRotation().t(),
            pKF1->GetRotation() * (-pKF2->GetRightRotation().t() * pKF2->GetRightTranslation()) + pKF1->GetTranslation());
        put(2, pKF1->GetRightRotation() * pKF2->GetRotation().t(),
            pKF1->GetRightRotation() * (-pKF2->GetRotation().t() * pKF2->GetTranslation()) + pKF1->GetRightTranslation());
        put(3, pKF1->GetRightRotation() * pKF2->GetRightRotation().t(),
            pKF1->GetRightRotation() * (-pKF2->GetRightRotation().t() * pKF2->GetRightTranslation()) + pKF1->GetRightTranslation());
    }
    P.ep[0] = ep.x; P.ep[1] = ep.y;
    for (int i = 0; i < 16; i++) {
        if (i < (int)pKF1->mvLevelSigma2.size()) P.level_sigma2_1[i] = pKF1->mvLevelSigma2[i];
        if (i < (int)pKF2->mvLevelSigma2.size()) P.level_sigma2_2[i] = pKF2->mvLevelSigma2[i];
        if (i < nLevels) P.scale_factors_2[i] = pKF2->mvScaleFactors[i];
    }
    return M.SearchForTriangulationKB8(K[0], pKF1->NLeft, K[1], pKF2->NLeft, P, vMatchedPairs, bOnlyStereo, bCoarse);
} ORBHIP_GLUE_CATCH("ORBmatcher::SearchForTriangulation", return 0;)

// ---- SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)   ORBmatcher.cc:2008-2232 (loop closing: more matches under a known Sim3) ------
// The two projection passes (:2044-2080 key frame 1's points into key frame 2, :2131-2167 the other way) are the reference's statements up to
// GetFeaturesInArea; both window searches and the mutual-agreement test (:2206-2220) run on the device.
int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12,
                             const float th) try {
    const float &fx = pKF1->fx, &fy = pKF1->fy, &cx = pKF1->cx, &cy = pKF1->cy;
    cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation();   // Camera 1 from world
    cv::Mat R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();   // Camera 2 from world
    cv::Mat sR12 = s12 * R12;                                          // Transformation between cameras
    cv::Mat sR21 = (1.0 / s12) * R12.t();
    cv::Mat t21 = -sR21 * t12;
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size();
    const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {
        MapPoint* pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            const int idx2 = std::get<0>(pMP->GetIndexInKeyFrame(pKF2));
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    // one pass: the points of `from` (through Rw, tw into their own camera, then sR, t into the other one) as window queries in `into`
    auto pass = [&](const std::vector<MapPoint*>& vp, const std::vector<bool>& already, const cv::Mat& Rw, const cv::Mat& tw, const cv::Mat& sR, const cv::Mat& t,
                    KeyFrame* into, std::vector<orbm_query>& q, std::vector<uint8_t>& qd) {
        const int N = (int)vp.size();
        q.assign((size_t)N, orbm_query{});
        qd.assign((size_t)N * 32, 0);
        for (int i = 0; i < N; i++) {
            MapPoint* pMP = vp[i];
            if (!pMP || already[i]) continue;
            if (pMP->isBad()) continue;
            cv::Mat p3Dw = pMP->GetWorldPos();
            cv::Mat p3Dc = Rw * p3Dw + tw;
            cv::Mat p3Do = sR * p3Dc + t;
            if (p3Do.at<float>(2) < 0.0) continue;             // Depth must be positive
            const float invz = 1.0 / p3Do.at<float>(2);
            const float x = p3Do.at<float>(0) * invz, y = p3Do.at<float>(1) * invz;
            const float u = fx * x + cx, v = fy * y + cy;
            if (!into->IsInImage(u, v)) continue;              // Point must be inside the image
            const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
            const float dist3D = cv::norm(p3Do);
            if (dist3D < minDistance || dist3D > maxDistance) continue;   // Depth must be inside the scale invariance region
            const int nPredictedLevel = pMP->PredictScale(dist3D, into);
            q[i].u = u; q[i].v = v; q[i].radius = th * into->mvScaleFactors[nPredictedLevel];
            q[i].min_level = (int16_t)(nPredictedLevel - 1); q[i].max_level = (int16_t)nPredictedLevel;
            q[i].flags = ORBM_Q_VALID;
            std::memcpy(&qd[(size_t)i * 32], pMP->GetDescriptor().data, 32);
        }
    };
    std::vector<orbm_query> q12, q21;
    std::vector<uint8_t> d12, d21;
    pass(vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, pKF2, q12, d12);
    pass(vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12, pKF1, q21, d21);
    std::vector<uint8_t> occ1, occ2;
    const std::vector<MapPoint*> none1(pKF1->mvKeysUn.size(), (MapPoint*)NULL), none2(pKF2->mvKeysUn.size(), (MapPoint*)NULL);
    const orbslam3_hip::FrameView V1 = keyframe_view(pKF1, occ1, none1), V2 = keyframe_view(pKF2, occ2, none2);
    std::vector<int> m12;
    const int nFound = device_matcher(mfNNratio, mbCheckOrientation).SearchBySim3(V1, V2, q12, d12, q21, d21, m12);
    for (int i1 = 0; i1 < N1; i1++)
        if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]];   // :2214
    return nFound;
} ORBHIP_GLUE_CATCH("ORBmatcher::SearchBySim3", return 0;)

// ---- Fuse(pKF, vpMapPoints, th, bRight)   ORBmatcher.cc:1630-1879, LocalMapping.cc:1006-1042 (every key frame, every neighbour) -----------
// Per map point the gates and the projection (:1666-1765) are the reference's statements; the window search with the chi2 gate and the
// best-descriptor choice (:1767-1826) runs on the device for all points at once; Replace / AddObservation (:1828-1855) are applied in index
// order afterwards.  The map changes while the reference's loop runs — a point that an earlier iteration replaced is bad when its own turn
// comes, a feature that an earlier iteration gave a map point has one — so isBad() / IsInKeyFrame() are read again at each point's turn and
// pKF->GetMapPoint(bestIdx) is read at that moment, exactly as the serial loop sees them; the search itself reads nothing that changes.
int ORBmatcher::Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th, const bool bRight) try {
    cv::Mat Rcw, tcw, Ow;
    GeometricCamera* pCamera;
    if (bRight) { Rcw = pKF->GetRightRotation(); tcw = pKF->GetRightTranslation(); Ow = pKF->GetRightCameraCenter(); pCamera = pKF->mpCamera2; }
    else { Rcw = pKF->GetRotation(); tcw = pKF->GetTranslation(); Ow = pKF->GetCameraCenter(); pCamera = pKF->mpCamera; }
    const float& bf = pKF->mbf;
    const int nMPs = (int)vpMapPoints.size();
    std::vector<orbm_query> q((size_t)nMPs);
    std::vector<uint8_t> qd((size_t)nMPs * 32, 0);
    for (int i = 0; i < nMPs; i++) {
        orbm_query& Q = q[i];
        Q = orbm_query{};
        MapPoint* pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad()) continue;                    // read again at the point's turn, below
        else if (pMP->IsInKeyFrame(pKF)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.at<float>(2) < 0.0f) continue;        // Depth must be positive
        const float invz = 1 / p3Dc.at<float>(2);
        const float x = p3Dc.at<float>(0), y = p3Dc.at<float>(1), z = p3Dc.at<float>(2);
        const cv::Point2f uv = pCamera->project(cv::Point3f(x, y, z));
        if (!pKF->IsInImage(uv.x, uv.y)) continue;     // Point must be inside the image
        const float ur = uv.x - bf * invz;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        cv::Mat PO = p3Dw - Ow;
        const float dist3D = cv::norm(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;   // Depth must be inside the scale pyramid of the image
        cv::Mat Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist3D) continue;       // Viewing angle must be less than 60 deg
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
        Q.u = uv.x; Q.v = uv.y; Q.u_right = ur;
        Q.radius = th * pKF->mvScaleFactors[nPredictedLevel];
        Q.min_level = (int16_t)(nPredictedLevel - 1); Q.max_level = (int16_t)nPredictedLevel;
        Q.flags = ORBM_Q_VALID;
        const cv::Mat dMP = pMP->GetDescriptor();
        std::memcpy(&qd[(size_t)i * 32], dMP.data, 32);
    }
    // the key frame as GetFeaturesInArea(x, y, r, bRight) and the candidate loop index it: mvKeysUn (+ mvuRight), or one side of a rig
    orbslam3_hip::FrameView V;
    V.grid = orbm_grid_params{(float)pKF->mnMinX, (float)pKF->mnMinY, pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv};
    int idxOffset = 0;
    if (pKF->NLeft == -1) {
        V.N = (int)pKF->mvKeysUn.size();
        V.keysUn = (const orb_keypoint*)pKF->mvKeysUn.data();
        V.descriptors = pKF->mDescriptors.data;
        V.uRight = pKF->mvuRight.data();
    } else if (!bRight) {
        V.N = pKF->NLeft;
        V.keysUn = (const orb_keypoint*)pKF->mvKeys.data();
        V.descriptors = pKF->mDescriptors.data;
    } else {
        V.N = (int)pKF->mvKeysRight.size();
        V.keysUn = (const orb_keypoint*)pKF->mvKeysRight.data();
        V.descriptors = pKF->mDescriptors.data + (size_t)pKF->NLeft * 32;   // `if(bRight) idx += pKF->NLeft;` :1817
        idxOffset = pKF->NLeft;
    }
    std::vector<int> bestIdx, bestDist;
    device_matcher(mfNNratio, mbCheckOrientation).Fuse(V, q, qd, pKF->mvInvLevelSigma2.data(), (int)pKF->mvInvLevelSigma2.size(), bestIdx, bestDist);
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {
        MapPoint* pMP = vpMapPoints[i];
        if (!pMP || bestIdx[i] < 0) continue;          // no candidate at all, or bestDist > TH_LOW (:1828)
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;   // what :1678-1687 sees at this point's turn
        const int idx = bestIdx[i] + idxOffset;
        MapPoint* pMPinKF = pKF->GetMapPoint(idx);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                else pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, idx);
            pKF->AddMapPoint(pMP, idx);
        }
        nFused++;
    }
    return nFused;
} ORBHIP_GLUE_CATCH("ORBmatcher::Fuse", return 0;)

// ---- Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)   ORBmatcher.cc:1881-2006 (loop closing / map merging: fuse the points seen from the other side) ----
// Gates and projection (:1883-1935) are the loop the Sim3 SearchByProjection overloads share (gather_sim3: the same statements with
// spAlreadyFound = pKF->GetMapPoints(), taken once before the loop as in the reference); one device search without the chi2 gate; the scatter
// (:1975-1990) in index order, pKF->GetMapPoint(bestIdx) read at each point's turn (an earlier iteration may have filled the feature).
int ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint) try {
    const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
    const std::vector<MapPoint*> already(spAlreadyFound.begin(), spAlreadyFound.end());
    const Sim3Gather S = gather_sim3(pKF, Scw, vpPoints, already, (int)th, th);
    std::vector<uint8_t> occ;
    const std::vector<MapPoint*> none(pKF->mvKeysUn.size(), (MapPoint*)NULL);
    const orbslam3_hip::FrameView V = keyframe_view(pKF, occ, none);
    std::vector<int> bestIdx, bestDist;
    device_matcher(mfNNratio, mbCheckOrientation).Fuse(V, S.q, S.qd, nullptr, 0, bestIdx, bestDist);
    int nFused = 0;
    for (size_t k = 0; k < S.src.size(); k++) {
        if (bestIdx[k] < 0) continue;                  // no candidate, or bestDist > TH_LOW (:1975)
        const int iMP = S.src[k];
        MapPoint* pMP = vpPoints[iMP];
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx[k]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bestIdx[k]);
            pKF->AddMapPoint(pMP, bestIdx[k]);
        }
        nFused++;
    }
    return nFused;
} ORBHIP_GLUE_CATCH("ORBmatcher::Fuse", return 0;)

}  // namespace ORB_SLAM3
#endif  // ORBHIP_WITH_ORBSLAM3
