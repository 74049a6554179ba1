// TEST INFRASTRUCTURE (see README.md): include/MapPoint.h — members used by the glue
#pragma once
#include <opencv2/core/core.hpp>
#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
namespace ORB_SLAM3 {
class KeyFrame;
class Frame;
class Map;
class MapPoint {
public:
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    void SetWorldPos(const cv::Mat& Pos) { mWorldPos = Pos.clone(); }
    cv::Mat GetNormal() { return mNormalVector.clone(); }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    bool isBad() { return mbBad; }
    int Observations() { return nObs; }
    std::map<KeyFrame*, std::tuple<int, int>> GetObservations() { return mObservations; }
    void EraseObservation(KeyFrame* pKF) { mObservations.erase(pKF); nErased++; }
    std::tuple<int, int> GetIndexInKeyFrame(KeyFrame* pKF) {                                          // MapPoint.cc:393-403
        std::map<KeyFrame*, std::tuple<int, int>>::iterator it = mObservations.find(pKF);
        return it != mObservations.end() ? it->second : std::tuple<int, int>(-1, -1);
    }
    bool IsInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) != 0; }                       // MapPoint.cc:405-409
    void AddObservation(KeyFrame* pKF, int idx) { mObservations[pKF] = std::make_tuple(idx, -1); nObs++; }   // :123-148 (monocular count)
    void Replace(MapPoint* pMP) { if (pMP == this) return; mbBad = true; mpReplaced = pMP; pMP->nObs += nObs; }   // :281-338, as far as Fuse can observe it
    MapPoint* mpReplaced = nullptr;
    void UpdateNormalAndDepth() { nNormalUpdates++; }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    int PredictScale(const float& currentDist, KeyFrame* pKF);   // MapPoint.cc:513-529
    int PredictScale(const float& currentDist, Frame* pF);       // MapPoint.cc:531-547
    Map* GetMap() { return mpMap; }
    long unsigned int mnId = 0;
    static std::mutex mGlobalMutex;   // include/MapPoint.h:112 (defined by the test program)
    // tracking (set by Frame::isInFrustum)
    float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackDepthR = 0, mTrackProjXR = 0, mTrackProjYR = 0;
    bool mbTrackInView = false, mbTrackInViewR = false;
    int mnTrackScaleLevel = 0, mnTrackScaleLevelR = -1;
    float mTrackViewCos = 1, mTrackViewCosR = 1;
    long unsigned int mnBALocalForKF = 0;
    // state
    cv::Mat mWorldPos, mNormalVector, mDescriptor;
    std::map<KeyFrame*, std::tuple<int, int>> mObservations;
    int nObs = 0, nErased = 0, nNormalUpdates = 0;
    bool mbBad = false;
    float mfMinDistance = 0, mfMaxDistance = 1e9f;
    Map* mpMap = nullptr;
};
}  // namespace ORB_SLAM3
