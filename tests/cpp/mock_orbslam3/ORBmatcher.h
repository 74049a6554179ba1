// TEST INFRASTRUCTURE (see README.md): include/ORBmatcher.h:36-112 — the declarations integration/ORBmatcher_hip.cc defines
#pragma once
#include <opencv2/core/core.hpp>
#include <set>
#include <vector>
#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
namespace ORB_SLAM3 {
class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3, const bool bFarPoints = false, const float thFarPoints = 50.0f);
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist);
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th, float ratioHamming = 1.0);
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, const std::vector<KeyFrame*>& vpPointsKFs,
                           std::vector<MapPoint*>& vpMatched, std::vector<KeyFrame*>& vpMatchedKF, int th, float ratioHamming = 1.0);
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10);   // :71
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo,
                               const bool bCoarse = false);   // include/ORBmatcher.h:74
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12,
                     const float th);   // include/ORBmatcher.h:82
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0, const bool bRight = false);   // include/ORBmatcher.h:85
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint);   // include/ORBmatcher.h:88
    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;
protected:
    float RadiusByViewingCos(const float& viewCos);
    float mfNNratio;
    bool mbCheckOrientation;
};
}  // namespace ORB_SLAM3
