// TEST INFRASTRUCTURE (see README.md): include/Frame.h — members used by the glue
#pragma once
#include <opencv2/core/core.hpp>
#include <vector>
#include "GeometricCamera.h"
#include "ORBextractor.h"
#include "MapPoint.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
namespace ORB_SLAM3 {
#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64
class Frame {
public:
    int N = 0;
    float mbf = 0, mb = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    DBoW2::FeatureVector mFeatVec;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    cv::Mat mTcw;
    void SetPose(cv::Mat Tcw) { mTcw = Tcw.clone(); }   // Frame.cc:481-485 (UpdatePoseMatrices is not needed by the glue)
    std::vector<float> mvInvLevelSigma2;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    int mnScaleLevels = 8;
    float mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
    GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
    int Nleft = -1, Nright = -1;
    std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;
    cv::Mat mTlr, mRlr, mtlr, mTrl;
    // the Frame-constructor steps integration/Frame_hip.cc defines (include/Frame.h:104-118, 264-270) and what they touch
    void ComputeStereoMatches();
    void UndistortKeyPoints();
    void ComputeStereoFishEyeMatches();
    ORBextractor *mpORBextractorLeft = nullptr, *mpORBextractorRight = nullptr;
    cv::Mat mK, mDistCoef;
    std::vector<float> mvLevelSigma2;
    int monoLeft = -1, monoRight = -1;
    std::vector<cv::Mat> mvStereo3Dpoints;
    int mnCloseMPs = 0;
};
}  // namespace ORB_SLAM3
