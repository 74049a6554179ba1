// TEST INFRASTRUCTURE (see README.md): include/KeyFrame.h — members used by the glue
#pragma once
#include <opencv2/core/core.hpp>
#include <algorithm>
#include <set>
#include <vector>
#include "GeometricCamera.h"
#include "ImuTypes.h"
#include "Map.h"
#include "MapPoint.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
namespace ORB_SLAM3 {
class KeyFrame {
public:
    KeyFrame(float fx_, float fy_, float cx_, float cy_, float mbf_, int nleft, int minx, int miny, int maxx, int maxy, float gwInv, float ghInv,
             const std::vector<float>& scale, const std::vector<float>& invSigma2)
        : mnGridCols(64), mnGridRows(48), mfGridElementWidthInv(gwInv), mfGridElementHeightInv(ghInv), fx(fx_), fy(fy_), cx(cx_), cy(cy_), mbf(mbf_),
          mvScaleFactors(scale), mvInvLevelSigma2(invSigma2), mnMinX(minx), mnMinY(miny), mnMaxX(maxx), mnMaxY(maxy), NLeft(nleft), NRight(-1) {
        for (float s : invSigma2) mvLevelSigma2.push_back(1.0f / s);
    }
    void SetPose(const cv::Mat& Tcw_) { Tcw = Tcw_.clone(); nPoseSets++; }
    cv::Mat GetPose() { return Tcw.clone(); }
    cv::Mat GetRotation() { return Tcw.rowRange(0, 3).colRange(0, 3).clone(); }                    // KeyFrame.cc:210-214
    cv::Mat GetTranslation() { return Tcw.rowRange(0, 3).col(3).clone(); }                         // :216-220
    cv::Mat GetCameraCenter() { return -GetRotation().t() * GetTranslation(); }                    // Ow = -Rwc * tcw (:70-77, :180-184)
    cv::Mat GetRightRotation() { return mTlr.rowRange(0, 3).colRange(0, 3).t() * GetRotation(); }  // :1289-1296
    cv::Mat GetRightTranslation() {                                                                // :1299-1308
        cv::Mat Rrl = mTlr.rowRange(0, 3).colRange(0, 3).t();
        return Rrl * GetTranslation() + (-Rrl * mTlr.rowRange(0, 3).col(3));
    }
    // inertial state (KeyFrame.cc:161-165, 192-208, 222-226, 916-940)
    cv::Mat GetImuPosition() { cv::Mat Rwc = GetRotation().t(); return Rwc * mImuCalib.Tcb.rowRange(0, 3).col(3) + GetCameraCenter(); }   // Owb (:100-108)
    cv::Mat GetImuRotation() { return GetRotation().t() * mImuCalib.Tcb.rowRange(0, 3).colRange(0, 3); }
    cv::Mat GetVelocity() { return Vw.clone(); }
    void SetVelocity(const cv::Mat& Vw_) { Vw = Vw_.clone(); nVelSets++; }
    void SetNewBias(const IMU::Bias& b) { mImuBias = b; if (mpImuPreintegrated) mpImuPreintegrated->SetNewBias(b); nBiasSets++; }
    cv::Mat GetGyroBias() { cv::Mat m(3, 1, CV_32F); m.at<float>(0) = mImuBias.bwx; m.at<float>(1) = mImuBias.bwy; m.at<float>(2) = mImuBias.bwz; return m; }
    cv::Mat GetAccBias() { cv::Mat m(3, 1, CV_32F); m.at<float>(0) = mImuBias.bax; m.at<float>(1) = mImuBias.bay; m.at<float>(2) = mImuBias.baz; return m; }
    IMU::Bias GetImuBias() { return mImuBias; }
    KeyFrame* mPrevKF = nullptr;
    KeyFrame* mNextKF = nullptr;
    IMU::Preintegrated* mpImuPreintegrated = nullptr;
    IMU::Calib mImuCalib;
    bool bImu = false;
    cv::Mat Vw;
    IMU::Bias mImuBias;
    int nVelSets = 0, nBiasSets = 0;
    std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { return mvpOrderedConnectedKeyFrames; }
    void EraseMapPointMatch(MapPoint* pMP) { for (auto& p : mvpMapPoints) if (p == pMP) p = nullptr; nErased++; }
    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    std::set<MapPoint*> GetMapPoints() { std::set<MapPoint*> s; for (MapPoint* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p); return s; }   // KeyFrame.cc:465-478
    MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
    void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }                  // KeyFrame.cc:437-441
    cv::Mat GetRightCameraCenter() { return -GetRightRotation().t() * GetRightTranslation(); }
    bool IsInImage(const float& x, const float& y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }
    bool isBad() { return mbBad; }
    Map* GetMap() { return mpMap; }
    long unsigned int mnId = 0;
    const int mnGridCols, mnGridRows;
    const float mfGridElementWidthInv, mfGridElementHeightInv;
    long unsigned int mnBALocalForKF = 0, mnBAFixedForKF = 0;
    const float fx, fy, cx, cy, mbf;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn, mvKeysRight;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    const std::vector<float> mvScaleFactors, mvInvLevelSigma2;
    int mnScaleLevels = 8;
    float mfLogScaleFactor = 0;
    const int mnMinX, mnMinY, mnMaxX, mnMaxY;
    GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
    cv::Mat mTrl, mTlr;
    std::vector<float> mvLevelSigma2;
    const int NLeft, NRight;
    // state
    cv::Mat Tcw;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<KeyFrame*> mvpOrderedConnectedKeyFrames;
    bool mbBad = false;
    Map* mpMap = nullptr;
    int nPoseSets = 0, nErased = 0;
};
// MapPoint::PredictScale (MapPoint.cc:513-547)
inline int MapPoint::PredictScale(const float& currentDist, KeyFrame* pKF) {
    const float ratio = mfMaxDistance / currentDist;
    int nScale = (int)std::ceil(std::log(ratio) / pKF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0; else if (nScale >= pKF->mnScaleLevels) nScale = pKF->mnScaleLevels - 1;
    return nScale;
}
}  // namespace ORB_SLAM3
#include "Frame.h"
namespace ORB_SLAM3 {
inline int MapPoint::PredictScale(const float& currentDist, Frame* pF) {
    const float ratio = mfMaxDistance / currentDist;
    int nScale = (int)std::ceil(std::log(ratio) / pF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0; else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
    return nScale;
}
}  // namespace ORB_SLAM3
