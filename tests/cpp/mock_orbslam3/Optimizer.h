// TEST INFRASTRUCTURE (see README.md): include/Optimizer.h:53,58,99 — the declarations integration/Optimizer_hip.cc defines
#pragma once
#include "Frame.h"
#include "KeyFrame.h"
#include "Map.h"
#include "MapPoint.h"
namespace ORB_SLAM3 {
class Optimizer {
public:
    void static LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF);
    int static PoseOptimization(Frame* pFrame);   // include/Optimizer.h:53
    void static LocalInertialBA(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, bool bLarge = false, bool bRecInit = false);   // include/Optimizer.h:99
};
}  // namespace ORB_SLAM3
