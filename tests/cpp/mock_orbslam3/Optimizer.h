// TEST INFRASTRUCTURE (see README.md): include/Optimizer.h:58 — the declaration integration/Optimizer_hip.cc defines
#pragma once
#include "KeyFrame.h"
#include "Map.h"
#include "MapPoint.h"
namespace ORB_SLAM3 {
class Optimizer {
public:
    void static LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF);
};
}  // namespace ORB_SLAM3
