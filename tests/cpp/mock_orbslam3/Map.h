// TEST INFRASTRUCTURE (see README.md): include/Map.h — members used by the glue
#pragma once
#include <mutex>
namespace ORB_SLAM3 {
class Map {
public:
    long unsigned int GetInitKFid() { return mnInitKFid; }
    void IncreaseChangeIndex() { mnMapChange++; }
    bool IsInertial() { return mbIsInertial; }
    long unsigned KeyFramesInMap() { return nKeyFrames; }
    std::mutex mMutexMapUpdate;
    long unsigned int mnInitKFid = 0;
    int mnMapChange = 0;
    bool mbIsInertial = false;
    long unsigned nKeyFrames = 0;
};
}  // namespace ORB_SLAM3
