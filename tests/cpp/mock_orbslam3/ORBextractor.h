// TEST INFRASTRUCTURE (see README.md): include/ORBextractor.h as the tree holds it after the stage-1 swap: the reference's class name
// aliased to the adapter with the reference's signature.
#pragma once
#include <orbslam3_hip/ORBextractor.h>
namespace ORB_SLAM3 {
using ORBextractor = orbslam3_hip::ORBextractor;
}
