// TEST INFRASTRUCTURE (see README.md): include/ImuTypes.h — IMU::Bias (:58-84), IMU::Calib (Tcb / Tbc, :87-150) and the public members of
// IMU::Preintegrated (:235-246) the glue reads, plus SetNewBias (ImuTypes.cc:354-365).
#pragma once
#include <opencv2/core/core.hpp>
namespace ORB_SLAM3 {
namespace IMU {
const float GRAVITY_VALUE = 9.81f;
class Bias {
public:
    Bias() : bax(0), bay(0), baz(0), bwx(0), bwy(0), bwz(0) {}
    Bias(const float& b_acc_x, const float& b_acc_y, const float& b_acc_z, const float& b_ang_vel_x, const float& b_ang_vel_y, const float& b_ang_vel_z)
        : bax(b_acc_x), bay(b_acc_y), baz(b_acc_z), bwx(b_ang_vel_x), bwy(b_ang_vel_y), bwz(b_ang_vel_z) {}
    float bax, bay, baz;
    float bwx, bwy, bwz;
};
class Calib {
public:
    cv::Mat Tcb, Tbc;
};
class Preintegrated {
public:
    void SetNewBias(const Bias& bu_) { bu = bu_; nSetNewBias++; }
    Bias GetUpdatedBias() { return bu; }
    float dT = 0;
    cv::Mat C;              // 15x15 CV_32F covariance
    Bias b;                 // the bias the measurements were integrated with
    cv::Mat dR, dV, dP;
    cv::Mat JRg, JVg, JVa, JPg, JPa;
    int nSetNewBias = 0;    // test bookkeeping
private:
    Bias bu;
};
}  // namespace IMU
}  // namespace ORB_SLAM3
