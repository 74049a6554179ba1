// TEST INFRASTRUCTURE (see README.md): include/G2oTypes.h — EdgeInertial as far as the glue uses it: the constructor's information matrix
// (src/G2oTypes.cc:706-725: inverse of the 9x9 covariance block in float, symmetrised, eigenvalues below 1e-12 zeroed) and information().
// The inverse is Gauss-Jordan and the eigen-decomposition cyclic Jacobi here (OpenCV / Eigen are absent); the glue test compares the glue with
// a path that takes the SAME matrix, so only the data flow is under test, not these numerics.
#pragma once
#include <Eigen/Core>
#include <cmath>
#include "ImuTypes.h"
namespace ORB_SLAM3 {
typedef Eigen::Matrix<double, 9, 9> Matrix9d;
class EdgeInertial {
public:
    explicit EdgeInertial(IMU::Preintegrated* pInt) {
        const cv::Mat cvInfo = pInt->C.rowRange(0, 9).colRange(0, 9).inv(cv::DECOMP_SVD);
        double A[9][9], V[9][9];
        for (int r = 0; r < 9; r++) for (int c = 0; c < 9; c++) A[r][c] = 0.5 * ((double)cvInfo.at<float>(r, c) + (double)cvInfo.at<float>(c, r));
        for (int r = 0; r < 9; r++) for (int c = 0; c < 9; c++) V[r][c] = r == c;
        for (int sweep = 0; sweep < 60; sweep++) {
            double off = 0;
            for (int p = 0; p < 9; p++) for (int q = p + 1; q < 9; q++) off += A[p][q] * A[p][q];
            if (off < 1e-300) break;
            for (int p = 0; p < 9; p++)
                for (int q = p + 1; q < 9; q++) {
                    if (A[p][q] == 0.0) continue;
                    const double th = (A[q][q] - A[p][p]) / (2 * A[p][q]), t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1)), cs = 1 / std::sqrt(t * t + 1), sn = t * cs;
                    for (int k = 0; k < 9; k++) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = cs * akp - sn * akq; A[k][q] = sn * akp + cs * akq; }
                    for (int k = 0; k < 9; k++) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = cs * apk - sn * aqk; A[q][k] = sn * apk + cs * aqk; }
                    for (int k = 0; k < 9; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = cs * vkp - sn * vkq; V[k][q] = sn * vkp + cs * vkq; }
                }
        }
        for (int r = 0; r < 9; r++)
            for (int c = 0; c < 9; c++) {
                double s = 0;
                for (int k = 0; k < 9; k++) s += V[r][k] * (A[k][k] < 1e-12 ? 0.0 : A[k][k]) * V[c][k];
                info_(r, c) = s;
            }
    }
    const Matrix9d& information() const { return info_; }
private:
    Matrix9d info_;
};
}  // namespace ORB_SLAM3
