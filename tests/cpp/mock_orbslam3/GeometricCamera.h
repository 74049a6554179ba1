// TEST INFRASTRUCTURE (see README.md): CameraModels/GeometricCamera.h — the members the glue uses; `Pinhole` as the concrete model.
#pragma once
#include <opencv2/core/core.hpp>
#include <Eigen/Core>
#include <vector>
namespace ORB_SLAM3 {
class GeometricCamera {
public:
    virtual ~GeometricCamera() {}
    virtual cv::Point2f project(const cv::Point3f& p3D) = 0;
    virtual cv::Point2f project(const cv::Mat& m3D) = 0;
    virtual cv::Mat toK() = 0;
    virtual float uncertainty2(const Eigen::Matrix<double, 2, 1>& p2D) { (void)p2D; return 1.0f; }   // Pinhole.cpp:56-59, KannalaBrandt8.cpp:86-95
    float getParameter(const int i) { return mvParameters[i]; }
    size_t size() { return mvParameters.size(); }
    unsigned int GetType() { return mnType; }
    const unsigned int CAM_PINHOLE = 0, CAM_FISHEYE = 1;
    std::vector<float> mvParameters;
    unsigned int mnType = 0;
};
class Pinhole : public GeometricCamera {   // CameraModels/Pinhole.cpp:43-66
public:
    Pinhole(float fx, float fy, float cx, float cy) { mvParameters = {fx, fy, cx, cy}; mnType = 0; }
    cv::Point2f project(const cv::Point3f& p) override { return cv::Point2f(mvParameters[0] * p.x / p.z + mvParameters[2], mvParameters[1] * p.y / p.z + mvParameters[3]); }
    cv::Point2f project(const cv::Mat& m) override { const float* p = m.ptr<float>(); return project(cv::Point3f(p[0], p[1], p[2])); }
    cv::Mat toK() override {   // Pinhole.cpp:149-153
        cv::Mat K = cv::Mat::eye(3, 3, CV_32F);
        K.at<float>(0, 0) = mvParameters[0]; K.at<float>(0, 2) = mvParameters[2]; K.at<float>(1, 1) = mvParameters[1]; K.at<float>(1, 2) = mvParameters[3];
        return K;
    }
};
class KannalaBrandt8 : public GeometricCamera {   // CameraModels/KannalaBrandt8.cpp:41-66 (float project), toK :282-286
public:
    KannalaBrandt8(const std::vector<float>& p) { mvParameters = p; mnType = 1; }
    cv::Point2f project(const cv::Point3f& p3D) override {
        const float x2_plus_y2 = p3D.x * p3D.x + p3D.y * p3D.y;
        const float theta = atan2f(sqrtf(x2_plus_y2), p3D.z);
        const float psi = atan2f(p3D.y, p3D.x);
        const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
        const float r = theta + mvParameters[4] * theta3 + mvParameters[5] * theta5 + mvParameters[6] * theta7 + mvParameters[7] * theta9;
        return cv::Point2f(mvParameters[0] * r * cos(psi) + mvParameters[2], mvParameters[1] * r * sin(psi) + mvParameters[3]);
    }
    cv::Point2f project(const cv::Mat& m) override { const float* p = m.ptr<float>(); return project(cv::Point3f(p[0], p[1], p[2])); }
    cv::Mat toK() override {
        cv::Mat K = cv::Mat::eye(3, 3, CV_32F);
        K.at<float>(0, 0) = mvParameters[0]; K.at<float>(0, 2) = mvParameters[2]; K.at<float>(1, 1) = mvParameters[1]; K.at<float>(1, 2) = mvParameters[3];
        return K;
    }
};
}  // namespace ORB_SLAM3
