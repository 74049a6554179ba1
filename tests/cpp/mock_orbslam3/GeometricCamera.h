// TEST INFRASTRUCTURE (see README.md): CameraModels/GeometricCamera.h — the members the glue uses; `Pinhole` as the concrete model.
#pragma once
#include <opencv2/core/core.hpp>
#include <vector>
namespace ORB_SLAM3 {
class GeometricCamera {
public:
    virtual ~GeometricCamera() {}
    virtual cv::Point2f project(const cv::Point3f& p3D) = 0;
    virtual cv::Point2f project(const cv::Mat& m3D) = 0;
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
};
}  // namespace ORB_SLAM3
