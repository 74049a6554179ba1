// orbslam3_hip/Frame.h — adapter for the ORB_SLAM3::Frame constructor steps that sit between ORBextractor and ORBmatcher
// (reference src/Frame.cc: UndistortKeyPoints :874-925, ComputeImageBounds :926-953 with the grid scalars :394-397,
// ComputeStereoFromRGBD :1136-1157, ComputeStereoFishEyeMatches :1281-1325) over liborbhip.so (include/orbhip.h, "Frame constructor steps").
//
// Host-vector form for a drop-in Frame; a device-resident pipeline calls orbf_* directly on the extractor's output slabs.
#ifndef ORBSLAM3_HIP_FRAME_H
#define ORBSLAM3_HIP_FRAME_H
#include <stdexcept>
#include <vector>

#include "ORBmatcher.h"

namespace orbslam3_hip {

class FrameOps {
public:
    // K = Pinhole::toK(), distCoef = mDistCoef (4 or 5 entries: k1, k2, p1, p2[, k3])
    FrameOps(float fx, float fy, float cx, float cy, const std::vector<float>& distCoef, int cols, int rows) : cols_(cols), rows_(rows) {
        cam_.fx = fx; cam_.fy = fy; cam_.cx = cx; cam_.cy = cy;
        for (int i = 0; i < 5; i++) cam_.dist[i] = i < (int)distCoef.size() ? distCoef[i] : 0.0f;
        float b[4];
        if (orbf_image_bounds(&cam_, cols, rows, b, &grid_) != ORB_OK) throw std::runtime_error("orbf_image_bounds failed");
        mnMinX = b[0]; mnMaxX = b[1]; mnMinY = b[2]; mnMaxY = b[3];
        mfGridElementWidthInv = grid_.grid_w_inv; mfGridElementHeightInv = grid_.grid_h_inv;
    }

    // Frame::UndistortKeyPoints: mvKeysUn from mvKeys
    void UndistortKeyPoints(const std::vector<orb_keypoint>& mvKeys, std::vector<orb_keypoint>& mvKeysUn) {
        const int N = (int)mvKeys.size();
        mvKeysUn.resize(N);
        if (N == 0) return;
        orb_keypoint* d = kps_.upload(mvKeys.data(), (size_t)N);
        int32_t* dn = cnt_.upload(&N, 1);
        if (orbf_undistort_keypoints(d, dn, 1, N, 1, &cam_, d, nullptr) != ORB_OK) throw std::runtime_error("orbf_undistort_keypoints failed");
        if (orb_memcpy_d2h(mvKeysUn.data(), d, (size_t)N * sizeof(orb_keypoint), nullptr) != ORB_OK || orb_stream_sync(nullptr) != ORB_OK)
            throw std::runtime_error("orbf_undistort_keypoints: copy back failed");
    }

    // Frame::ComputeStereoFromRGBD: imDepth = rows x cols float32 (row stride in floats)
    void ComputeStereoFromRGBD(const std::vector<orb_keypoint>& mvKeys, const std::vector<orb_keypoint>& mvKeysUn, const float* imDepth, int rowStride,
                               float mbf, std::vector<float>& mvuRight, std::vector<float>& mvDepth) {
        const int N = (int)mvKeys.size();
        mvuRight.assign(N, -1.0f); mvDepth.assign(N, -1.0f);
        if (N == 0) return;
        orb_keypoint* d = kps_.upload(mvKeys.data(), (size_t)N);
        orb_keypoint* du = kpsUn_.upload(mvKeysUn.data(), (size_t)N);
        int32_t* dn = cnt_.upload(&N, 1);
        float* dd = depth_.upload(imDepth, (size_t)rowStride * rows_);
        float* out = (float*)out_.ensure((size_t)2 * N * sizeof(float));
        if (orbf_stereo_from_rgbd(d, du, dn, 1, N, 1, dd, (size_t)rowStride * rows_, rowStride, cols_, rows_, mbf, out, out + N, nullptr) != ORB_OK)
            throw std::runtime_error("orbf_stereo_from_rgbd failed");
        if (orb_memcpy_d2h(mvuRight.data(), out, (size_t)N * sizeof(float), nullptr) != ORB_OK ||
            orb_memcpy_d2h(mvDepth.data(), out + N, (size_t)N * sizeof(float), nullptr) != ORB_OK || orb_stream_sync(nullptr) != ORB_OK)
            throw std::runtime_error("orbf_stereo_from_rgbd: copy back failed");
    }

    // FrameView::grid for the matcher adapter (mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv)
    const orbm_grid_params& grid() const { return grid_; }

    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;

private:
    orbf_camera cam_{};
    orbm_grid_params grid_{};
    int cols_, rows_;
    detail::DevBuf kps_, kpsUn_, cnt_, depth_, out_;
};

// Frame::ComputeStereoFishEyeMatches (Frame.cc:1281-1325) for a KannalaBrandt8 stereo rig, host-vector form.  mvKeys / mvKeysRight, mDescriptors /
// mDescriptorsRight and monoLeft / monoRight as the two extractors returned them; rig = camera parameters, mRlr / mtlr, mvLevelSigma2.
// mvStereo3Dpoints comes back as 3 floats per left keypoint (valid where mvLeftToRightMatch[i] >= 0).  Returns nMatches.
class FisheyeStereoMatcher {
public:
    explicit FisheyeStereoMatcher(const orbf_fisheye_rig& rig) : rig_(rig) {}
    int ComputeStereoFishEyeMatches(const std::vector<orb_keypoint>& mvKeys, const uint8_t* mDescriptors, int monoLeft,
                                    const std::vector<orb_keypoint>& mvKeysRight, const uint8_t* mDescriptorsRight, int monoRight,
                                    std::vector<int>& mvLeftToRightMatch, std::vector<int>& mvRightToLeftMatch, std::vector<float>& mvDepth,
                                    std::vector<float>& mvStereo3Dpoints) {
        const int nl = (int)mvKeys.size(), nr = (int)mvKeysRight.size();
        mvLeftToRightMatch.assign(nl, -1); mvRightToLeftMatch.assign(nr, -1); mvDepth.assign(nl, -1.0f); mvStereo3Dpoints.assign((size_t)nl * 3, 0.0f);
        if (nl == 0 || nr == 0) return 0;
        const orb_keypoint* dkl = kl_.upload(mvKeys.data(), nl);
        const orb_keypoint* dkr = kr_.upload(mvKeysRight.data(), nr);
        const uint8_t* ddl = dl_.upload(mDescriptors, (size_t)nl * 32);
        const uint8_t* ddr = dr_.upload(mDescriptorsRight, (size_t)nr * 32);
        const int32_t c[4] = {nl, monoLeft, nr, monoRight};
        const int32_t* dc = cnt_.upload(c, 4);
        int32_t* l2r = (int32_t*)o1_.ensure((size_t)nl * 4);
        int32_t* r2l = (int32_t*)o2_.ensure((size_t)nr * 4);
        float* dep = (float*)o3_.ensure((size_t)nl * 4);
        float* p3d = (float*)o4_.ensure((size_t)nl * 12);
        int32_t* dn = (int32_t*)o5_.ensure(16);
        if (orbf_stereo_fisheye_matches(dkl, ddl, dc, dc + 1, dkr, ddr, dc + 2, dc + 3, nl, nr, 1, 1, &rig_, l2r, r2l, dep, p3d, dn, nullptr) != ORB_OK)
            throw std::runtime_error("orbf_stereo_fisheye_matches failed");
        int n = 0;
        orb_memcpy_d2h(mvLeftToRightMatch.data(), l2r, (size_t)nl * 4, nullptr);
        orb_memcpy_d2h(mvRightToLeftMatch.data(), r2l, (size_t)nr * 4, nullptr);
        orb_memcpy_d2h(mvDepth.data(), dep, (size_t)nl * 4, nullptr);
        orb_memcpy_d2h(mvStereo3Dpoints.data(), p3d, (size_t)nl * 12, nullptr);
        orb_memcpy_d2h(&n, dn, 4, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        return n;
    }

private:
    orbf_fisheye_rig rig_;
    detail::DevBuf kl_, kr_, dl_, dr_, cnt_, o1_, o2_, o3_, o4_, o5_;
};

}  // namespace orbslam3_hip
#endif
