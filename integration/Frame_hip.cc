// integration/Frame_hip.cc — the Frame-constructor steps between ORBextractor and ORBmatcher under the reference's member signatures, over
// liborbhip.so: Frame::ComputeStereoMatches() (reference include/Frame.h, src/Frame.cc:955-1134; called by the stereo constructor :132),
// Frame::UndistortKeyPoints() (:874-924; every constructor) and Frame::ComputeStereoFishEyeMatches() (:1281-1325; the fisheye-rig constructor).
//
// Drop-in for the same-named bodies of src/Frame.cc: compile this file INSIDE the ORB-SLAM3 tree instead of them (integration/README.md), with
// -DORBHIP_WITH_ORBSLAM3 -DORBHIP_WITH_OPENCV and `ORB_SLAM3::ORBextractor` aliased to orbslam3_hip::ORBextractor (Stage 1).
//   * ComputeStereoMatches: the reference searches row bands, matches descriptors, refines by 11x11 SAD on mvImagePyramid of both extractors and
//     culls by the median SAD on one core; here the two extractors' pyramids are still on the device from the ExtractORB calls just before
//     (Frame.cc:110-114), so one call computes mvuRight / mvDepth there and only N floats x 2 come back.
//   * UndistortKeyPoints: cv::undistortPoints(mat, mat, toK(), mDistCoef, cv::Mat(), mK) restated on the device (R = I, P = K; mK is built from the
//     same intrinsics as Pinhole::toK() by the callers in Tracking, so P == K).
//   * ComputeStereoFishEyeMatches: BFMatcher::knnMatch(k = 2) + Lowe ratio 0.7 + KannalaBrandt8::TriangulateMatches per pair, one launch.
// Not taken over: ComputeStereoFromRGBD (:1136-1157) — N look-ups into a host depth image; uploading the image would cost more than the loop.
#ifdef ORBHIP_WITH_ORBSLAM3
#include "Frame.h"

#include <cstring>
#include <memory>
#include <vector>

#include <orbslam3_hip/Frame.h>
#include <orbslam3_hip/GlueGuard.h>

namespace ORB_SLAM3 {

static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint is read as orb_keypoint (pt.x, pt.y, size, angle, response, octave, class_id)");

// With this file linked, Frame::ComputeStereoMatches — the only reader of ORBextractor::mvImagePyramid outside the extractor (src/Frame.cc:962,1052,
// 1071) — runs on the device: extractors constructed from here on do not bring the pyramid to the host (setKeepHostPyramid(true) turns it back on
// for a consumer of the member that is not in the reference).  Runs before main(); Tracking builds its extractors long after.
static const bool kHostPyramidOff = (orbslam3_hip::ORBextractor::hostPyramidDefault() = false);

void Frame::ComputeStereoMatches() try {
    // mvuRight / mvDepth sized N, -1 = no match (:957-958); everything between :960 and :1133 runs on the device
    const int nR = (int)mvKeysRight.size();
    mpORBextractorLeft->ComputeStereoMatches(*mpORBextractorRight, reinterpret_cast<const orb_keypoint*>(mvKeys.data()), mDescriptors.data, N,
                                             reinterpret_cast<const orb_keypoint*>(mvKeysRight.data()), mDescriptorsRight.data, nR, mb, mbf, mvuRight, mvDepth);
} ORBHIP_GLUE_CATCH("Frame::ComputeStereoMatches", { mvuRight = std::vector<float>(N, -1.0f); mvDepth = std::vector<float>(N, -1.0f); return; })   // "no stereo match" (:957-958): the constructor indexes both

void Frame::UndistortKeyPoints() try {
    if (mDistCoef.at<float>(0) == 0.0) {   // :879-883
        mvKeysUn = mvKeys;
        return;
    }
    // one FrameOps (camera record + device buffers) per calling thread, rebuilt when the calibration changes
    struct Cached { float k[4]; std::vector<float> d; std::unique_ptr<orbslam3_hip::FrameOps> ops; };
    thread_local Cached C;
    const cv::Mat K = static_cast<Pinhole*>(mpCamera)->toK();
    const float k[4] = {K.at<float>(0, 0), K.at<float>(1, 1), K.at<float>(0, 2), K.at<float>(1, 2)};
    std::vector<float> d;
    for (int i = 0; i < mDistCoef.rows * mDistCoef.cols && i < 5; i++) d.push_back(mDistCoef.at<float>(i));
    if (!C.ops || std::memcmp(C.k, k, sizeof(k)) != 0 || C.d != d) {
        C.ops.reset(new orbslam3_hip::FrameOps(k[0], k[1], k[2], k[3], d, 640, 480));   // the image size only feeds the bounds, which are not used here
        std::memcpy(C.k, k, sizeof(k));
        C.d = d;
    }
    std::vector<orb_keypoint> in(N), out;
    if (N) std::memcpy(in.data(), mvKeys.data(), (size_t)N * sizeof(orb_keypoint));
    C.ops->UndistortKeyPoints(in, out);
    mvKeysUn.resize(N);   // every other attribute of the key point is the distorted one's (:913-921)
    if (N) std::memcpy((void*)mvKeysUn.data(), out.data(), (size_t)N * sizeof(orb_keypoint));
} ORBHIP_GLUE_CATCH("Frame::UndistortKeyPoints", { mvKeysUn = mvKeys; return; })   // sized like mvKeys, as every later step assumes

void Frame::ComputeStereoFishEyeMatches() try {
    orbf_fisheye_rig rig{};
    for (int i = 0; i < 8; i++) { rig.k_left[i] = mpCamera->getParameter(i); rig.k_right[i] = mpCamera2->getParameter(i); }
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) rig.R_lr[r * 3 + c] = mRlr.at<float>(r, c); rig.t_lr[r] = mtlr.at<float>(r); }
    for (size_t i = 0; i < mvLevelSigma2.size() && i < 16; i++) rig.level_sigma2[i] = mvLevelSigma2[i];
    struct Cached { orbf_fisheye_rig rig; std::unique_ptr<orbslam3_hip::FisheyeStereoMatcher> m; };
    thread_local Cached C;
    if (!C.m || std::memcmp(&C.rig, &rig, sizeof(rig)) != 0) { C.m.reset(new orbslam3_hip::FisheyeStereoMatcher(rig)); C.rig = rig; }

    std::vector<orb_keypoint> kl(mvKeys.size()), kr(mvKeysRight.size());
    if (!kl.empty()) std::memcpy(kl.data(), mvKeys.data(), kl.size() * sizeof(orb_keypoint));
    if (!kr.empty()) std::memcpy(kr.data(), mvKeysRight.data(), kr.size() * sizeof(orb_keypoint));
    std::vector<float> p3d;
    C.m->ComputeStereoFishEyeMatches(kl, mDescriptors.data, monoLeft, kr, mDescriptorsRight.data, monoRight, mvLeftToRightMatch, mvRightToLeftMatch, mvDepth, p3d);
    // :1292-1297: the five vectors are (re)sized to Nleft / Nright whatever the matcher found
    mvLeftToRightMatch.resize(Nleft, -1);
    mvRightToLeftMatch.resize(Nright, -1);
    mvDepth.resize(Nleft, -1.0f);
    mvuRight = std::vector<float>(Nleft, -1);
    mvStereo3Dpoints = std::vector<cv::Mat>(Nleft);
    mnCloseMPs = 0;
    for (int i = 0; i < Nleft; i++)
        if (mvLeftToRightMatch[i] >= 0) {
            cv::Mat p(3, 1, CV_32F);
            for (int c = 0; c < 3; c++) p.at<float>(c) = p3d[(size_t)i * 3 + c];
            mvStereo3Dpoints[i] = p;
        }
} ORBHIP_GLUE_CATCH("Frame::ComputeStereoFishEyeMatches", {   // "nothing matched", sized as :1292-1297 leave them
    mvLeftToRightMatch = std::vector<int>(Nleft, -1); mvRightToLeftMatch = std::vector<int>(Nright, -1); mvDepth = std::vector<float>(Nleft, -1.0f);
    mvuRight = std::vector<float>(Nleft, -1); mvStereo3Dpoints = std::vector<cv::Mat>(Nleft); mnCloseMPs = 0; return; })

}  // namespace ORB_SLAM3
#endif  // ORBHIP_WITH_ORBSLAM3
