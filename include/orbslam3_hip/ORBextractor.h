// orbslam3_hip/ORBextractor.h — header-only adapter with the signature of ORB_SLAM3::ORBextractor
// (reference include/ORBextractor.h:49-83) forwarding to liborbhip.so (include/orbhip.h, stage 1).
//
// Drop-in use inside the reference: compile with -DORBHIP_WITH_OPENCV, include this header instead
// of "ORBextractor.h" and alias `namespace ORB_SLAM3 { using ORBextractor = orbslam3_hip::ORBextractor; }`.
// Without OpenCV the POD overload `extract()` offers the same call on plain buffers (used by tests/test_cpp_adapter).
#ifndef ORBSLAM3_HIP_ORBEXTRACTOR_H
#define ORBSLAM3_HIP_ORBEXTRACTOR_H
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../orbhip.h"
#include "GlueGuard.h"
#ifdef ORBHIP_WITH_OPENCV
#include <opencv2/core/core.hpp>
#endif

namespace orbslam3_hip {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };  // ORBextractor.h:53

    // ORBextractor.h:49-50
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
        : cfg_{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST}, device_(device) {
        // the scale tables do not depend on the image size: take them from a handle of the reference's default geometry
        // when that is valid, otherwise they are filled on the first call
        orbx_handle h = nullptr;
        if (orbx_create(&cfg_, 752, 480, 1, device_, &h) == ORB_OK) { loadTables(h); orbx_destroy(h); }
    }
    ~ORBextractor() { if (h_) orbx_destroy(h_); if (stereoBuf_) orb_dev_free(stereoBuf_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // POD form of operator() (ORBextractor.cc:1074-1156): returns monoIndex, or -1 for an empty image.
    int extract(const uint8_t* image, int width, int height, int stride, std::vector<orb_keypoint>& keypoints,
                std::vector<uint8_t>& descriptors, const std::vector<int>& vLappingArea) {
        const orb_keypoint* k = nullptr; const uint8_t* d = nullptr;
        int n = 0;
        const int mono = extractView(image, width, height, stride, vLappingArea, k, d, n);
        keypoints.assign(k, k + n);   // (n == 0: both empty)
        descriptors.assign(d, d + (size_t)n * 32);
        return mono;
    }

    // mvImagePyramid on the host.  The reference fills it on every call (ORBextractor.cc:1158-1183) and exactly one function outside the extractor
    // reads it: Frame::ComputeStereoMatches (Frame.cc:962,1052,1071).  With integration/Frame_hip.cc linked that function runs on the device and
    // nobody reads the host copy, so Frame_hip.cc switches the default off (hostPyramidDefault()); otherwise it is on, as in the reference.  When on,
    // the 19-px bordered pyramid comes down as one pinned slab under the call's own kernels (orbx_set_host_pyramid) and mvImagePyramid[l] are cv::Mat
    // HEADERS over that slab: no allocation per call, contents replaced by the next call (the reference's are replaced by the next call too).
    static bool& hostPyramidDefault() { static bool v = true; return v; }
    void setKeepHostPyramid(bool keep) { keepHostPyr_ = keep; if (h_) applyHostPyramid(); }
    bool keepHostPyramid() const { return keepHostPyr_; }

#ifdef ORBHIP_WITH_OPENCV
    // ORBextractor.h:57-59 — identical signature; `_mask` is ignored exactly as in the reference.
    // The reference calls this on two bare std::threads (Frame::ExtractORB, Frame.cc:111-112,1209-1210) where an exception is std::terminate, and its
    // own operator() never throws: a failure below (device lost, allocation) is reported through GlueGuard.h and the call returns what the reference
    // returns for an image it finds nothing in — no key points, released descriptors, -1.  extract() (the flattened form) keeps throwing.
    int operator()(cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint>& _keypoints,
                   cv::OutputArray _descriptors, std::vector<int>& vLappingArea) noexcept {
        try { return call(_image, _mask, _keypoints, _descriptors, vLappingArea); }
        catch (const std::exception& e) { (void)glue_failed("ORBextractor::operator()", e); }
        catch (...) { (void)glue_failed("ORBextractor::operator()"); }
        try { _keypoints.clear(); _descriptors.release(); } catch (...) {}
        return -1;
    }
    // _descriptors.create(n, 32, CV_8U) yields a continuous Mat (a fresh allocation, or the caller's own n x 32 CV_8U Mat of the same size, which
    // cv::Mat::create leaves in place only if it already is that shape): the single memcpy below relies on it, so it is checked.
    int call(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
             cv::OutputArray _descriptors, std::vector<int>& vLappingArea) {
        if (_image.empty()) return -1;
        cv::Mat image = _image.getMat();
        CV_Assert(image.type() == CV_8UC1);  // ORBextractor.cc:1082
        static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint layout");
        const orb_keypoint* k = nullptr; const uint8_t* d = nullptr;
        int n = 0;
        // one copy each, from the handle's pinned output block straight into the caller's containers
        const int mono = extractView(image.data, image.cols, image.rows, (int)image.step, vLappingArea, k, d, n);
        _keypoints.resize((size_t)n);
        if (n) std::memcpy((void*)_keypoints.data(), k, (size_t)n * sizeof(orb_keypoint));
        if (n == 0) _descriptors.release();
        else {
            _descriptors.create(n, 32, CV_8U);
            if (_descriptors.getMat().isContinuous()) std::memcpy(_descriptors.getMat().data, d, (size_t)n * 32);
            else {   // (an ROI header of a wider Mat handed in as the output: row by row through its step)
                cv::Mat dm = _descriptors.getMat();
                for (int r = 0; r < dm.rows; r++) std::memcpy(dm.ptr(r), d + (size_t)r * 32, 32);
            }
        }
        if (keepHostPyr_) {
            // headers over the handle's slab, each level the interior of its bordered parent exactly like ORBextractor.cc:1164-1179; rebuilt only when
            // the handle (image size) changed or somebody resized the member
            const uint8_t* p0 = nullptr;
            if (orbx_host_pyramid_level(h_, 0, &p0, nullptr, nullptr, nullptr) != ORB_OK) throw std::runtime_error(std::string("orbx_host_pyramid_level: ") + orbx_last_error(h_));
            if ((int)mvImagePyramid.size() != cfg_.nlevels || mvImagePyramid[0].data != p0) {
                mvImagePyramid.resize(cfg_.nlevels);
                for (int l = 0; l < cfg_.nlevels; l++) {
                    const uint8_t* p = nullptr; int w = 0, hgt = 0, st = 0;
                    orbx_host_pyramid_level(h_, l, &p, &w, &hgt, &st);
                    mvImagePyramid[l] = cv::Mat(hgt, w, CV_8UC1, (void*)p, (size_t)st);
                }
            }
        } else if (!mvImagePyramid.empty()) mvImagePyramid.clear();   // never a stale pyramid of an earlier image
        return mono;
    }
    std::vector<cv::Mat> mvImagePyramid;  // ORBextractor.h:83
#endif

    // Frame::ComputeStereoMatches (reference src/Frame.cc:955-1134) for the rectified pair THIS extractor (left image) and `right` just extracted:
    // the SAD refinement reads both extractors' pyramids where the last extract() left them on the device (the reference reads mvImagePyramid of
    // both, Frame.cc:1052,1071).  kps / desc: what the two extract() calls returned.  mvuRight / mvDepth come back sized N with -1 = no match.
    // The Frame constructor's own sequence (ExtractORB x 2, then this, Frame.cc:110-132) finds key points and descriptors still on the device where
    // the two calls wrote them: nothing is uploaded (orbx_stereo_matches_last).  Other inputs (edited key points, a subset) take the upload path.
    void ComputeStereoMatches(ORBextractor& right, const orb_keypoint* kpsL, const uint8_t* descL, int nL, const orb_keypoint* kpsR, const uint8_t* descR, int nR,
                              float mb, float mbf, std::vector<float>& mvuRight, std::vector<float>& mvDepth) {
        mvuRight.assign(nL, -1.0f); mvDepth.assign(nL, -1.0f);
        if (nL == 0 || nR == 0) return;
        if (!h_ || !right.h_) throw std::runtime_error("ComputeStereoMatches: both extractors must have extracted their image first");
        if (nL == lastN_ && nR == right.lastN_ && sameAsLast(kpsL, descL, nL) && right.sameAsLast(kpsR, descR, nR)) {
            int n = 0;
            const int rc = orbx_stereo_matches_last(h_, right.h_, mb, mbf, mvuRight.data(), mvDepth.data(), nL, &n);
            if (rc == ORB_OK && n == nL) return;
            if (rc != ORB_E_INVALID) throw std::runtime_error(std::string("orbx_stereo_matches_last: ") + orbx_last_error(h_));
            mvuRight.assign(nL, -1.0f); mvDepth.assign(nL, -1.0f);   // (handles of different configurations: the general path below)
        }
        const int cap = nL > nR ? nL : nR;
        // one device block: [kps L | kps R | desc L | desc R | counts L, R (2 x int32 each) | u_right | depth | work], every section on a 256-byte
        // boundary (the kernels read descriptors with 16-byte vector loads; 28-byte key point records would leave them 8-byte aligned at best)
        size_t off = 0;
        auto sec = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
        const size_t oKL = sec((size_t)cap * sizeof(orb_keypoint)), oKR = sec((size_t)cap * sizeof(orb_keypoint)), oDL = sec((size_t)cap * 32), oDR = sec((size_t)cap * 32),
                     oC = sec(16), oU = sec((size_t)cap * 4), oP = sec((size_t)cap * 4), oW = sec((size_t)cap * 4);
        (void)oKL;
        const size_t need = off;
        if (need > stereoBytes_) {
            if (stereoBuf_) orb_dev_free(stereoBuf_);
            stereoBuf_ = nullptr; stereoBytes_ = 0;
            if (orb_dev_alloc(device_, need, &stereoBuf_) != ORB_OK) throw std::runtime_error("ComputeStereoMatches: device allocation failed");
            stereoBytes_ = need;
        }
        unsigned char* d = (unsigned char*)stereoBuf_;
        orb_keypoint *dkl = (orb_keypoint*)d, *dkr = (orb_keypoint*)(d + oKR);
        uint8_t *ddl = d + oDL, *ddr = d + oDR;
        int32_t* dc = (int32_t*)(d + oC);
        float *dur = (float*)(d + oU), *ddp = (float*)(d + oP);
        int32_t* dwork = (int32_t*)(d + oW);
        const int32_t cnt[4] = {nL, 0, nR, 0};
        bool ok = orb_memcpy_h2d(dkl, kpsL, (size_t)nL * sizeof(orb_keypoint), nullptr) == ORB_OK && orb_memcpy_h2d(dkr, kpsR, (size_t)nR * sizeof(orb_keypoint), nullptr) == ORB_OK &&
                  orb_memcpy_h2d(ddl, descL, (size_t)nL * 32, nullptr) == ORB_OK && orb_memcpy_h2d(ddr, descR, (size_t)nR * 32, nullptr) == ORB_OK &&
                  orb_memcpy_h2d(dc, cnt, sizeof(cnt), nullptr) == ORB_OK;
        if (!ok) throw std::runtime_error("ComputeStereoMatches: upload failed");
        if (orbx_stereo_matches(h_, right.h_, dkl, ddl, dc, dkr, ddr, dc + 2, cap, 1, mb, mbf, dur, ddp, dwork, nullptr) != ORB_OK)
            throw std::runtime_error(std::string("orbx_stereo_matches: ") + orbx_last_error(h_));
        if (orb_memcpy_d2h(mvuRight.data(), dur, (size_t)nL * 4, nullptr) != ORB_OK || orb_memcpy_d2h(mvDepth.data(), ddp, (size_t)nL * 4, nullptr) != ORB_OK ||
            orb_stream_sync(nullptr) != ORB_OK)
            throw std::runtime_error("ComputeStereoMatches: copy back failed");
    }

    // host copy of pyramid level `level` of the last call; border = 0 or 19 (reference layout)
    std::vector<uint8_t> pyramidLevel(int level, int border, int& w, int& h) {
        if (!h_ || orbx_pyramid_level(h_, 0, level, nullptr, &w, &h, nullptr) != ORB_OK) throw std::runtime_error("no pyramid");
        std::vector<uint8_t> out((size_t)(w + 2 * border) * (h + 2 * border));
        if (orbx_copy_level(h_, 0, level, border, out.data()) != ORB_OK) throw std::runtime_error(orbx_last_error(h_));
        return out;
    }

    // getters ORBextractor.h:61-81 (return by value like the reference)
    int GetLevels() { return cfg_.nlevels; }
    float GetScaleFactor() { return cfg_.scale_factor; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

private:
    void loadTables(orbx_handle h) {
        const int n = cfg_.nlevels;
        mvScaleFactor.resize(n); mvInvScaleFactor.resize(n); mvLevelSigma2.resize(n); mvInvLevelSigma2.resize(n);
        orbx_get_tables(h, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(), nullptr);
    }
    void ensure(int w, int h) {
        if (h_ && w == w_ && h == hgt_) return;
        if (h_) { orbx_destroy(h_); h_ = nullptr; }
        lastN_ = -1; lastK_ = nullptr; lastD_ = nullptr;
        const int rc = orbx_create(&cfg_, w, h, 1, device_, &h_);
        if (rc != ORB_OK) throw std::runtime_error(std::string("orbx_create: ") + orbx_last_error(nullptr));
        w_ = w; hgt_ = h;
        loadTables(h_);
        applyHostPyramid();
    }
    void applyHostPyramid() {
        if (orbx_set_host_pyramid(h_, keepHostPyr_ ? 1 : 0) != ORB_OK) throw std::runtime_error(std::string("orbx_set_host_pyramid: ") + orbx_last_error(h_));
    }
    // one call of the extractor; k / d point into the handle's pinned output block (valid until the next call)
    int extractView(const uint8_t* image, int width, int height, int stride, const std::vector<int>& vLappingArea, const orb_keypoint*& k, const uint8_t*& d, int& n) {
        k = nullptr; d = nullptr; n = 0;
        if (!image || width <= 0 || height <= 0) return -1;  // ORBextractor.cc:1078-1079
        ensure(width, height);
        int mono = 0;
        const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0, lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
        lastN_ = -1;
        const int rc = orbx_extract_view(h_, image, width, height, stride, lap0, lap1, &k, &d, &n, &mono);
        if (rc == ORB_E_EMPTY_IMAGE) { n = 0; return -1; }
        if (rc != ORB_OK) throw std::runtime_error(std::string("orbx_extract: ") + orbx_last_error(h_));
        lastN_ = n; lastK_ = k; lastD_ = d;
        return mono;
    }
    // are these the key points / descriptors the last call returned (then the device still holds them)?  The pinned block keeps that call's outputs.
    bool sameAsLast(const orb_keypoint* k, const uint8_t* d, int n) const {
        return n == lastN_ && lastK_ && std::memcmp(k, lastK_, (size_t)n * sizeof(orb_keypoint)) == 0 && std::memcmp(d, lastD_, (size_t)n * 32) == 0;
    }
    orbx_config cfg_;
    int device_;
    orbx_handle h_ = nullptr;
    int w_ = 0, hgt_ = 0;
    void* stereoBuf_ = nullptr;
    size_t stereoBytes_ = 0;
    bool keepHostPyr_ = hostPyramidDefault();
    int lastN_ = -1; const orb_keypoint* lastK_ = nullptr; const uint8_t* lastD_ = nullptr;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

}  // namespace orbslam3_hip
#endif
