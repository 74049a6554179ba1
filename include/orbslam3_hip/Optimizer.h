// orbslam3_hip/Optimizer.h — adapter for the linearisation inside Optimizer::LocalBundleAdjustment
// (reference include/Optimizer.h:58, src/Optimizer.cc:1811-2523) over liborbhip.so (include/orbhip.h, stage 3).
//
// `LbaLinearizer` owns the flattened window (== the g2o graph of Optimizer.cc:1957-2193).  Window selection
// (:1816-1945), the LM loop and the write-back (:2375-2522) stay in the caller; per iteration it calls buildSystem()
// (== BlockSolver::buildSystem, block_solver.hpp:502-560) or computeErrors() (== computeActiveErrors).
#ifndef ORBSLAM3_HIP_OPTIMIZER_H
#define ORBSLAM3_HIP_OPTIMIZER_H
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <stdexcept>
#include <vector>

#include "../orbhip.h"
#include "ORBmatcher.h"  // detail::DevBuf

namespace orbslam3_hip {

struct LbaHostSystem {  // column-major doubles, g2o block conventions
    std::vector<double> Hpp, bp, Hll, bl, Hpl, err, chi2, rho, depth;
    double robustChi2 = 0;
};

class LbaLinearizer {
public:
    // Converter::toSE3Quat (Converter.cc:34-44): float Tcw (row-major 3x4 or 4x4, `ld` floats per row) -> SE3Quat(R,t) = (t, q)
    static void poseFromTcw(const float* Tcw, int ld, double out7[7]) {
        double m[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m[r * 3 + c] = (double)Tcw[r * ld + c];
        double q[4];  // Eigen Quaterniond(Matrix3d) + SE3Quat::normalizeRotation
        double t = m[0] + m[4] + m[8];
        if (t > 0) { t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t; q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t; }
        else {
            int i = 0;
            if (m[4] > m[0]) i = 1;
            if (m[8] > m[i * 3 + i]) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
            q[i] = 0.5 * t; t = 0.5 / t;
            q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t; q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t; q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        }
        if (q[3] < 0) for (double& v : q) v = -v;
        const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int r = 0; r < 3; r++) out7[r] = (double)Tcw[r * ld + 3];
        for (int c = 0; c < 4; c++) out7[3 + c] = q[c] / n;
    }

    // graph build (Optimizer.cc:1982-2190): vertices first, then edges in insertion order (landmark-major)
    int addPose(const double pose7[7], bool fixed) {
        poses_.insert(poses_.end(), pose7, pose7 + 7);
        hidx_.push_back(fixed ? -1 : nfree_++);
        return (int)hidx_.size() - 1;
    }
    int addPoint(const double xyz[3]) { points_.insert(points_.end(), xyz, xyz + 3); return (int)points_.size() / 3 - 1; }
    int addCamera(const lba_camera& c) { cams_.push_back(c); return (int)cams_.size() - 1; }
    void addEdge(int pose, int point, int kind, int cam, float u, float v, float uR, float invSigma2) {
        if (!edges_.empty() && point < edges_.back().point) throw std::runtime_error("edges must be added landmark-major");
        lba_edge e{pose, point, (int16_t)kind, (int16_t)cam, {u, v, uR}, invSigma2};
        edges_.push_back(e);
        dirty_ = true;
    }
    void setPose(int i, const double pose7[7]) { std::copy(pose7, pose7 + 7, poses_.begin() + 7 * i); posesDirty_ = true; }
    void setPoint(int i, const double xyz[3]) { std::copy(xyz, xyz + 3, points_.begin() + 3 * i); pointsDirty_ = true; }
    int numFreePoses() const { return nfree_; }

    void buildSystem(LbaHostSystem& out) { run(out, true); }
    void computeErrors(LbaHostSystem& out) { run(out, false); }

    // optimizer.optimize(iterations) (Optimizer.cc:2205 / :2290): Levenberg-Marquardt with Schur complement on the device.
    // Poses / points held by this object are updated (read them back with pose(i) / point(i)); returns the iterations run.
    // pbStopFlag is polled between lambda trials like g2o's terminate().
    int optimize(int iterations, const volatile int* pbStopFlag = nullptr, double* finalChi2 = nullptr, const bool* pbStopFlagBool = nullptr) {
        upload();
        const int np = (int)hidx_.size(), nl = (int)points_.size() / 3, ne = (int)edges_.size();
        lba_problem P{};
        fillProblem(P, np, nl, ne);
        void* ws = lm_ws_.ensure(lba_lm_workspace_bytes(&P, 1));
        double stats[4] = {0, 0, 0, 0};
        static_assert(sizeof(bool) == 1, "the reference's bool* pbStopFlag is polled as one byte");
        const int rc = pbStopFlagBool ? lba_optimize_stopflag(&P, 1, iterations, ws, stats, (const volatile unsigned char*)pbStopFlagBool, nullptr)
                                      : lba_optimize(&P, 1, iterations, ws, stats, pbStopFlag, nullptr);
        if (rc == ORB_E_CAPACITY) throw std::length_error("lba_optimize: ORB_E_CAPACITY, the reduced camera system of this window does not fit the device solver (> ~3 300 free key frames)");
        if (rc != ORB_OK && rc != ORB_E_ABORTED) throw std::runtime_error("lba_optimize failed");
        orb_memcpy_d2h(poses_.data(), dPoses_, poses_.size() * 8, nullptr);
        orb_memcpy_d2h(points_.data(), dPoints_, points_.size() * 8, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        if (finalChi2) *finalChi2 = stats[1];
        return (int)stats[0];
    }
    int edgeKind(int i) const { return edges_[i].kind; }           // LBA_EDGE_* of the i-th added edge
    const double* pose(int i) const { return &poses_[7 * i]; }     // t(x y z), q(x y z w) — SE3Quat estimate
    const double* point(int i) const { return &points_[3 * i]; }

private:
    void upload() {
        const int np = (int)hidx_.size(), nl = (int)points_.size() / 3, ne = (int)edges_.size();
        if (dirty_) {  // == BlockSolver::buildStructure: the two CSR views
            std::vector<int32_t> lm(nl + 1, 0), ps(np + 1, 0), pe(ne);
            for (const lba_edge& e : edges_) { lm[e.point + 1]++; ps[e.pose + 1]++; }
            std::partial_sum(lm.begin(), lm.end(), lm.begin());
            std::partial_sum(ps.begin(), ps.end(), ps.begin());
            std::vector<int32_t> fill(ps.begin(), ps.end() - 1);
            for (int i = 0; i < ne; i++) pe[fill[edges_[i].pose]++] = i;
            dEdges_ = e_.upload(edges_.data(), ne); dLm_ = lm_.upload(lm.data(), nl + 1); dPs_ = ps_.upload(ps.data(), np + 1);
            dPe_ = pe_.upload(pe.data(), ne); dH_ = h_.upload(hidx_.data(), np); dCams_ = c_.upload(cams_.data(), cams_.size());
            int32_t cnt[3] = {np, nl, ne};
            dCnt_ = cnt_.upload(cnt, 3);
            posesDirty_ = pointsDirty_ = true; dirty_ = false;
        }
        if (posesDirty_) { dPoses_ = p_.upload(poses_.data(), poses_.size()); posesDirty_ = false; }
        if (pointsDirty_) { dPoints_ = x_.upload(points_.data(), points_.size()); pointsDirty_ = false; }
    }
    void fillProblem(lba_problem& P, int np, int nl, int ne) {
        P.poses = dPoses_; P.pose_hidx = dH_; P.points = dPoints_; P.edges = dEdges_; P.lm_start = dLm_; P.pose_start = dPs_;
        P.pose_edges = dPe_; P.cameras = dCams_; P.n_poses = dCnt_; P.n_points = dCnt_ + 1; P.n_edges = dCnt_ + 2;
        P.cap_p = np; P.cap_l = nl; P.cap_e = ne; P.n_cameras = (int)cams_.size();
        P.huber_mono = (double)std::sqrt(5.991f);    // const float thHuberMono = sqrt(5.991)   Optimizer.cc:2052
        P.huber_stereo = (double)std::sqrt(7.815f);  // const float thHuberStereo = sqrt(7.815) Optimizer.cc:2053
    }
    void run(LbaHostSystem& o, bool full) {
        upload();
        const int np = (int)hidx_.size(), nl = (int)points_.size() / 3, ne = (int)edges_.size();
        lba_problem P{};
        fillProblem(P, np, nl, ne);
        lba_system S{};
        const size_t sz[10] = {(size_t)np * 36, (size_t)np * 6, (size_t)nl * 9, (size_t)nl * 3, (size_t)ne * 18, (size_t)ne * 3, (size_t)ne, (size_t)ne * 2, (size_t)ne, 1};
        double** ptr[10] = {&S.Hpp, &S.bp, &S.Hll, &S.bl, &S.Hpl, &S.err, &S.chi2, &S.rho, &S.depth, &S.robust_chi2_sum};
        std::vector<double>* host[9] = {&o.Hpp, &o.bp, &o.Hll, &o.bl, &o.Hpl, &o.err, &o.chi2, &o.rho, &o.depth};
        for (int i = 0; i < 10; i++) {
            const bool want = full ? i < 9 : i >= 5;
            *ptr[i] = want ? (double*)out_[i].ensure(sz[i] * 8 + 16) : nullptr;
        }
        // the adapter flattened the graph itself: it knows whether the fisheye model, right-camera edges or stereo edges occur
        const int rc = full ? lba_build_system_hint(&P, 1, &S, hints(), nullptr) : lba_compute_errors(&P, 1, &S, nullptr);
        if (rc != ORB_OK) throw std::runtime_error("lba call failed");
        for (int i = 0; i < 9; i++)
            if (*ptr[i]) { host[i]->resize(sz[i]); orb_memcpy_d2h(host[i]->data(), *ptr[i], sz[i] * 8, nullptr); }
        if (S.robust_chi2_sum) orb_memcpy_d2h(&o.robustChi2, S.robust_chi2_sum, 8, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
    }
    std::vector<double> poses_, points_;
    std::vector<int32_t> hidx_;
    unsigned hints() const {   // what the edges this adapter was given allow: monocular pinhole, pinhole (monocular + stereo), or nothing
        bool stereo = false;
        for (const lba_edge& e : edges_) {
            if (e.kind == LBA_EDGE_BODY || cams_[(size_t)e.cam].model != LBA_CAM_PINHOLE) return 0u;
            stereo = stereo || e.kind == LBA_EDGE_STEREO;
        }
        return stereo ? LBA_HINT_PINHOLE : LBA_HINT_MONO_PINHOLE;
    }
    std::vector<lba_edge> edges_;
    std::vector<lba_camera> cams_;
    int nfree_ = 0;
    bool dirty_ = true, posesDirty_ = true, pointsDirty_ = true;
    detail::DevBuf p_, x_, e_, lm_, ps_, pe_, h_, c_, cnt_, out_[10], lm_ws_;
    const double *dPoses_ = nullptr, *dPoints_ = nullptr;
    const lba_edge* dEdges_ = nullptr;
    const int32_t *dLm_ = nullptr, *dPs_ = nullptr, *dPe_ = nullptr, *dH_ = nullptr, *dCnt_ = nullptr;
    const lba_camera* dCams_ = nullptr;
};

// Optimizer::PoseOptimization(Frame* pFrame) (reference include/Optimizer.h:53, src/Optimizer.cc:907-1273) over pose_optimize().
// The caller walks pFrame->mvpMapPoints exactly like Optimizer.cc:966-1127 and adds one observation per map point; optimize()
// returns nInitialCorrespondences - nBad and leaves the pose (-> pFrame->SetPose) and the mvbOutlier flags (in insertion order).
class PoseOptimizer {
public:
    int addCamera(const lba_camera& c) { cams_.push_back(c); return (int)cams_.size() - 1; }
    void clear() { edges_.clear(); }
    void reset() { edges_.clear(); cams_.clear(); }   // a new frame with (possibly) other cameras; the device buffers stay
    // monocular observation (Optimizer.cc:979-1011, or :1059-1089 for the left fisheye camera)
    void addMono(const float Xw[3], float u, float v, float invSigma2, int cam = 0) { add(Xw, u, v, 0.f, invSigma2, LBA_EDGE_MONO, cam); }
    // stereo observation (Optimizer.cc:1013-1049)
    void addStereo(const float Xw[3], float u, float v, float uR, float invSigma2, int cam = 0) { add(Xw, u, v, uR, invSigma2, LBA_EDGE_STEREO, cam); }
    // right fisheye camera observation through mTrl (Optimizer.cc:1091-1122)
    void addBody(const float Xw[3], float u, float v, float invSigma2, int cam) { add(Xw, u, v, 0.f, invSigma2, LBA_EDGE_BODY, cam); }
    int size() const { return (int)edges_.size(); }

    // pose7 (t, q of Tcw as SE3Quat; LbaLinearizer::poseFromTcw converts) is updated in place
    int optimize(double pose7[7], std::vector<bool>& mvbOutlier) {
        const int ne = (int)edges_.size();
        mvbOutlier.assign(ne, false);
        if (ne == 0 || cams_.empty()) return 0;
        const pose_edge* dE = e_.upload(edges_.data(), ne);
        const lba_camera* dC = c_.upload(cams_.data(), cams_.size());
        const double* dP = p_.upload(pose7, 7);
        const int32_t* dN = n_.upload(&ne, 1);
        double* dO = (double*)o_.ensure(7 * 8 + 16);
        uint8_t* dOut = (uint8_t*)f_.ensure((size_t)ne + 16);
        int32_t* dG = (int32_t*)g_.ensure(16);
        bool pinhole = true;   // the adapter was handed the edges: it knows whether the fisheye model or a right-camera edge occurs
        for (const pose_edge& e : edges_)
            if (e.kind == LBA_EDGE_BODY || cams_[(size_t)e.cam].model != LBA_CAM_PINHOLE) { pinhole = false; break; }
        if (pose_optimize_hint(dP, dE, dN, ne, 1, dC, (int)cams_.size(), dO, dOut, dG, pinhole ? LBA_HINT_PINHOLE : 0u, nullptr) != ORB_OK)
            throw std::runtime_error("pose_optimize");
        std::vector<uint8_t> fl(ne);
        int32_t good = 0;
        orb_memcpy_d2h(pose7, dO, 56, nullptr);
        orb_memcpy_d2h(fl.data(), dOut, ne, nullptr);
        orb_memcpy_d2h(&good, dG, 4, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        for (int i = 0; i < ne; i++) mvbOutlier[i] = fl[i] != 0;
        return good;
    }

private:
    void add(const float Xw[3], float u, float v, float uR, float s2, int kind, int cam) {
        edges_.push_back(pose_edge{{Xw[0], Xw[1], Xw[2]}, {u, v, uR}, s2, (int16_t)kind, (int16_t)cam});
    }
    std::vector<pose_edge> edges_;
    std::vector<lba_camera> cams_;
    detail::DevBuf e_, c_, p_, n_, o_, f_, g_;
};

// Optimizer::LocalInertialBA(KeyFrame*, bool*, Map*, bool bLarge, bool bRecInit) (reference include/Optimizer.h:97, src/Optimizer.cc:4753-5365)
// over liba_optimize().  The caller keeps the reference's graph-building walk and calls, in the same order:
//   setRig(...)                    once (ImuCamPose calibration members, G2oTypes.cc:43-66)
//   addKeyFrame(...)               in ascending mnId order: VertexPose (+ VertexVelocity / GyroBias / AccBias when bImu), Optimizer.cc:4899-4960
//   addInertial(...)               per preintegration, newest first (Optimizer.cc:4964-5062); `info` = EdgeInertial::information() as built by the
//                                  reference's constructor (G2oTypes.cc:706-725) and scaled at :5009, infoG / infoA = Optimizer.cc:5024-5043
//   addPoint / addMono / addStereo per map point and observation (Optimizer.cc:5078-5215); observations of one point are contiguous
//   optimize(lambdaInit, its)      lambdaInit = 1e-2 (bLarge) / 1e0, its = opt_it (Optimizer.cc:4884-4896, :5225)
// and then reads keyFrame(i) / point(i) / visualChi2 / depthPositive for the outlier pass and the write-back (Optimizer.cc:5237-5360).
class InertialBA {
public:
    void setRig(const liba_rig& rig) { rig_ = rig; }
    void clear() { kfs_.clear(); pts_.clear(); edges_.clear(); imu_.clear(); chi2_.clear(); depth_.clear(); }   // a new window; the device buffers stay
    // Rwb / twb: pKF->GetImuRotation / GetImuPosition; Rcw / tcw of camera 0: pKF->GetRotation / GetTranslation (row-major doubles widened from
    // the float cv::Mat); camera 1 (if any) is derived like G2oTypes.cc:55-63 by the caller and passed in Rcw1 / tcw1 (may be null)
    int addKeyFrame(const double Rwb[9], const double twb[3], const double Rcw0[9], const double tcw0[3], const double* Rcw1, const double* tcw1,
                    const double v[3], const double bg[3], const double ba[3], bool poseFixed, bool hasImu, bool imuFixed) {
        liba_keyframe k{};
        for (int i = 0; i < 9; i++) { k.Rwb[i] = Rwb[i]; k.Rcw[0][i] = Rcw0[i]; if (Rcw1) k.Rcw[1][i] = Rcw1[i]; }
        for (int i = 0; i < 3; i++) { k.twb[i] = twb[i]; k.tcw[0][i] = tcw0[i]; if (tcw1) k.tcw[1][i] = tcw1[i]; k.v[i] = v ? v[i] : 0; k.bg[i] = bg ? bg[i] : 0; k.ba[i] = ba ? ba[i] : 0; }
        k.pose_fixed = poseFixed; k.has_imu = hasImu; k.imu_fixed = imuFixed;
        kfs_.push_back(k);
        return (int)kfs_.size() - 1;
    }
    void addInertial(const liba_imu_edge& e) { imu_.push_back(e); }
    int addPoint(const float Xw[3]) { pts_.push_back(Xw[0]); pts_.push_back(Xw[1]); pts_.push_back(Xw[2]); return (int)pts_.size() / 3 - 1; }
    void addMono(int kf, int point, float u, float v, float invSigma2, int camIdx = 0) { edges_.push_back(lba_edge{kf, point, LBA_EDGE_MONO, (int16_t)camIdx, {u, v, 0.f}, invSigma2}); }
    void addStereo(int kf, int point, float u, float v, float uR, float invSigma2) { edges_.push_back(lba_edge{kf, point, LBA_EDGE_STEREO, 0, {u, v, uR}, invSigma2}); }

    // returns the iterations run (-1: the window was rejected: unsorted edges, > LIBA_MAX_FREE optimisable key frames, bad indices)
    int optimize(double lambdaInit, int iterations, double* err = nullptr, double* errEnd = nullptr) {
        const int nk = (int)kfs_.size(), nl = (int)pts_.size() / 3, ne = (int)edges_.size(), ni = (int)imu_.size();
        if (!nk || !nl || !ne) return 0;
        int nfree = 0;
        for (auto& k : kfs_) nfree += !k.pose_fixed;
        if (nfree == 0 || nfree > LIBA_MAX_FREE) return -1;
        liba_problem P{};
        P.kfs = dk_.upload(kfs_.data(), nk); P.points = dp_.upload(pts_.data(), pts_.size()); P.edges = de_.upload(edges_.data(), ne);
        P.imu = di_.upload(imu_.data(), ni ? ni : 0); P.rigs = dr_.upload(&rig_, 1);
        const int32_t n4[4] = {nk, nl, ne, ni};
        const int32_t* dn = dn_.upload(n4, 4);
        P.n_kf = dn; P.n_points = dn + 1; P.n_edges = dn + 2; P.n_imu = dn + 3;
        P.cap_kf = nk; P.cap_l = nl; P.cap_e = ne; P.cap_i = ni > 0 ? ni : 1; P.rig_stride = 0; P.max_free = nfree;
        P.huber_mono = (double)std::sqrt(5.991f); P.huber_stereo = (double)std::sqrt(7.815f);   // const float thHuberMono = sqrt(5.991) (Optimizer.cc:5071)
        void* work = dw_.ensure(liba_workspace_bytes(&P, 1));
        double* dst = (double*)ds_.ensure(5 * 8 + (size_t)ne * 9 + 64);
        uint8_t* ddp = (uint8_t*)(dst + 5 + ne);
        if (liba_optimize(&P, 1, lambdaInit, iterations, work, dst, nullptr) != ORB_OK) throw std::runtime_error("liba_optimize");
        if (liba_compute_errors(&P, 1, dst + 5, ddp, nullptr, nullptr, nullptr) != ORB_OK) throw std::runtime_error("liba_compute_errors");
        double st[5];
        chi2_.resize(ne); depth_.resize(ne);
        orb_memcpy_d2h(st, dst, sizeof(st), nullptr);
        orb_memcpy_d2h(chi2_.data(), dst + 5, (size_t)ne * 8, nullptr);
        orb_memcpy_d2h(depth_.data(), ddp, (size_t)ne, nullptr);
        orb_memcpy_d2h(kfs_.data(), P.kfs, (size_t)nk * sizeof(liba_keyframe), nullptr);
        orb_memcpy_d2h(pts_.data(), P.points, pts_.size() * 8, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        if (err) *err = st[4];
        if (errEnd) *errEnd = st[1];
        return (int)st[0];
    }
    const liba_keyframe& keyFrame(int i) const { return kfs_[i]; }
    const double* point(int i) const { return &pts_[(size_t)i * 3]; }
    double visualChi2(int edge) const { return chi2_[edge]; }          // e->chi2() after the optimisation (Optimizer.cc:5247,5264)
    bool depthPositive(int edge) const { return depth_[edge] != 0; }   // e->isDepthPositive()

private:
    liba_rig rig_{};
    std::vector<liba_keyframe> kfs_;
    std::vector<double> pts_;
    std::vector<lba_edge> edges_;
    std::vector<liba_imu_edge> imu_;
    std::vector<double> chi2_;
    std::vector<uint8_t> depth_;
    detail::DevBuf dk_, dp_, de_, di_, dr_, dn_, dw_, ds_;
};


// Optimizer::PoseInertialOptimizationLastKeyFrame(Frame*, bool bRecInit) (reference include/Optimizer.h:70, src/Optimizer.cc:7665-8067) over
// liba_pose_inertial_kf().  The caller walks pFrame->mvpMapPoints like :7706-7790 (addMono / addStereo, bClose = pMP->mTrackDepth < 10.f), passes the
// frame's and the last key frame's ImuCamPose / velocity / biases and the preintegration record (kf1 = key frame, kf2 = frame; `info` from the
// reference's EdgeInertial constructor), and reads back the state (-> SetImuPoseVelocity, mImuBias), mvbOutlier, H (-> new ConstraintPoseImu) and
// the return value.
class PoseInertialOptimizer {
public:
    void setRig(const liba_rig& rig) { rig_ = rig; }
    void clear() { edges_.clear(); }
    void addMono(const float Xw[3], float u, float v, float invSigma2, bool bClose, int camIdx = 0) {
        edges_.push_back(pose_edge{{Xw[0], Xw[1], Xw[2]}, {u, v, 0.f}, invSigma2, (int16_t)(LBA_EDGE_MONO | (bClose ? LIBA_EDGE_CLOSE : 0)), (int16_t)camIdx});
    }
    void addStereo(const float Xw[3], float u, float v, float uR, float invSigma2) {
        edges_.push_back(pose_edge{{Xw[0], Xw[1], Xw[2]}, {u, v, uR}, invSigma2, (int16_t)LBA_EDGE_STEREO, 0});
    }
    int size() const { return (int)edges_.size(); }
    // frame is updated in place; H15 = row-major 15x15 (pose 6, velocity 3, gyro bias 3, acc bias 3)
    int optimize(liba_keyframe& frame, const liba_keyframe& lastKeyFrame, const liba_imu_edge& preint, bool bRecInit, std::vector<bool>& mvbOutlier, double H15[225]) {
        const int ne = (int)edges_.size();
        mvbOutlier.assign(ne, false);
        if (ne == 0) return 0;
        liba_keyframe* dF = f_.upload(&frame, 1);
        const liba_keyframe* dK = k_.upload(&lastKeyFrame, 1);
        const liba_rig* dR = r_.upload(&rig_, 1);
        const pose_edge* dE = e_.upload(edges_.data(), ne);
        const liba_imu_edge* dI = i_.upload(&preint, 1);
        const int32_t* dN = n_.upload(&ne, 1);
        uint8_t* dO = (uint8_t*)o_.ensure((size_t)ne + 16);
        double* dH = (double*)h_.ensure(225 * 8);
        int32_t* dG = (int32_t*)g_.ensure(16);
        if (liba_pose_inertial_kf(dF, dK, dR, 0, dE, dN, ne, dI, 1, bRecInit ? 1 : 0, dO, dH, dG, nullptr) != ORB_OK) throw std::runtime_error("liba_pose_inertial_kf");
        std::vector<uint8_t> fl(ne);
        int32_t good = 0;
        orb_memcpy_d2h(&frame, dF, sizeof(frame), nullptr);
        orb_memcpy_d2h(fl.data(), dO, ne, nullptr);
        orb_memcpy_d2h(H15, dH, 225 * 8, nullptr);
        orb_memcpy_d2h(&good, dG, 4, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        for (int i = 0; i < ne; i++) mvbOutlier[i] = fl[i] != 0;
        return good;
    }

    // Optimizer::PoseInertialOptimizationLastFrame (include/Optimizer.h:71, src/Optimizer.cc:8068-8415): prevFrame (pFrame->mpPrevFrame's state) is free too
    // and updated in place; prior = pFp->mpcpi's members (after the ConstraintPoseImu constructor); preint = mpImuPreintegratedFrame (kf1 = previous frame);
    // H15 = H.block<15,15>(15,15) of the marginalised Hessian -> new ConstraintPoseImu(...).
    int optimizeLastFrame(liba_keyframe& frame, liba_keyframe& prevFrame, const liba_prior& prior, const liba_imu_edge& preint, bool bRecInit,
                          std::vector<bool>& mvbOutlier, double H15[225]) {
        const int ne = (int)edges_.size();
        mvbOutlier.assign(ne, false);
        if (ne == 0) return 0;
        liba_keyframe* dF = f_.upload(&frame, 1);
        liba_keyframe* dK = k_.upload(&prevFrame, 1);
        const liba_rig* dR = r_.upload(&rig_, 1);
        const pose_edge* dE = e_.upload(edges_.data(), ne);
        const liba_imu_edge* dI = i_.upload(&preint, 1);
        const liba_prior* dC = c_.upload(&prior, 1);
        const int32_t* dN = n_.upload(&ne, 1);
        uint8_t* dO = (uint8_t*)o_.ensure((size_t)ne + 16);
        double* dH = (double*)h_.ensure(225 * 8);
        int32_t* dG = (int32_t*)g_.ensure(16);
        if (liba_pose_inertial_lastframe(dF, dK, dR, 0, dE, dN, ne, dI, dC, 1, bRecInit ? 1 : 0, dO, dH, dG, nullptr) != ORB_OK) throw std::runtime_error("liba_pose_inertial_lastframe");
        std::vector<uint8_t> fl(ne);
        int32_t good = 0;
        orb_memcpy_d2h(&frame, dF, sizeof(frame), nullptr);
        orb_memcpy_d2h(&prevFrame, dK, sizeof(prevFrame), nullptr);
        orb_memcpy_d2h(fl.data(), dO, ne, nullptr);
        orb_memcpy_d2h(H15, dH, 225 * 8, nullptr);
        orb_memcpy_d2h(&good, dG, 4, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        for (int i = 0; i < ne; i++) mvbOutlier[i] = fl[i] != 0;
        return good;
    }

private:
    liba_rig rig_{};
    std::vector<pose_edge> edges_;
    detail::DevBuf f_, k_, r_, e_, i_, n_, o_, h_, g_, c_;
};

}  // namespace orbslam3_hip
#endif
