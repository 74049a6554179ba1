// orbslam3_hip/ORBmatcher.h — adapter for ORB_SLAM3::ORBmatcher (reference include/ORBmatcher.h:39-94) over
// liborbhip.so (include/orbhip.h, stage 2).
//
// The reference methods take Frame& / KeyFrame* / MapPoint*; `FrameView` and `ProjectedPoint` are the flattened records an
// integration gathers from those objects (the gather loops are shown in INTEGRATION.md, one per call site) and
// `SearchByProjection*` scatter the result back in the reference's serial order (mvpMapPoints[idx] = pMP).
#ifndef ORBSLAM3_HIP_ORBMATCHER_H
#define ORBSLAM3_HIP_ORBMATCHER_H
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <utility>
#include <vector>

#include "../orbhip.h"

namespace orbslam3_hip {

namespace detail {
struct DevBuf {
    void* p = nullptr; size_t cap = 0; int device = 0;
    ~DevBuf() { if (p) orb_dev_free(p); }
    void* ensure(size_t n) {
        if (n > cap) { if (p) orb_dev_free(p); p = nullptr; if (orb_dev_alloc(device, n, &p) != ORB_OK) throw std::runtime_error("orb_dev_alloc"); cap = n; }
        return p;
    }
    template <class T> T* upload(const T* h, size_t count) {
        T* d = (T*)ensure(count * sizeof(T) + 16);
        if (count && orb_memcpy_h2d(d, h, count * sizeof(T), nullptr) != ORB_OK) throw std::runtime_error("orb_memcpy_h2d");
        return d;
    }
};
// page-locked host block that only ever grows (orb_host_alloc): the packed staging blocks of the per-frame searches
struct HostBuf {
    uint8_t* p = nullptr; size_t cap = 0;
    HostBuf() = default;
    HostBuf(const HostBuf&) = delete;
    HostBuf& operator=(const HostBuf&) = delete;
    ~HostBuf() { if (p) orb_host_free(p); }
    uint8_t* ensure(size_t n) {
        if (n > cap) {
            if (p) orb_host_free(p);
            p = nullptr; void* q = nullptr;
            const size_t want = n + n / 2;
            if (orb_host_alloc(want, &q) != ORB_OK) throw std::runtime_error("orb_host_alloc");
            p = (uint8_t*)q; cap = want;
        }
        return p;
    }
};
}  // namespace detail

// What the matcher reads from an ORB_SLAM3::Frame (Nleft == -1): N, mvKeysUn, mDescriptors, mvuRight, and the static
// bounds mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv (Frame.cc:388-399).
struct FrameView {
    int N = 0;
    const orb_keypoint* keysUn = nullptr;     // == cv::KeyPoint array (28-byte layout)
    const uint8_t* descriptors = nullptr;     // N x 32
    const float* uRight = nullptr;            // mvuRight or nullptr (monocular)
    const uint8_t* occupied = nullptr;        // 1 where mvpMapPoints[i] && ->Observations()>0 before the call (or nullptr)
    orbm_grid_params grid{0, 0, 0, 0};
    // fisheye rig (Frame::Nleft != -1): keysUn / descriptors = [mvKeys | mvKeysRight] (N entries), Nleft as in the reference, kpLink[i] =
    // mvLeftToRightMatch[i] + Nleft for i < Nleft, mvRightToLeftMatch[i - Nleft] otherwise (or -1); queries then come as left/right twins
    // (ORBM_Q_RIGHT | ORBM_Q_TWIN, see orbhip.h)
    int Nleft = -1;
    const int32_t* kpLink = nullptr;
};

class ORBmatcher {
public:
    static const int TH_LOW = ORBM_TH_LOW, TH_HIGH = ORBM_TH_HIGH, HISTO_LENGTH = ORBM_HISTO_LENGTH;  // ORBmatcher.h:92-94

    ORBmatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}  // ORBmatcher.h:39

    // ORBmatcher::DescriptorDistance (ORBmatcher.cc:2700-2716).  A single 256-bit pair is not device work: this scalar
    // form serves the callers that stay on the CPU (e.g. MapPoint::ComputeDistinctiveDescriptors); bulk distances go
    // through orbm_hamming / the search kernels.
    static int DescriptorDistance(const uint8_t* a, const uint8_t* b) {
        int dist = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t x, y;
            std::memcpy(&x, a + 4 * i, 4);
            std::memcpy(&y, b + 4 * i, 4);
            dist += __builtin_popcount(x ^ y);
        }
        return dist;
    }

    // Flattened SearchByProjection.  mode = ORBM_MODE_LOCAL_MAP  <=> SearchByProjection(Frame&, vector<MapPoint*>&, th, ...)  (ORBmatcher.cc:59)
    //                                mode = ORBM_MODE_BEST_ONLY  <=> SearchByProjection(Frame&, const Frame&, th, bMono)      (ORBmatcher.cc:2244)
    // queries[i]/qdesc[i] describe map point i (see orbm_query); kpMatch[idx] receives the index of the query whose map point
    // ends up in mvpMapPoints[idx] (-1: none).  Returns the reference's nmatches.
    int SearchByProjection(const FrameView& F, const std::vector<orbm_query>& queries, const std::vector<uint8_t>& qdesc, int mode,
                           int thDist, std::vector<int>& kpMatch, std::vector<int>& queryMatch) {
        const int n = F.N, nq = (int)queries.size();
        kpMatch.assign(n, -1);
        queryMatch.assign(nq, -1);
        if (n == 0 || nq == 0) return 0;
        // ONE packed host->device transfer per call: every input is laid out in a host staging block (256-byte aligned sections) that mirrors a
        // persistent device block; ONE device->host transfer brings back [q_match | kp_match | nmatches].  The buffers only ever grow.
        size_t off = 0;
        auto sec = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
        const size_t oK = sec((size_t)n * sizeof(orb_keypoint)), oD = sec((size_t)n * 32), oU = sec(F.uRight ? (size_t)n * 4 : 0),
                     oO = sec(F.occupied ? (size_t)n : 0), oQ = sec((size_t)nq * sizeof(orbm_query)), oQD = sec((size_t)nq * 32), oC = sec(16),
                     oL = sec((F.Nleft != -1 && F.kpLink) ? (size_t)n * 4 : 0);
        uint8_t* stage = stage_.ensure(off);
        std::memcpy(stage + oK, F.keysUn, (size_t)n * sizeof(orb_keypoint));
        std::memcpy(stage + oD, F.descriptors, (size_t)n * 32);
        if (F.uRight) std::memcpy(stage + oU, F.uRight, (size_t)n * 4);
        if (F.occupied) std::memcpy(stage + oO, F.occupied, (size_t)n);
        std::memcpy(stage + oQ, queries.data(), (size_t)nq * sizeof(orbm_query));
        std::memcpy(stage + oQD, qdesc.data(), (size_t)nq * 32);
        const int32_t counts[3] = {n, nq, F.Nleft};
        std::memcpy(stage + oC, counts, sizeof(counts));
        if (F.Nleft != -1 && F.kpLink) std::memcpy(stage + oL, F.kpLink, (size_t)n * 4);
        uint8_t* dIn = in_.upload(stage, off);
        const orb_keypoint* dk = (const orb_keypoint*)(dIn + oK);
        const uint8_t* dd = dIn + oD;
        const float* dur = F.uRight ? (const float*)(dIn + oU) : nullptr;
        const uint8_t* docc = F.occupied ? dIn + oO : nullptr;
        const orbm_query* dq = (const orbm_query*)(dIn + oQ);
        const uint8_t* dqd = dIn + oQD;
        const int32_t* dc = (const int32_t*)(dIn + oC);
        int32_t* gs = (int32_t*)gs_.ensure((2 * ORBM_GRID_COLS * ORBM_GRID_ROWS + 1) * 4);
        int32_t* gi = (int32_t*)gi_.ensure((size_t)n * 4);
        const size_t oQM = 0, oKM = ((size_t)nq * 4 + 255) & ~(size_t)255, oNM = oKM + (((size_t)n * 4 + 255) & ~(size_t)255);
        uint8_t* dOut = (uint8_t*)out_.ensure(oNM + 256);
        int32_t* dqm = (int32_t*)(dOut + oQM); int32_t* dkm = (int32_t*)(dOut + oKM); int32_t* dnm = (int32_t*)(dOut + oNM);
        void* work = work_.ensure(orbm_search_workspace_bytes(1, nq));
        orbm_search_params prm{mode, thDist, mfNNratio, mbCheckOrientation ? 1 : 0, F.grid};
        if (F.Nleft == -1) {
            if (orbm_grid_build(dk, dc, 1, n, 1, &F.grid, gs, gi, nullptr) != ORB_OK) throw std::runtime_error("orbm_grid_build");
            if (orbm_search_by_projection(dk, dd, dur, docc, dc, 1, n, gs, gi, dq, dqd, dc + 1, nq, 1, &prm, dqm, dkm, dnm, work, nullptr) != ORB_OK)
                throw std::runtime_error("orbm_search_by_projection");
        } else {
            const int32_t* dlk = F.kpLink ? (const int32_t*)(dIn + oL) : nullptr;
            if (orbm_grid_build_rig(dk, dc, dc + 2, 1, n, 1, &F.grid, gs, gi, nullptr) != ORB_OK) throw std::runtime_error("orbm_grid_build_rig");
            if (orbm_search_by_projection_rig(dk, dd, docc, dlk, dc, 1, n, gs, gi, dq, dqd, dc + 1, nq, 1, &prm, dqm, dkm, dnm, work, nullptr) != ORB_OK)
                throw std::runtime_error("orbm_search_by_projection_rig");
        }
        const uint8_t* back = back_.ensure(oNM + 4);
        orb_memcpy_d2h(back_.p, dOut, oNM + 4, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        std::memcpy(queryMatch.data(), back + oQM, (size_t)nq * 4);
        std::memcpy(kpMatch.data(), back + oKM, (size_t)n * 4);
        int nmatches = 0;
        std::memcpy(&nmatches, back + oNM, 4);
        return nmatches;
    }

    // ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.h:71, ORBmatcher.cc:838-979).
    // vbPrevMatched: x,y pairs per F1 keypoint, updated in place for the matched ones (:972-975).
    int SearchForInitialization(const FrameView& F1, const FrameView& F2, std::vector<float>& vbPrevMatched, std::vector<int>& vnMatches12,
                                int windowSize = 10) {
        std::vector<orbm_query> q(F1.N);
        for (int i = 0; i < F1.N; i++) {
            q[i] = orbm_query{vbPrevMatched[2 * i], vbPrevMatched[2 * i + 1], (float)windowSize, 0.f, F1.keysUn[i].angle, 0, 0,
                              F1.keysUn[i].octave == 0 ? ORBM_Q_VALID : 0u};
        }
        std::vector<uint8_t> qd(F1.descriptors, F1.descriptors + (size_t)F1.N * 32);
        std::vector<int> vnMatches21;
        const int n = SearchByProjection(F2, q, qd, ORBM_MODE_INIT, TH_LOW, vnMatches21, vnMatches12);
        for (int i = 0; i < F1.N; i++)
            if (vnMatches12[i] >= 0) { vbPrevMatched[2 * i] = F2.keysUn[vnMatches12[i]].x; vbPrevMatched[2 * i + 1] = F2.keysUn[vnMatches12[i]].y; }
        return n;
    }

    // The search half of ORBmatcher::Fuse (ORBmatcher.h:85-88).  The caller computes per map point what the reference computes before
    // KeyFrame::GetFeaturesInArea (uv, ur, radius = th*mvScaleFactors[nPredictedLevel], levels [nPredictedLevel-1, nPredictedLevel];
    // flags = ORBM_Q_VALID iff the point passed the gates of ORBmatcher.cc:1700-1765 / :1910-1955) and afterwards applies
    // Replace / AddObservation / vpReplacePoint in index order on bestIdx[i] >= 0 (ORBmatcher.cc:1832-1855 / :1987-2000).
    // invLevelSigma2 != nullptr selects the KeyFrame overload's chi2 gate (:1791-1815); nullptr = the Sim3 overload.
    int Fuse(const FrameView& KF, const std::vector<orbm_query>& queries, const std::vector<uint8_t>& qdesc, const float* invLevelSigma2,
             int nLevels, std::vector<int>& bestIdx, std::vector<int>& bestDist) {
        const int n = KF.N, nq = (int)queries.size();
        bestIdx.assign(nq, -1);
        bestDist.assign(nq, 256);
        if (n == 0 || nq == 0) return 0;
        const orb_keypoint* dk = kps_.upload(KF.keysUn, n);
        const uint8_t* dd = desc_.upload(KF.descriptors, (size_t)n * 32);
        const float* dur = KF.uRight ? ur_.upload(KF.uRight, n) : nullptr;
        const orbm_query* dq = q_.upload(queries.data(), nq);
        const uint8_t* dqd = qd_.upload(qdesc.data(), (size_t)nq * 32);
        int32_t counts[2] = {n, nq};
        const int32_t* dc = cnt_.upload(counts, 2);
        int32_t* gs = (int32_t*)gs_.ensure((ORBM_GRID_COLS * ORBM_GRID_ROWS + 1) * 4);
        int32_t* gi = (int32_t*)gi_.ensure((size_t)n * 4);
        int32_t* dqm = (int32_t*)qm_.ensure((size_t)nq * 4);
        int32_t* dqdist = (int32_t*)km_.ensure((size_t)nq * 4);
        int32_t* dnm = (int32_t*)nm_.ensure(4);
        if (orbm_grid_build(dk, dc, 1, n, 1, &KF.grid, gs, gi, nullptr) != ORB_OK) throw std::runtime_error("orbm_grid_build");
        orbm_fuse_params prm{};
        prm.th_dist = TH_LOW; prm.chi2_gate = invLevelSigma2 ? 1 : 0; prm.grid = KF.grid;
        for (int i = 0; i < 16 && i < nLevels && invLevelSigma2; i++) prm.inv_level_sigma2[i] = invLevelSigma2[i];
        if (orbm_fuse(dk, dd, dur, dc, 1, n, gs, gi, dq, dqd, dc + 1, nq, 1, &prm, dqm, dqdist, dnm, nullptr) != ORB_OK) throw std::runtime_error("orbm_fuse");
        int nFused = 0;
        orb_memcpy_d2h(bestIdx.data(), dqm, (size_t)nq * 4, nullptr);
        orb_memcpy_d2h(bestDist.data(), dqdist, (size_t)nq * 4, nullptr);
        orb_memcpy_d2h(&nFused, dnm, 4, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        return nFused;
    }

    // ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (ORBmatcher.h:79, ORBmatcher.cc:2008-2220).  q12[i1] = the values the
    // reference computes for key frame 1's map point i1 before pKF2->GetFeaturesInArea (:2044-2080: VALID iff it exists, is not in vpMatches12
    // already, is not bad and passed the depth / image / distance gates; u, v, radius = th*mvScaleFactors[nPredictedLevel], max_level =
    // nPredictedLevel), q21[i2] likewise for key frame 2 (:2122-2158); the descriptors are pMP->GetDescriptor().  matches12[i1] = index into key
    // frame 2 of the agreed match or -1 (the caller sets vpMatches12[i1] = vpMapPoints2[idx2]); returns nFound.
    int SearchBySim3(const FrameView& KF1, const FrameView& KF2, const std::vector<orbm_query>& q12, const std::vector<uint8_t>& q12desc,
                     const std::vector<orbm_query>& q21, const std::vector<uint8_t>& q21desc, std::vector<int>& matches12) {
        const int n1 = KF1.N, n2 = KF2.N;
        matches12.assign(n1, -1);
        if ((int)q12.size() != n1 || (int)q21.size() != n2 || q12desc.size() != (size_t)n1 * 32 || q21desc.size() != (size_t)n2 * 32)
            throw std::invalid_argument("SearchBySim3: one query (and one 32-byte descriptor) per keypoint of each key frame is required");
        if (n1 == 0 || n2 == 0) return 0;
        const int32_t counts[2] = {n1, n2};
        const int32_t* dc = cnt_.upload(counts, 2);
        int32_t* vn1 = (int32_t*)qm_.ensure((size_t)n1 * 4);
        int32_t* vn2 = (int32_t*)km_.ensure((size_t)n2 * 4);
        int32_t* dist = (int32_t*)work_.ensure((size_t)std::max(n1, n2) * 4);
        int32_t* dnm = (int32_t*)nm_.ensure(8);
        orbm_fuse_params prm{};
        prm.th_dist = TH_HIGH; prm.chi2_gate = 0;
        for (int dir = 0; dir < 2; dir++) {      // dir 0: key frame 1's points searched in key frame 2
            const FrameView& T = dir == 0 ? KF2 : KF1;
            const std::vector<orbm_query>& q = dir == 0 ? q12 : q21;
            const std::vector<uint8_t>& qd = dir == 0 ? q12desc : q21desc;
            const orb_keypoint* dk = kps_.upload(T.keysUn, T.N);
            const uint8_t* dd = desc_.upload(T.descriptors, (size_t)T.N * 32);
            const orbm_query* dq = q_.upload(q.data(), q.size());
            const uint8_t* dqd = qd_.upload(qd.data(), qd.size());
            int32_t* gs = (int32_t*)gs_.ensure((ORBM_GRID_COLS * ORBM_GRID_ROWS + 1) * 4);
            int32_t* gi = (int32_t*)gi_.ensure((size_t)T.N * 4);
            prm.grid = T.grid;
            if (orbm_grid_build(dk, dc + (dir == 0 ? 1 : 0), 1, T.N, 1, &T.grid, gs, gi, nullptr) != ORB_OK) throw std::runtime_error("orbm_grid_build");
            if (orbm_fuse(dk, dd, nullptr, dc + (dir == 0 ? 1 : 0), 1, T.N, gs, gi, dq, dqd, dc + (dir == 0 ? 0 : 1), (int)q.size(), 1, &prm,
                          dir == 0 ? vn1 : vn2, dist, dnm, nullptr) != ORB_OK) throw std::runtime_error("orbm_fuse");
            if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");   // the upload buffers are reused by the second direction
        }
        int32_t* out = (int32_t*)ur_.ensure((size_t)n1 * 4);
        if (orbm_mutual_matches(vn1, vn2, dc, dc + 1, n1, n2, 1, out, dnm + 1, nullptr) != ORB_OK) throw std::runtime_error("orbm_mutual_matches");
        int nFound = 0;
        orb_memcpy_d2h(matches12.data(), out, (size_t)n1 * 4, nullptr);
        orb_memcpy_d2h(&nFound, dnm + 1, 4, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        return nFound;
    }

    // One key frame as SearchForTriangulation reads it: mvKeysUn, mDescriptors, mvuRight, GetMapPoint(i) != NULL, and mFeatVec as CSR
    // (node ids ascending = std::map order; featIdx = the concatenated per-node index vectors).
    struct KeyFrameView {
        int N = 0;
        const orb_keypoint* keysUn = nullptr;
        const uint8_t* descriptors = nullptr;
        const float* uRight = nullptr;
        const uint8_t* hasMapPoint = nullptr;
        std::vector<int32_t> nodeId, nodeStart, featIdx;
    };
    // ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse) (ORBmatcher.h:74, ORBmatcher.cc:1138-1428)
    // for pinhole key frames without mpCamera2.  F12: row-major K1^-T [t12]x R12 K2^-1 (the expression of Pinhole.cpp:157-160, evaluated
    // by the caller with the same cv::Mat arithmetic); ep: pKF2->mpCamera->project(R2w*Cw+t2w) (:1149-1152).
    int SearchForTriangulation(const KeyFrameView& K1, const KeyFrameView& K2, const float F12[9], const float ep[2], const float* levelSigma2_2,
                               const float* scaleFactors_2, int nLevels, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, bool bOnlyStereo,
                               bool bCoarse = false) {
        vMatchedPairs.clear();
        if (K1.N == 0 || K2.N == 0 || K1.nodeId.empty() || K2.nodeId.empty()) return 0;
        orbm_tri_side s[2];
        const KeyFrameView* K[2] = {&K1, &K2};
        int32_t nn[2] = {(int32_t)K1.nodeId.size(), (int32_t)K2.nodeId.size()};
        const int32_t* dnn = cnt_.upload(nn, 2);
        for (int i = 0; i < 2; i++) {
            s[i].kps = tk_[i].upload(K[i]->keysUn, K[i]->N);
            s[i].desc = td_[i].upload(K[i]->descriptors, (size_t)K[i]->N * 32);
            s[i].u_right = K[i]->uRight ? tu_[i].upload(K[i]->uRight, K[i]->N) : nullptr;
            s[i].has_mp = tm_[i].upload(K[i]->hasMapPoint, K[i]->N);
            s[i].node_id = tn_[i].upload(K[i]->nodeId.data(), K[i]->nodeId.size());
            s[i].node_start = ts_[i].upload(K[i]->nodeStart.data(), K[i]->nodeStart.size());
            s[i].feat_idx = tf_[i].upload(K[i]->featIdx.data(), K[i]->featIdx.size());
            s[i].n_nodes = dnn + i;
            s[i].cap_f = K[i]->N; s[i].cap_nodes = nn[i];
        }
        orbm_tri_pair P{};
        for (int i = 0; i < 9; i++) P.F12[i] = F12[i];
        P.ep[0] = ep[0]; P.ep[1] = ep[1];
        for (int i = 0; i < 16 && i < nLevels; i++) { P.level_sigma2_2[i] = levelSigma2_2[i]; P.scale_factors_2[i] = scaleFactors_2[i]; }
        const orbm_tri_pair* dP = q_.upload(&P, 1);
        int32_t* dm = (int32_t*)qm_.ensure((size_t)K1.N * 4);
        int32_t* dnm = (int32_t*)nm_.ensure(4);
        if (orbm_search_for_triangulation(&s[0], &s[1], dP, 1, bOnlyStereo ? 1 : 0, bCoarse ? 1 : 0, mbCheckOrientation ? 1 : 0, dm, dnm, nullptr) != ORB_OK)
            throw std::runtime_error("orbm_search_for_triangulation");
        std::vector<int32_t> m12(K1.N);
        int nmatches = 0;
        orb_memcpy_d2h(m12.data(), dm, (size_t)K1.N * 4, nullptr);
        orb_memcpy_d2h(&nmatches, dnm, 4, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        for (int i = 0; i < K1.N; i++)
            if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));   // :1415-1422
        return nmatches;
    }

    // The same call for key frames with KannalaBrandt8 cameras — a fisheye rig (pKF->mpCamera2 != NULL: K.keysUn = [mvKeys | mvKeysRight],
    // nLeft = pKF->NLeft) or one fisheye camera (nLeft = -1).  `pair` carries what the reference computes before its loops (ORBmatcher.cc:1144-1193):
    // the four (R12, t12) combinations, both cameras' parameters, the epipole and the level tables; see orbm_tri_kb8_pair in orbhip.h.
    int SearchForTriangulationKB8(const KeyFrameView& K1, int nLeft1, const KeyFrameView& K2, int nLeft2, const orbm_tri_kb8_pair& pair,
                                  std::vector<std::pair<size_t, size_t>>& vMatchedPairs, bool bOnlyStereo, bool bCoarse = false) {
        vMatchedPairs.clear();
        if (K1.N == 0 || K2.N == 0 || K1.nodeId.empty() || K2.nodeId.empty()) return 0;
        orbm_tri_side s[2];
        const KeyFrameView* K[2] = {&K1, &K2};
        int32_t nn[4] = {(int32_t)K1.nodeId.size(), (int32_t)K2.nodeId.size(), nLeft1, nLeft2};
        const int32_t* dnn = cnt_.upload(nn, 4);
        for (int i = 0; i < 2; i++) {
            s[i].kps = tk_[i].upload(K[i]->keysUn, K[i]->N);
            s[i].desc = td_[i].upload(K[i]->descriptors, (size_t)K[i]->N * 32);
            s[i].u_right = nullptr;
            s[i].has_mp = tm_[i].upload(K[i]->hasMapPoint, K[i]->N);
            s[i].node_id = tn_[i].upload(K[i]->nodeId.data(), K[i]->nodeId.size());
            s[i].node_start = ts_[i].upload(K[i]->nodeStart.data(), K[i]->nodeStart.size());
            s[i].feat_idx = tf_[i].upload(K[i]->featIdx.data(), K[i]->featIdx.size());
            s[i].n_nodes = dnn + i;
            s[i].cap_f = K[i]->N; s[i].cap_nodes = nn[i];
        }
        const orbm_tri_kb8_pair* dP = q_.upload(&pair, 1);
        int32_t* dm = (int32_t*)qm_.ensure((size_t)K1.N * 4);
        int32_t* dnm = (int32_t*)nm_.ensure(4);
        if (orbm_search_for_triangulation_kb8(&s[0], &s[1], dnn + 2, dnn + 3, dP, 1, bOnlyStereo ? 1 : 0, bCoarse ? 1 : 0, mbCheckOrientation ? 1 : 0, dm, dnm,
                                              nullptr) != ORB_OK)
            throw std::runtime_error("orbm_search_for_triangulation_kb8");
        std::vector<int32_t> m12(K1.N);
        int nmatches = 0;
        orb_memcpy_d2h(m12.data(), dm, (size_t)K1.N * 4, nullptr);
        orb_memcpy_d2h(&nmatches, dnm, 4, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        for (int i = 0; i < K1.N; i++)
            if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));   // :1415-1422
        return nmatches;
    }

    // ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (ORBmatcher.h:67, ORBmatcher.cc:323-587).
    // KF / F as KeyFrameView (descriptors + mFeatVec CSR; hasMapPoint on the KF side = "pKF map point exists and is not bad");
    // angleKF / angleF = keypoint angles (mvKeysUn / mvKeys / mvKeysRight as the reference picks them); nLeftF = F.Nleft (-1: one camera).
    // fMatch[j] = index of the KF feature whose map point lands in vpMapPointMatches[j], or -1.
    int SearchByBoW(const KeyFrameView& KF, const float* angleKF, const KeyFrameView& F, const float* angleF, int nLeftF, std::vector<int>& fMatch) {
        fMatch.assign(F.N, -1);
        if (KF.N == 0 || F.N == 0 || KF.nodeId.empty() || F.nodeId.empty()) return 0;
        orbm_bow_side s[2];
        const KeyFrameView* K[2] = {&KF, &F};
        const float* ang[2] = {angleKF, angleF};
        int32_t nn[3] = {(int32_t)KF.nodeId.size(), (int32_t)F.nodeId.size(), nLeftF};
        const int32_t* dnn = cnt_.upload(nn, 3);
        for (int i = 0; i < 2; i++) {
            s[i].desc = td_[i].upload(K[i]->descriptors, (size_t)K[i]->N * 32);
            s[i].angle = ta_[i].upload(ang[i], K[i]->N);
            s[i].node_id = tn_[i].upload(K[i]->nodeId.data(), K[i]->nodeId.size());
            s[i].node_start = ts_[i].upload(K[i]->nodeStart.data(), K[i]->nodeStart.size());
            s[i].feat_idx = tf_[i].upload(K[i]->featIdx.data(), K[i]->featIdx.size());
            s[i].n_nodes = dnn + i;
            s[i].cap_f = K[i]->N; s[i].cap_nodes = nn[i];
            s[i].n_left = nullptr;
        }
        if (nLeftF != -1) s[1].n_left = dnn + 2;
        const uint8_t* dv = tm_[0].upload(KF.hasMapPoint, KF.N);
        int32_t* dm = (int32_t*)qm_.ensure((size_t)F.N * 4);
        int32_t* dnm = (int32_t*)nm_.ensure(4);
        if (orbm_search_by_bow(&s[0], dv, &s[1], 1, mfNNratio, mbCheckOrientation ? 1 : 0, dm, dnm, nullptr) != ORB_OK) throw std::runtime_error("orbm_search_by_bow");
        int nmatches = 0;
        orb_memcpy_d2h(fMatch.data(), dm, (size_t)F.N * 4, nullptr);
        orb_memcpy_d2h(&nmatches, dnm, 4, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        return nmatches;
    }

    // ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (ORBmatcher.h:68, ORBmatcher.cc:984-1124; call site
    // LoopClosing.cc:697).  K1 / K2 as KeyFrameView; hasMapPoint[i] = "GetMapPointMatches()[i] exists and is not bad" AND, for a fisheye-rig
    // key frame (NLeft != -1), i < mvKeysUn.size() (:1020-1022, :1043-1045).  matches12[i1] = index of the pKF2 feature whose map point
    // vpMatches12[i1] receives, or -1.
    int SearchByBoW(const KeyFrameView& K1, const float* angle1, const KeyFrameView& K2, const float* angle2, std::vector<int>& matches12) {
        matches12.assign(K1.N, -1);
        if (K1.N == 0 || K2.N == 0 || K1.nodeId.empty() || K2.nodeId.empty()) return 0;
        orbm_bow_side s[2];
        const KeyFrameView* K[2] = {&K1, &K2};
        const float* ang[2] = {angle1, angle2};
        int32_t nn[2] = {(int32_t)K1.nodeId.size(), (int32_t)K2.nodeId.size()};
        const int32_t* dnn = cnt_.upload(nn, 2);
        const uint8_t* dv[2];
        for (int i = 0; i < 2; i++) {
            s[i].desc = td_[i].upload(K[i]->descriptors, (size_t)K[i]->N * 32);
            s[i].angle = ta_[i].upload(ang[i], K[i]->N);
            s[i].node_id = tn_[i].upload(K[i]->nodeId.data(), K[i]->nodeId.size());
            s[i].node_start = ts_[i].upload(K[i]->nodeStart.data(), K[i]->nodeStart.size());
            s[i].feat_idx = tf_[i].upload(K[i]->featIdx.data(), K[i]->featIdx.size());
            s[i].n_nodes = dnn + i;
            s[i].cap_f = K[i]->N; s[i].cap_nodes = nn[i];
            s[i].n_left = nullptr;
            dv[i] = tm_[i].upload(K[i]->hasMapPoint, K[i]->N);
        }
        int32_t* dm = (int32_t*)qm_.ensure((size_t)K1.N * 4);
        int32_t* dnm = (int32_t*)nm_.ensure(4);
        if (orbm_search_by_bow_kf(&s[0], dv[0], &s[1], dv[1], 1, mfNNratio, mbCheckOrientation ? 1 : 0, dm, dnm, nullptr) != ORB_OK)
            throw std::runtime_error("orbm_search_by_bow_kf");
        int nmatches = 0;
        orb_memcpy_d2h(matches12.data(), dm, (size_t)K1.N * 4, nullptr);
        orb_memcpy_d2h(&nmatches, dnm, 4, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        return nmatches;
    }

    float mfNNratio;
    bool mbCheckOrientation;

private:
    detail::DevBuf kps_, desc_, ur_, occ_, q_, qd_, cnt_, gs_, gi_, qm_, km_, nm_, work_, in_, out_;
    detail::HostBuf stage_, back_;   // page-locked host mirrors of the packed input / output blocks of SearchByProjection
    detail::DevBuf tk_[2], td_[2], tu_[2], tm_[2], tn_[2], ts_[2], tf_[2], nl_, lk_, ta_[2];
};

}  // namespace orbslam3_hip
#endif
