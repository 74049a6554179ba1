// orbslam3_hip/ORBmatcher.h — adapter for ORB_SLAM3::ORBmatcher (reference include/ORBmatcher.h:39-94) over
// liborbhip.so (include/orbhip.h, stage 2).
//
// The reference methods take Frame& / KeyFrame* / MapPoint*; `FrameView` and `ProjectedPoint` are the flattened records an
// integration gathers from those objects (the gather loops are shown in INTEGRATION.md, one per call site) and
// `SearchByProjection*` scatter the result back in the reference's serial order (mvpMapPoints[idx] = pMP).
#ifndef ORBSLAM3_HIP_ORBMATCHER_H
#define ORBSLAM3_HIP_ORBMATCHER_H
#include <cstdint>
#include <cstring>
#include <stdexcept>
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
        const orb_keypoint* dk = kps_.upload(F.keysUn, n);
        const uint8_t* dd = desc_.upload(F.descriptors, (size_t)n * 32);
        const float* dur = F.uRight ? ur_.upload(F.uRight, n) : nullptr;
        const uint8_t* docc = F.occupied ? occ_.upload(F.occupied, n) : nullptr;
        const orbm_query* dq = q_.upload(queries.data(), nq);
        const uint8_t* dqd = qd_.upload(qdesc.data(), (size_t)nq * 32);
        int32_t counts[2] = {n, nq};
        const int32_t* dc = cnt_.upload(counts, 2);
        int32_t* gs = (int32_t*)gs_.ensure((ORBM_GRID_COLS * ORBM_GRID_ROWS + 1) * 4);
        int32_t* gi = (int32_t*)gi_.ensure((size_t)n * 4);
        int32_t* dqm = (int32_t*)qm_.ensure((size_t)nq * 4);
        int32_t* dkm = (int32_t*)km_.ensure((size_t)n * 4);
        int32_t* dnm = (int32_t*)nm_.ensure(4);
        void* work = work_.ensure(orbm_search_workspace_bytes(1, nq));
        if (orbm_grid_build(dk, dc, 1, n, 1, &F.grid, gs, gi, nullptr) != ORB_OK) throw std::runtime_error("orbm_grid_build");
        orbm_search_params prm{mode, thDist, mfNNratio, mbCheckOrientation ? 1 : 0, F.grid};
        if (orbm_search_by_projection(dk, dd, dur, docc, dc, 1, n, gs, gi, dq, dqd, dc + 1, nq, 1, &prm, dqm, dkm, dnm, work, nullptr) != ORB_OK)
            throw std::runtime_error("orbm_search_by_projection");
        int nmatches = 0;
        orb_memcpy_d2h(kpMatch.data(), dkm, (size_t)n * 4, nullptr);
        orb_memcpy_d2h(queryMatch.data(), dqm, (size_t)nq * 4, nullptr);
        orb_memcpy_d2h(&nmatches, dnm, 4, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        return nmatches;
    }

    float mfNNratio;
    bool mbCheckOrientation;

private:
    detail::DevBuf kps_, desc_, ur_, occ_, q_, qd_, cnt_, gs_, gi_, qm_, km_, nm_, work_;
};

}  // namespace orbslam3_hip
#endif
