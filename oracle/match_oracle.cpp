// ORACLE — TEST INFRASTRUCTURE ONLY (see orb_oracle.cpp header).  Stage 2: ORBmatcher + Frame grid helpers.
//
// CPU restatement of the reference's matcher loops over the SAME flattened records the product C ABI takes
// (include/orbhip.h, stage 2).  Everything here lives in the reference's own source (no third-party arithmetic
// except cv::BFMatcher::knnMatch, restated from its documented semantics): the loops below follow
//   ORBmatcher::DescriptorDistance                    src/ORBmatcher.cc:2700-2716
//   Frame::AssignFeaturesToGrid / PosInGrid           src/Frame.cc:444-478, 852-862
//   Frame::GetFeaturesInArea                          src/Frame.cc:755-850
//   ORBmatcher::SearchByProjection(F, vpMapPoints)    src/ORBmatcher.cc:59-255   (left-camera branch, Nleft == -1)
//   ORBmatcher::SearchByProjection(Cur, Last)         src/ORBmatcher.cc:2244-2509 (Nleft == -1)
//   ORBmatcher::ComputeThreeMaxima                    src/ORBmatcher.cc:2654-2695
//   ORBmatcher::SearchByBoW(KF, F)                    src/ORBmatcher.cc:323-587  (Nleft == -1)
//   Frame::ComputeStereoFishEyeMatches' knnMatch      src/Frame.cc:1300 (cv::BFMatcher NORM_HAMMING, k=2)
// line by line, with the pointer-graph reads replaced by the flattened fields.  Parity for this stage is pinned
// by the reference source itself (integer Hamming + float compares), except knnMatch's tie rule [recalled].
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct KeyPoint { float x, y, size, angle, response; int octave, class_id; };

struct Query {  // == orbm_query
    float u, v, radius, u_right, angle;
    int16_t min_level, max_level;
    uint32_t flags;
};
enum { Q_VALID = 1, Q_STEREO = 2, Q_HAS_OBS = 4 };
const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;
const int GRID_COLS = 64, GRID_ROWS = 48;

int DescriptorDistance(const uint8_t* a, const uint8_t* b) {  // ORBmatcher.cc:2700-2716 (bit-hack popcount)
    const int32_t* pa = (const int32_t*)a;
    const int32_t* pb = (const int32_t*)b;
    int dist = 0;
    for (int i = 0; i < 8; i++, pa++, pb++) {
        unsigned int v = *pa ^ *pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

struct FrameView {
    int N;
    const KeyPoint* kps;
    const uint8_t* desc;
    const float* uRight;  // may be null
    float mnMinX, mnMinY, gwInv, ghInv;
    std::vector<int> mGrid[GRID_COLS][GRID_ROWS];

    bool PosInGrid(const KeyPoint& kp, int& posX, int& posY) const {  // Frame.cc:852-862
        posX = (int)std::round((kp.x - mnMinX) * gwInv);
        posY = (int)std::round((kp.y - mnMinY) * ghInv);
        if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) return false;
        return true;
    }
    void AssignFeaturesToGrid() {  // Frame.cc:444-478
        for (int i = 0; i < GRID_COLS; i++)
            for (int j = 0; j < GRID_ROWS; j++) mGrid[i][j].clear();
        for (int i = 0; i < N; i++) {
            int gx, gy;
            if (PosInGrid(kps[i], gx, gy)) mGrid[gx][gy].push_back(i);
        }
    }
    std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel,
                                          const int maxLevel) const {  // Frame.cc:755-850
        std::vector<size_t> vIndices;
        float factorX = r, factorY = r;
        const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - factorX) * gwInv));
        if (nMinCellX >= GRID_COLS) return vIndices;
        const int nMaxCellX = std::min((int)GRID_COLS - 1, (int)std::ceil((x - mnMinX + factorX) * gwInv));
        if (nMaxCellX < 0) return vIndices;
        const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - factorY) * ghInv));
        if (nMinCellY >= GRID_ROWS) return vIndices;
        const int nMaxCellY = std::min((int)GRID_ROWS - 1, (int)std::ceil((y - mnMinY + factorY) * ghInv));
        if (nMaxCellY < 0) return vIndices;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<int>& vCell = mGrid[ix][iy];
                for (size_t j = 0; j < vCell.size(); j++) {
                    const KeyPoint& kpUn = kps[vCell[j]];
                    if (bCheckLevels) {
                        if (kpUn.octave < minLevel) continue;
                        if (maxLevel >= 0)
                            if (kpUn.octave > maxLevel) continue;
                    }
                    const float distx = kpUn.x - x, disty = kpUn.y - y;
                    if (std::fabs(distx) < factorX && std::fabs(disty) < factorY) vIndices.push_back(vCell[j]);
                }
            }
        return vIndices;
    }
};

void ComputeThreeMaxima(const std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {  // :2654-2695
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

}  // namespace

extern "C" {

int omo_hamming(const uint8_t* a, const uint8_t* b) { return DescriptorDistance(a, b); }

// grid as CSR (cell = ix*48+iy), the layout orbm_grid_build produces
void omo_grid_build(const void* kps, int n, float minX, float minY, float gwInv, float ghInv, int32_t* grid_start,
                    int32_t* grid_idx) {
    FrameView F{n, (const KeyPoint*)kps, nullptr, nullptr, minX, minY, gwInv, ghInv};
    F.AssignFeaturesToGrid();
    int pos = 0;
    for (int ix = 0; ix < GRID_COLS; ix++)
        for (int iy = 0; iy < GRID_ROWS; iy++) {
            grid_start[ix * GRID_ROWS + iy] = pos;
            for (int v : F.mGrid[ix][iy]) grid_idx[pos++] = v;
        }
    grid_start[GRID_COLS * GRID_ROWS] = pos;
}

// Windowed projection search.  mode 0 = local map (best/second + level rule + ratio), 1 = best only (+ rot. histogram).
// mvpMapPoints is modelled as kp_match[idx] = query index (or -1); occupied0[idx] != 0 models a pre-existing observed point.
int omo_search_by_projection(const void* kps, const uint8_t* desc, const float* uRight, const uint8_t* occupied0, int n,
                             float minX, float minY, float gwInv, float ghInv, const void* queries_, const uint8_t* qdesc,
                             int nq, int mode, int th_dist, float nnratio, int checkOri, int32_t* q_match, int32_t* kp_match) {
    FrameView F{n, (const KeyPoint*)kps, desc, uRight, minX, minY, gwInv, ghInv};
    F.AssignFeaturesToGrid();
    const Query* Q = (const Query*)queries_;
    std::vector<int> holder(n, -1);          // which query's MP sits in mvpMapPoints[idx]
    std::vector<char> holderObs(n, 0);       // ... and whether that MP has Observations()>0
    for (int i = 0; i < n; i++) { kp_match[i] = -1; if (occupied0 && occupied0[i]) holderObs[i] = 1; }
    for (int q = 0; q < nq; q++) q_match[q] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int iMP = 0; iMP < nq; iMP++) {
        const Query& pMP = Q[iMP];
        if (!(pMP.flags & Q_VALID)) continue;
        const std::vector<size_t> vIndices = F.GetFeaturesInArea(pMP.u, pMP.v, pMP.radius, pMP.min_level, pMP.max_level);
        if (vIndices.empty()) continue;
        const uint8_t* MPdescriptor = qdesc + (size_t)iMP * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            if (holderObs[idx]) continue;  // F.mvpMapPoints[idx] && ->Observations()>0   (:125-127, :2350-2352)
            if ((pMP.flags & Q_STEREO) && F.uRight && F.uRight[idx] > 0) {
                const float er = std::fabs(pMP.u_right - F.uRight[idx]);
                if (er > pMP.radius) continue;
            }
            const int dist = DescriptorDistance(MPdescriptor, F.desc + idx * 32);
            if (mode == 0) {
                if (dist < bestDist) {
                    bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel;
                    bestLevel = F.kps[idx].octave; bestIdx = (int)idx;
                } else if (dist < bestDist2) {
                    bestLevel2 = F.kps[idx].octave; bestDist2 = dist;
                }
            } else {
                if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
            }
        }
        if (bestDist <= th_dist) {
            if (mode == 0) {
                if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
                if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                    holder[bestIdx] = iMP; holderObs[bestIdx] = (pMP.flags & Q_HAS_OBS) ? 1 : 0;
                    q_match[iMP] = bestIdx;
                    nmatches++;
                }
            } else {
                holder[bestIdx] = iMP; holderObs[bestIdx] = (pMP.flags & Q_HAS_OBS) ? 1 : 0;
                q_match[iMP] = bestIdx;
                nmatches++;
                if (checkOri) {
                    float rot = pMP.angle - F.kps[bestIdx].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(bestIdx);
                }
            }
        }
    }
    if (mode == 1 && checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) { holder[rotHist[i][j]] = -1; nmatches--; }
    }
    for (int i = 0; i < n; i++) kp_match[i] = holder[i];
    // a query's match is reported only while its keypoint still holds it
    for (int q = 0; q < nq; q++)
        if (q_match[q] >= 0 && holder[q_match[q]] != q) q_match[q] = -1;
    return nmatches;
}

// SearchByBoW(KF, F): FeatureVectors as CSR (node ids ascending).  f_match[j] = KF feature index or -1.
int omo_search_by_bow(const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_valid, const int32_t* kf_node_id,
                      const int32_t* kf_node_start, const int32_t* kf_feat, int kf_nodes, const uint8_t* f_desc,
                      const float* f_angle, int fN, const int32_t* f_node_id, const int32_t* f_node_start,
                      const int32_t* f_feat, int f_nodes, float nnratio, int checkOri, int32_t* f_match) {
    std::vector<int> vpMapPointMatches(fN, -1);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int KFit = 0, Fit = 0;
    auto lower_bound = [](const int32_t* ids, int n, int key) { int lo = 0, hi = n; while (lo < hi) { int m = (lo + hi) / 2; if (ids[m] < key) lo = m + 1; else hi = m; } return lo; };
    while (KFit != kf_nodes && Fit != f_nodes) {
        if (kf_node_id[KFit] == f_node_id[Fit]) {
            for (int iKF = kf_node_start[KFit]; iKF < kf_node_start[KFit + 1]; iKF++) {
                const unsigned int realIdxKF = kf_feat[iKF];
                if (!kf_valid[realIdxKF]) continue;  // !pMP || pMP->isBad()
                const uint8_t* dKF = kf_desc + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int iF = f_node_start[Fit]; iF < f_node_start[Fit + 1]; iF++) {
                    const unsigned int realIdxF = f_feat[iF];
                    if (vpMapPointMatches[realIdxF] >= 0) continue;
                    const int dist = DescriptorDistance(dKF, f_desc + (size_t)realIdxF * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) { bestDist2 = dist; }
                }
                if (bestDist1 <= TH_LOW) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        vpMapPointMatches[bestIdxF] = realIdxKF;
                        if (checkOri) {
                            float rot = kf_angle[realIdxKF] - f_angle[bestIdxF];
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(bestIdxF);
                        }
                        nmatches++;
                    }
                }
            }
            KFit++;
            Fit++;
        } else if (kf_node_id[KFit] < f_node_id[Fit]) {
            KFit = lower_bound(kf_node_id, kf_nodes, f_node_id[Fit]);
        } else {
            Fit = lower_bound(f_node_id, f_nodes, kf_node_id[KFit]);
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { vpMapPointMatches[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    for (int j = 0; j < fN; j++) f_match[j] = vpMapPointMatches[j];
    return nmatches;
}

// cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, k=2): ascending train scan, strict '<' insertion [recalled, Appendix B5]
void omo_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* out_idx, int32_t* out_dist) {
    for (int i = 0; i < nq; i++) {
        int d0 = 256, d1 = 256, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            const int d = DescriptorDistance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
            else if (d < d1) { d1 = d; i1 = j; }
        }
        out_idx[2 * i] = i0; out_idx[2 * i + 1] = i1;
        out_dist[2 * i] = d0; out_dist[2 * i + 1] = d1;
    }
}

}  // extern "C"
