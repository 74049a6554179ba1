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
#include <climits>
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
enum { Q_VALID = 1, Q_STEREO = 2, Q_HAS_OBS = 4, Q_RIGHT = 8, Q_TWIN = 16 };
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
    int Nleft = -1;   // fisheye rig: keypoints [0, Nleft) = mvKeys (left), [Nleft, N) = mvKeysRight; -1 = single camera
    std::vector<int> mGrid[GRID_COLS][GRID_ROWS];
    std::vector<int> mGridRight[GRID_COLS][GRID_ROWS];   // holds right-local indices (i - Nleft), Frame.cc:470-476

    bool PosInGrid(const KeyPoint& kp, int& posX, int& posY) const {  // Frame.cc:852-862
        posX = (int)std::round((kp.x - mnMinX) * gwInv);
        posY = (int)std::round((kp.y - mnMinY) * ghInv);
        if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) return false;
        return true;
    }
    void AssignFeaturesToGrid() {  // Frame.cc:444-478
        for (int i = 0; i < GRID_COLS; i++)
            for (int j = 0; j < GRID_ROWS; j++) { mGrid[i][j].clear(); mGridRight[i][j].clear(); }
        for (int i = 0; i < N; i++) {
            int gx, gy;
            if (PosInGrid(kps[i], gx, gy)) {
                if (Nleft == -1 || i < Nleft) mGrid[gx][gy].push_back(i);
                else mGridRight[gx][gy].push_back(i - Nleft);
            }
        }
    }
    std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel,
                                          const int maxLevel, const bool bRight = false) const {  // Frame.cc:755-850
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
                const std::vector<int>& vCell = (!bRight) ? mGrid[ix][iy] : mGridRight[ix][iy];
                for (size_t j = 0; j < vCell.size(); j++) {
                    const KeyPoint& kpUn = (Nleft == -1) ? kps[vCell[j]] : (!bRight) ? kps[vCell[j]] : kps[vCell[j] + Nleft];
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
                for (size_t j = 0; j < rotHist[i].size(); j++) { holder[rotHist[i][j]] = -2; nmatches--; }   // NULL written by the orientation cull (reported as -2: "claimed during the call, then culled")
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
                      const int32_t* f_feat, int f_nodes, float nnratio, int checkOri, int32_t* f_match, int f_nleft) {
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
                int bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
                for (int iF = f_node_start[Fit]; iF < f_node_start[Fit + 1]; iF++) {
                    const unsigned int realIdxF = f_feat[iF];
                    if (vpMapPointMatches[realIdxF] >= 0) continue;
                    const int dist = DescriptorDistance(dKF, f_desc + (size_t)realIdxF * 32);
                    if (f_nleft == -1) {
                        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                        else if (dist < bestDist2) { bestDist2 = dist; }
                    } else {   // :411-436
                        if ((int)realIdxF < f_nleft && dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                        else if ((int)realIdxF < f_nleft && dist < bestDist2) { bestDist2 = dist; }
                        if ((int)realIdxF >= f_nleft && dist < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = dist; bestIdxFR = realIdxF; }
                        else if ((int)realIdxF >= f_nleft && dist < bestDist2R) { bestDist2R = dist; }
                    }
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
                    if (bestDist1R <= TH_LOW) {   // :505-540
                        if (static_cast<float>(bestDist1R) < nnratio * static_cast<float>(bestDist2R) || true) {
                            vpMapPointMatches[bestIdxFR] = realIdxKF;
                            if (checkOri) {
                                float rot = kf_angle[realIdxKF] - f_angle[bestIdxFR];
                                if (rot < 0.0) rot += 360.0f;
                                int bin = (int)std::round(rot * factor);
                                if (bin == HISTO_LENGTH) bin = 0;
                                rotHist[bin].push_back(bestIdxFR);
                            }
                            nmatches++;
                        }
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

// SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12)  ORBmatcher.cc:984-1124 (call site LoopClosing.cc:697).
// valid1[i] / valid2[i]: the key frame's feature i holds a map point that is not bad (`!pMP1 / pMP1->isBad()` :1025-1028, `!pMP2 /
// pMP2->isBad()` :1049-1053) AND, for a fisheye-rig key frame, i < mvKeysUn.size() (:1020-1022, :1043-1045) — the caller folds both
// into the flag.  match12[i1] = index of the KF2 feature whose map point vpMatches12[i1] holds, or -1.  Differences from the
// (KeyFrame, Frame) overload that matter bit for bit: the acceptance is `bestDist1 < TH_LOW` (strict, :1072), the blocking state is
// vbMatched2 (never reset by the orientation cull), and the rotation histogram holds idx1 (:1088).
int omo_search_by_bow_kf(const uint8_t* desc1, const float* angle1, const uint8_t* valid1, const int32_t* node_id1,
                         const int32_t* node_start1, const int32_t* feat1, int nodes1, int n1, const uint8_t* desc2,
                         const float* angle2, const uint8_t* valid2, const int32_t* node_id2, const int32_t* node_start2,
                         const int32_t* feat2, int nodes2, int n2, float nnratio, int checkOri, int32_t* match12) {
    std::vector<int> vpMatches12(n1, -1);
    std::vector<bool> vbMatched2(n2, false);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    int f1it = 0, f2it = 0;
    auto lower_bound = [](const int32_t* ids, int n, int key) { int lo = 0, hi = n; while (lo < hi) { int m = (lo + hi) / 2; if (ids[m] < key) lo = m + 1; else hi = m; } return lo; };
    while (f1it != nodes1 && f2it != nodes2) {
        if (node_id1[f1it] == node_id2[f2it]) {
            for (int i1 = node_start1[f1it]; i1 < node_start1[f1it + 1]; i1++) {
                const size_t idx1 = feat1[i1];
                if (!valid1[idx1]) continue;
                const uint8_t* d1 = desc1 + idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int i2 = node_start2[f2it]; i2 < node_start2[f2it + 1]; i2++) {
                    const size_t idx2 = feat2[i2];
                    if (vbMatched2[idx2] || !valid2[idx2]) continue;
                    const int dist = DescriptorDistance(d1, desc2 + idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)idx2; }
                    else if (dist < bestDist2) { bestDist2 = dist; }
                }
                if (bestDist1 < TH_LOW) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        vpMatches12[idx1] = bestIdx2;
                        vbMatched2[bestIdx2] = true;
                        if (checkOri) {
                            float rot = angle1[idx1] - angle2[bestIdx2];
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back((int)idx1);
                        }
                        nmatches++;
                    }
                }
            }
            f1it++;
            f2it++;
        } else if (node_id1[f1it] < node_id2[f2it]) {
            f1it = lower_bound(node_id1, nodes1, node_id2[f2it]);
        } else {
            f2it = lower_bound(node_id2, nodes2, node_id1[f1it]);
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { vpMatches12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < n1; i++) match12[i] = vpMatches12[i];
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

// ---- fisheye rig (F.Nleft != -1) twins of the two projection searches -----------------------------------------------------------
// Grid of a rig frame as CSR with 2 x 64 x 48 cells: cells [0, G) = mGrid, [G, 2G) = mGridRight; entries are GLOBAL keypoint indices.
void omo_grid_build_rig(const void* kps, int n, int nleft, float minX, float minY, float gwInv, float ghInv, int32_t* grid_start, int32_t* grid_idx) {
    FrameView F{n, (const KeyPoint*)kps, nullptr, nullptr, minX, minY, gwInv, ghInv};
    F.Nleft = nleft;
    F.AssignFeaturesToGrid();
    int pos = 0;
    for (int side = 0; side < 2; side++)
        for (int ix = 0; ix < GRID_COLS; ix++)
            for (int iy = 0; iy < GRID_ROWS; iy++) {
                grid_start[side * GRID_COLS * GRID_ROWS + ix * GRID_ROWS + iy] = pos;
                const std::vector<int>& c = side ? F.mGridRight[ix][iy] : F.mGrid[ix][iy];
                for (int v : c) grid_idx[pos++] = side ? v + nleft : v;
            }
    grid_start[2 * GRID_COLS * GRID_ROWS] = pos;
}

// SearchByProjection on a rig frame.  Queries come in map-point order; a map point seen by both cameras contributes its left query
// followed by its right query (Q_RIGHT | Q_TWIN).  link[i] = global index of the stereo partner of keypoint i
// (mvLeftToRightMatch[i] + Nleft for i < Nleft, mvRightToLeftMatch[i - Nleft] otherwise) or -1.
//   mode 0: ORBmatcher.cc:59-258 — the `continue` of the left ratio test (:166-167) also skips the right camera of that map point;
//           an accepted match is copied to the stereo partner (:172-176, :239-243) and counted twice.
//   mode 1: ORBmatcher.cc:2244-2509 — `if(vIndices2.empty()) continue;` (:2332) skips the right camera when the left window is empty.
int omo_search_by_projection_rig(const void* kps, const uint8_t* desc, const uint8_t* occupied0, int n, int nleft, const int32_t* link,
                                 float minX, float minY, float gwInv, float ghInv, const void* queries_, const uint8_t* qdesc, int nq, int mode,
                                 int th_dist, float nnratio, int checkOri, int32_t* q_match, int32_t* kp_match) {
    FrameView F{n, (const KeyPoint*)kps, desc, nullptr, minX, minY, gwInv, ghInv};
    F.Nleft = nleft;
    F.AssignFeaturesToGrid();
    const Query* Q = (const Query*)queries_;
    std::vector<int> holder(n, -1);
    std::vector<char> holderObs(n, 0);
    for (int i = 0; i < n; i++) { kp_match[i] = -1; if (occupied0 && occupied0[i]) holderObs[i] = 1; }
    for (int q = 0; q < nq; q++) q_match[q] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    bool skipTwin = false;
    for (int iq = 0; iq < nq; iq++) {
        const Query& pMP = Q[iq];
        const bool twin = (pMP.flags & Q_TWIN) != 0, bRight = (pMP.flags & Q_RIGHT) != 0;
        if (!twin) skipTwin = false;
        if (!(pMP.flags & Q_VALID)) {
            if (mode == 1 && !twin) skipTwin = true;   // pMP == NULL / outlier / behind the camera / outside the image: `continue` (:2262-2290)
            continue;
        }
        if (twin && skipTwin) continue;
        const std::vector<size_t> vIndices = F.GetFeaturesInArea(pMP.u, pMP.v, pMP.radius, pMP.min_level, pMP.max_level, bRight);
        if (vIndices.empty()) {
            if (mode == 1 && !bRight) skipTwin = true;   // :2332
            continue;
        }
        const uint8_t* MPdescriptor = qdesc + (size_t)iq * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = bRight ? vIndices[k] + nleft : vIndices[k];
            if (holderObs[idx]) continue;
            const int dist = DescriptorDistance(MPdescriptor, F.desc + idx * 32);
            if (mode == 0) {
                if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F.kps[idx].octave; bestIdx = (int)idx; }
                else if (dist < bestDist2) { bestLevel2 = F.kps[idx].octave; bestDist2 = dist; }
            } else {
                if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
            }
        }
        if (bestDist <= th_dist) {
            const char obs = (pMP.flags & Q_HAS_OBS) ? 1 : 0;
            if (mode == 0) {
                if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) { if (!bRight) skipTwin = true; continue; }
                holder[bestIdx] = iq; holderObs[bestIdx] = obs;
                q_match[iq] = bestIdx;
                nmatches++;
                if (link && link[bestIdx] != -1) { holder[link[bestIdx]] = iq; holderObs[link[bestIdx]] = obs; nmatches++; }
            } else {
                holder[bestIdx] = iq; holderObs[bestIdx] = obs;
                q_match[iq] = bestIdx;
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
                for (size_t j = 0; j < rotHist[i].size(); j++) { holder[rotHist[i][j]] = -2; nmatches--; }   // NULL written by the orientation cull (reported as -2: "claimed during the call, then culled")
    }
    for (int i = 0; i < n; i++) kp_match[i] = holder[i];
    for (int q = 0; q < nq; q++)
        if (q_match[q] >= 0 && holder[q_match[q]] != q) q_match[q] = -1;
    return nmatches;
}

// ---- SURVEY N1 / rows M11, M12 -----------------------------------------------------------------------------------------------
// M11 ORBmatcher::SearchForInitialization (ORBmatcher.cc:838-979).  prev: vbPrevMatched [n1][2] (updated in place, :972-975).
// NOTE this fork computes the histogram bin with factor = HISTO_LENGTH/360.0f here (:852), unlike the other searches.
int omo_search_for_initialization(const void* kps1_, const uint8_t* desc1, int n1, const void* kps2_, const uint8_t* desc2, int n2,
                                  float minX, float minY, float gwInv, float ghInv, float* prev, int windowSize, float nnratio,
                                  int checkOri, int32_t* matches12) {
    const KeyPoint* kps1 = (const KeyPoint*)kps1_;
    FrameView F2{n2, (const KeyPoint*)kps2_, desc2, nullptr, minX, minY, gwInv, ghInv};
    F2.AssignFeaturesToGrid();
    int nmatches = 0;
    std::vector<int> vnMatches12(n1, -1);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = HISTO_LENGTH / 360.0f;
    std::vector<int> vMatchedDistance(n2, INT_MAX);
    std::vector<int> vnMatches21(n2, -1);
    for (size_t i1 = 0, iend1 = n1; i1 < iend1; i1++) {
        KeyPoint kp1 = kps1[i1];
        int level1 = kp1.octave;
        if (level1 > 0) continue;
        std::vector<size_t> vIndices2 = F2.GetFeaturesInArea(prev[2 * i1], prev[2 * i1 + 1], windowSize, level1, level1);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = desc1 + i1 * 32;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (std::vector<size_t>::iterator vit = vIndices2.begin(); vit != vIndices2.end(); vit++) {
            size_t i2 = *vit;
            int dist = DescriptorDistance(d1, desc2 + i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
            else if (dist < bestDist2) { bestDist2 = dist; }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = (int)i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOri) {
                    float rot = kps1[i1].angle - F2.kps[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back((int)i1);
                }
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                int idx1 = rotHist[i][j];
                if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (size_t i1 = 0; i1 < (size_t)n1; i1++) {
        matches12[i1] = vnMatches12[i1];
        if (vnMatches12[i1] >= 0) { prev[2 * i1] = F2.kps[vnMatches12[i1]].x; prev[2 * i1 + 1] = F2.kps[vnMatches12[i1]].y; }
    }
    return nmatches;
}

// M12 ORBmatcher::Fuse — the search half of both overloads (the pointer-graph mutations Replace/AddObservation stay on the host):
//   chi2_gate=1: Fuse(KeyFrame*, const vector<MapPoint*>&, th, bRight)  ORBmatcher.cc:1630-1882 (:1770-1830 is restated here)
//   chi2_gate=0: Fuse(KeyFrame*, cv::Mat Scw, vpPoints, th, vpReplacePoint)  ORBmatcher.cc:1884-2006 (:1960-1985)
// A query carries what the reference computes per map point before GetFeaturesInArea: uv, ur, radius, nPredictedLevel (=max_level;
// min_level = nPredictedLevel-1).  KeyFrame::GetFeaturesInArea (KeyFrame.cc:810-854) has no level filter.
// q_match[q] = bestIdx if bestDist <= th_dist else -1; q_dist[q] = bestDist (256 when the window was empty).  Returns nFused.
int omo_fuse(const void* kps, const uint8_t* desc, const float* uRight, int n, float minX, float minY, float gwInv, float ghInv,
             const void* queries_, const uint8_t* qdesc, int nq, int th_dist, int chi2_gate, const float* invLevelSigma2, int32_t* q_match,
             int32_t* q_dist) {
    FrameView F{n, (const KeyPoint*)kps, desc, uRight, minX, minY, gwInv, ghInv};
    F.AssignFeaturesToGrid();
    const Query* Q = (const Query*)queries_;
    int nFused = 0;
    for (int i = 0; i < nq; i++) {
        q_match[i] = -1; q_dist[i] = 256;
        const Query& q = Q[i];
        if (!(q.flags & Q_VALID)) continue;
        const int nPredictedLevel = q.max_level;
        const float ur = q.u_right;
        const std::vector<size_t> vIndices = F.GetFeaturesInArea(q.u, q.v, q.radius, -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = qdesc + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1;
        for (std::vector<size_t>::const_iterator vit = vIndices.begin(), vend = vIndices.end(); vit != vend; vit++) {
            size_t idx = *vit;
            const KeyPoint& kp = F.kps[idx];
            const int& kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (chi2_gate) {
                if (F.uRight && F.uRight[idx] >= 0) {
                    const float ex = q.u - kp.x, ey = q.v - kp.y, er = ur - F.uRight[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * invLevelSigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = q.u - kp.x, ey = q.v - kp.y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * invLevelSigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = DescriptorDistance(dMP, F.desc + idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        q_dist[i] = bestDist;
        if (bestDist <= th_dist) { q_match[i] = bestIdx; nFused++; }
    }
    return nFused;
}

// ORBmatcher::SearchBySim3 (ORBmatcher.cc:2008-2220).  q12[i1] / q21[i2]: the per-map-point values computed before KeyFrame::GetFeaturesInArea
// (:2044-2080, :2122-2158; VALID iff the point exists, is not already matched and passed the gates; max_level = nPredictedLevel).
int omo_search_by_sim3(const void* kps1, const uint8_t* desc1, int n1, float minX1, float minY1, float gwInv1, float ghInv1,
                       const void* kps2, const uint8_t* desc2, int n2, float minX2, float minY2, float gwInv2, float ghInv2,
                       const void* q12_, const uint8_t* q12desc, const void* q21_, const uint8_t* q21desc, int32_t* matches12) {
    FrameView K1{n1, (const KeyPoint*)kps1, desc1, nullptr, minX1, minY1, gwInv1, ghInv1};
    FrameView K2{n2, (const KeyPoint*)kps2, desc2, nullptr, minX2, minY2, gwInv2, ghInv2};
    K1.AssignFeaturesToGrid(); K2.AssignFeaturesToGrid();
    const Query* Q12 = (const Query*)q12_;
    const Query* Q21 = (const Query*)q21_;
    std::vector<int> vnMatch1(n1, -1), vnMatch2(n2, -1);
    // Transform from KF1 to KF2 and search (:2044-2120)
    for (int i1 = 0; i1 < n1; i1++) {
        const Query& q = Q12[i1];
        if (!(q.flags & Q_VALID)) continue;
        const int nPredictedLevel = q.max_level;
        const std::vector<size_t> vIndices = K2.GetFeaturesInArea(q.u, q.v, q.radius, -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = q12desc + (size_t)i1 * 32;
        int bestDist = INT_MAX, bestIdx = -1;
        for (std::vector<size_t>::const_iterator vit = vIndices.begin(), vend = vIndices.end(); vit != vend; vit++) {
            const size_t idx = *vit;
            const KeyPoint& kp = K2.kps[idx];
            if (kp.octave < nPredictedLevel - 1 || kp.octave > nPredictedLevel) continue;
            const int dist = DescriptorDistance(dMP, K2.desc + idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        if (bestDist <= TH_HIGH) vnMatch1[i1] = bestIdx;
    }
    // Transform from KF2 to KF1 and search (:2122-2201)
    for (int i2 = 0; i2 < n2; i2++) {
        const Query& q = Q21[i2];
        if (!(q.flags & Q_VALID)) continue;
        const int nPredictedLevel = q.max_level;
        const std::vector<size_t> vIndices = K1.GetFeaturesInArea(q.u, q.v, q.radius, -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = q21desc + (size_t)i2 * 32;
        int bestDist = INT_MAX, bestIdx = -1;
        for (std::vector<size_t>::const_iterator vit = vIndices.begin(), vend = vIndices.end(); vit != vend; vit++) {
            const size_t idx = *vit;
            const KeyPoint& kp = K1.kps[idx];
            if (kp.octave < nPredictedLevel - 1 || kp.octave > nPredictedLevel) continue;
            const int dist = DescriptorDistance(dMP, K1.desc + idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        if (bestDist <= TH_HIGH) vnMatch2[i2] = bestIdx;
    }
    // Check agreement (:2203-2219)
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        matches12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            const int idx1 = vnMatch2[idx2];
            if (idx1 == i1) { matches12[i1] = idx2; nFound++; }
        }
    }
    return nFound;
}

// M12 ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse)  ORBmatcher.cc:1138-1428, for pinhole
// key frames without a second camera (mpCamera2 == NULL).  The epipolar gate is Pinhole::epipolarConstrain (Pinhole.cpp:155-177) with
// its fundamental matrix F12 = K1^-T [t12]x R12 K2^-1 passed in (it is a constant of the key-frame pair; the reference recomputes it
// per candidate with cv::Mat arithmetic).  has_mp*: pKF->GetMapPoint(idx) != NULL.  Note vbMatched2 is never set by the reference.
struct TriSide {
    const KeyPoint* kps; const uint8_t* desc; const float* uRight; const uint8_t* has_mp;
    const int32_t* node_id; const int32_t* node_start; const int32_t* feat; int n_nodes, N;
};
int omo_search_for_triangulation(const void* s1_, const void* s2_, const float* F12, const float* ep, const float* levelSigma2_2,
                                 const float* scaleFactors_2, int bOnlyStereo, int bCoarse, int checkOri, int32_t* matches12) {
    const TriSide& K1 = *(const TriSide*)s1_;
    const TriSide& K2 = *(const TriSide*)s2_;
    int nmatches = 0;
    std::vector<bool> vbMatched2(K2.N, false);
    std::vector<int> vMatches12(K1.N, -1);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    auto lower_bound = [](const int32_t* ids, int n, int key) { int lo = 0, hi = n; while (lo < hi) { int m = (lo + hi) / 2; if (ids[m] < key) lo = m + 1; else hi = m; } return lo; };
    auto epipolarConstrain = [&](const KeyPoint& kp1, const KeyPoint& kp2, const float unc) {   // Pinhole.cpp:162-176
        const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
        const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
        const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
        const float num = a * kp2.x + b * kp2.y + c;
        const float den = a * a + b * b;
        if (den == 0) return false;
        const float dsqr = num * num / den;
        return dsqr < 3.84 * unc;
    };
    int f1 = 0, f2 = 0;
    while (f1 != K1.n_nodes && f2 != K2.n_nodes) {
        if (K1.node_id[f1] == K2.node_id[f2]) {
            for (int i1 = K1.node_start[f1]; i1 < K1.node_start[f1 + 1]; i1++) {
                const size_t idx1 = K1.feat[i1];
                if (K1.has_mp[idx1]) continue;
                const bool bStereo1 = K1.uRight && K1.uRight[idx1] >= 0;
                if (bOnlyStereo) if (!bStereo1) continue;
                const KeyPoint& kp1 = K1.kps[idx1];
                const uint8_t* d1 = K1.desc + idx1 * 32;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int i2 = K2.node_start[f2]; i2 < K2.node_start[f2 + 1]; i2++) {
                    size_t idx2 = K2.feat[i2];
                    if (vbMatched2[idx2] || K2.has_mp[idx2]) continue;
                    const bool bStereo2 = K2.uRight && K2.uRight[idx2] >= 0;
                    if (bOnlyStereo) if (!bStereo2) continue;
                    const int dist = DescriptorDistance(d1, K2.desc + idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const KeyPoint& kp2 = K2.kps[idx2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ep[0] - kp2.x, distey = ep[1] - kp2.y;
                        if (distex * distex + distey * distey < 100 * scaleFactors_2[kp2.octave]) continue;
                    }
                    if (epipolarConstrain(kp1, kp2, levelSigma2_2[kp2.octave]) || bCoarse) { bestIdx2 = (int)idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    const KeyPoint& kp2 = K2.kps[bestIdx2];
                    vMatches12[idx1] = bestIdx2;
                    nmatches++;
                    if (checkOri) {
                        float rot = kp1.angle - kp2.angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back((int)idx1);
                    }
                }
            }
            f1++; f2++;
        } else if (K1.node_id[f1] < K2.node_id[f2]) {
            f1 = lower_bound(K1.node_id, K1.n_nodes, K2.node_id[f2]);
        } else {
            f2 = lower_bound(K2.node_id, K2.n_nodes, K1.node_id[f1]);
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { vMatches12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < K1.N; i++) matches12[i] = vMatches12[i];
    return nmatches;
}

// M12 on KannalaBrandt8 key frames (monocular fisheye: n_cams = 1; fisheye rig with mpCamera2: n_cams = 2).  ORBmatcher.cc:1138-1428 with the
// rig branches: kp = mvKeys / mvKeysRight (concatenated in K.kps, :1249-1251 / :1261-1263), bStereo1 = bStereo2 = false (mvuRight is unset),
// the epipole test only without a second camera (:1269), the (R12, t12, pCamera1, pCamera2) combination per candidate (:1280-1315) and
// pCamera1->epipolarConstrain(...) = KannalaBrandt8::TriangulateMatches(...) > 0.0001f (KannalaBrandt8.cpp:235-238; frame_oracle.cpp, rule R4).
struct TriKb8Pair {
    int32_t n_cams, reserved;
    float k1[2][8], k2[2][8];
    float R12[4][9], t12[4][3];
    float ep[2];
    float level_sigma2_1[16], level_sigma2_2[16], scale_factors_2[16];
};
float ofr_triangulate_matches(const float*, const float*, const void*, const void*, const float*, const float*, float, float, float*);
int omo_search_for_triangulation_kb8(const void* s1_, const void* s2_, int nleft1, int nleft2, const void* pair_, int bOnlyStereo, int bCoarse,
                                     int checkOri, int32_t* matches12) {
    const TriSide& K1 = *(const TriSide*)s1_;
    const TriSide& K2 = *(const TriSide*)s2_;
    const TriKb8Pair& Q = *(const TriKb8Pair*)pair_;
    const bool hasCam2 = Q.n_cams == 2;
    int nmatches = 0;
    std::vector<bool> vbMatched2(K2.N, false);
    std::vector<int> vMatches12(K1.N, -1);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    auto lower_bound = [](const int32_t* ids, int n, int key) { int lo = 0, hi = n; while (lo < hi) { int m = (lo + hi) / 2; if (ids[m] < key) lo = m + 1; else hi = m; } return lo; };
    const float* R12 = Q.R12[0]; const float* t12 = Q.t12[0];
    const float* pCamera1 = Q.k1[0]; const float* pCamera2 = Q.k2[0];
    int f1 = 0, f2 = 0;
    while (f1 != K1.n_nodes && f2 != K2.n_nodes) {
        if (K1.node_id[f1] == K2.node_id[f2]) {
            for (int i1 = K1.node_start[f1]; i1 < K1.node_start[f1 + 1]; i1++) {
                const size_t idx1 = K1.feat[i1];
                if (K1.has_mp[idx1]) continue;
                const bool bStereo1 = false;   // (!pKF1->mpCamera2 && mvuRight[idx1] >= 0): mvuRight is -1 for fisheye frames
                if (bOnlyStereo) if (!bStereo1) continue;
                const KeyPoint& kp1 = K1.kps[idx1];
                const bool bRight1 = (!hasCam2 || (int)idx1 < nleft1) ? false : true;
                const uint8_t* d1 = K1.desc + idx1 * 32;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int i2 = K2.node_start[f2]; i2 < K2.node_start[f2 + 1]; i2++) {
                    size_t idx2 = K2.feat[i2];
                    if (vbMatched2[idx2] || K2.has_mp[idx2]) continue;
                    const bool bStereo2 = false;
                    if (bOnlyStereo) if (!bStereo2) continue;
                    const int dist = DescriptorDistance(d1, K2.desc + idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const KeyPoint& kp2 = K2.kps[idx2];
                    const bool bRight2 = (!hasCam2 || (int)idx2 < nleft2) ? false : true;
                    if (!bStereo1 && !bStereo2 && !hasCam2) {
                        const float distex = Q.ep[0] - kp2.x, distey = Q.ep[1] - kp2.y;
                        if (distex * distex + distey * distey < 100 * Q.scale_factors_2[kp2.octave]) continue;
                    }
                    if (hasCam2) {
                        const int c = (bRight1 ? 2 : 0) + (bRight2 ? 1 : 0);
                        R12 = Q.R12[c]; t12 = Q.t12[c];
                        pCamera1 = Q.k1[bRight1 ? 1 : 0]; pCamera2 = Q.k2[bRight2 ? 1 : 0];
                    }
                    float p3D[3];
                    if (ofr_triangulate_matches(pCamera1, pCamera2, &kp1, &kp2, R12, t12, Q.level_sigma2_1[kp1.octave], Q.level_sigma2_2[kp2.octave], p3D) > 0.0001f ||
                        bCoarse) {
                        bestIdx2 = (int)idx2; bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {
                    const KeyPoint& kp2 = K2.kps[bestIdx2];
                    vMatches12[idx1] = bestIdx2;
                    nmatches++;
                    if (checkOri) {
                        float rot = kp1.angle - kp2.angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back((int)idx1);
                    }
                }
            }
            f1++; f2++;
        } else if (K1.node_id[f1] < K2.node_id[f2]) {
            f1 = lower_bound(K1.node_id, K1.n_nodes, K2.node_id[f2]);
        } else {
            f2 = lower_bound(K2.node_id, K2.n_nodes, K1.node_id[f1]);
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { vMatches12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < K1.N; i++) matches12[i] = vMatches12[i];
    return nmatches;
}

}  // extern "C"
