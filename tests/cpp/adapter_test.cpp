// Exercises the header-only C++ adapters (include/orbslam3_hip/*.h) against the oracle's C API.
// Built by tests/test_cpp_adapter.py against either the emulated library (CPU tier) or the real liborbhip.so (GPU tier).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "orbslam3_hip/ORBextractor.h"
#include "orbslam3_hip/ORBmatcher.h"
#include "orbslam3_hip/Optimizer.h"
#include "orbslam3_hip/ORBVocabulary.h"
#include "orbslam3_hip/Frame.h"

extern "C" {
void* oro_create(int, float, int, int, int);
void oro_destroy(void*);
int oro_extract(void*, const uint8_t*, int, int, int, int, int, void*, uint8_t*, int, int*);
int omo_search_by_projection(const void*, const uint8_t*, const float*, const uint8_t*, int, float, float, float, float, const void*,
                             const uint8_t*, int, int, int, float, int, int32_t*, int32_t*);
void olb_build_system(const double*, const int32_t*, int, const double*, int, const void*, int, const void*, double, double, double*,
                      double*, double*, double*, double*, double*, double*, double*, double*, double*);
void olb_optimize(double*, const int32_t*, int, double*, int, const void*, int, const void*, double, double, int, double*);
int opo_pose_optimize(const double*, const void*, int, const void*, double*, uint8_t*);
int omo_search_for_initialization(const void*, const uint8_t*, int, const void*, const uint8_t*, int, float, float, float, float, float*, int, float,
                                  int, int32_t*);
int omo_fuse(const void*, const uint8_t*, const float*, int, float, float, float, float, const void*, const uint8_t*, int, int, int, const float*,
             int32_t*, int32_t*);
struct OTriSide { const void* kps; const uint8_t* desc; const float* uRight; const uint8_t* has_mp; const int32_t* node_id; const int32_t* node_start;
                  const int32_t* feat; int n_nodes, N; };
void* obw_load_binary(const uint8_t*, size_t);
void obw_destroy(void*);
int obw_transform(void*, const uint8_t*, int, int, int32_t*, int32_t*, double*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, double*);
int omo_search_by_bow(const uint8_t*, const float*, const uint8_t*, const int32_t*, const int32_t*, const int32_t*, int, const uint8_t*, const float*, int,
                      const int32_t*, const int32_t*, const int32_t*, int, float, int, int32_t*, int);
int omo_search_by_bow_kf(const uint8_t*, const float*, const uint8_t*, const int32_t*, const int32_t*, const int32_t*, int, int, const uint8_t*, const float*,
                         const uint8_t*, const int32_t*, const int32_t*, const int32_t*, int, int, float, int, int32_t*);
void ofr_undistort_keypoints(const void*, int, const float*, void*);
void ofr_image_bounds(const float*, int, int, float*);
int ofr_stereo_fisheye(const void*, const uint8_t*, int, int, const void*, const uint8_t*, int, int, const float*, const float*, int32_t*, int32_t*, float*, float*);
void ofr_stereo_from_rgbd(const void*, const void*, int, const float*, int, float, float*, float*);
int oib_pose_inertial_lastframe(void*, void*, const void*, const void*, int, const void*, const void*, int, uint8_t*, double*);
int oib_pose_inertial_kf(void*, const void*, const void*, const void*, int, const void*, int, uint8_t*, double*);
void oib_optimize(void*, int, const void*, double*, int, const void*, int, const void*, int, double, double, double, int, double*);
int omo_search_by_sim3(const void*, const uint8_t*, int, float, float, float, float, const void*, const uint8_t*, int, float, float, float, float, const void*,
                       const uint8_t*, const void*, const uint8_t*, int32_t*);
int omo_search_for_triangulation(const void*, const void*, const float*, const float*, const float*, const float*, int, int, int, int32_t*);
}

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static std::vector<uint8_t> make_image(int W, int H, int dx, int dy) {
    std::vector<uint8_t> img((size_t)W * H, 110);
    rng_state = 777u;
    for (int r = 0; r < 220; r++) {
        const int cx = rnd() % W + dx, cy = rnd() % H + dy, hw = 3 + rnd() % 25, hh = 3 + rnd() % 25, g = 20 + rnd() % 215;
        for (int y = cy - hh; y <= cy + hh; y++)
            for (int x = cx - hw; x <= cx + hw; x++)
                if (x >= 0 && x < W && y >= 0 && y < H) img[(size_t)y * W + x] = (uint8_t)g;
    }
    for (size_t i = 0; i < img.size(); i++) img[i] = (uint8_t)std::min(255, std::max(0, (int)img[i] + (int)(rnd() % 5) - 2));
    return img;
}

int main() {
    const int W = 400, H = 300, NF = 400;
    // ---- stage 1 through the adapter
    orbslam3_hip::ORBextractor ex(NF, 1.2f, 8, 20, 7);
    CHECK(ex.GetLevels() == 8 && std::fabs(ex.GetScaleFactors()[3] - 1.728f) < 1e-5f);
    std::vector<uint8_t> imgA = make_image(W, H, 0, 0), imgB = make_image(W, H, 5, -3);
    std::vector<orb_keypoint> kA, kB;
    std::vector<uint8_t> dA, dB;
    std::vector<int> lap = {0, 1000};
    const int monoA = ex.extract(imgA.data(), W, H, W, kA, dA, lap);
    const int monoB = ex.extract(imgB.data(), W, H, W, kB, dB, lap);
    void* o = oro_create(NF, 1.2f, 8, 20, 7);
    std::vector<orb_keypoint> ok(NF + 64);
    std::vector<uint8_t> od((NF + 64) * 32);
    int n = 0;
    const int omono = oro_extract(o, imgB.data(), W, H, W, 0, 1000, ok.data(), od.data(), NF + 64, &n);
    CHECK(omono == monoB && n == (int)kB.size() && n > 150);
    CHECK(std::memcmp(ok.data(), kB.data(), (size_t)n * sizeof(orb_keypoint)) == 0);
    CHECK(std::memcmp(od.data(), dB.data(), (size_t)n * 32) == 0);
    CHECK(ex.extract(nullptr, 0, 0, 0, ok, od, lap) == -1);   // empty image -> -1 (ORBextractor.cc:1078-1079)
    int pw, ph;
    CHECK(ex.pyramidLevel(2, 19, pw, ph).size() == (size_t)(pw + 38) * (ph + 38));
    oro_destroy(o);
    (void)monoA;
    // ---- stage 2: motion-model search of A's keypoints in B
    orbslam3_hip::FrameView F;
    F.N = (int)kB.size(); F.keysUn = kB.data(); F.descriptors = dB.data();
    F.grid = orbm_grid_params{0.f, 0.f, 64.f / W, 48.f / H};
    std::vector<orbm_query> q(kA.size());
    const std::vector<float> sf = ex.GetScaleFactors();
    for (size_t i = 0; i < kA.size(); i++) {
        q[i].u = kA[i].x + 5.f; q[i].v = kA[i].y - 3.f; q[i].radius = 15.f * sf[kA[i].octave]; q[i].u_right = 0; q[i].angle = kA[i].angle;
        q[i].min_level = (int16_t)(kA[i].octave - 1); q[i].max_level = (int16_t)(kA[i].octave + 1); q[i].flags = ORBM_Q_VALID | ORBM_Q_HAS_OBS;
    }
    orbslam3_hip::ORBmatcher m(0.9f, true);
    std::vector<int> kpMatch, qMatch;
    const int nm = m.SearchByProjection(F, q, dA, ORBM_MODE_BEST_ONLY, ORBM_TH_HIGH, kpMatch, qMatch);
    std::vector<int32_t> oq(q.size()), okm(kB.size());
    const int onm = omo_search_by_projection(kB.data(), dB.data(), nullptr, nullptr, F.N, 0.f, 0.f, 64.f / W, 48.f / H, q.data(), dA.data(),
                                             (int)q.size(), 1, 100, 0.9f, 1, oq.data(), okm.data());
    CHECK(nm == onm && nm > 40);
    CHECK(std::memcmp(okm.data(), kpMatch.data(), okm.size() * 4) == 0 && std::memcmp(oq.data(), qMatch.data(), oq.size() * 4) == 0);
    CHECK(orbslam3_hip::ORBmatcher::DescriptorDistance(dA.data(), dA.data()) == 0);
    // ---- rows M11 / M12 through the adapters vs the oracle
    {
        orbslam3_hip::FrameView FA;
        FA.N = (int)kA.size(); FA.keysUn = kA.data(); FA.descriptors = dA.data(); FA.grid = F.grid;
        std::vector<float> prev(kA.size() * 2), oprev;
        for (size_t i = 0; i < kA.size(); i++) { prev[2 * i] = kA[i].x; prev[2 * i + 1] = kA[i].y; }
        oprev = prev;
        std::vector<int> m12;
        orbslam3_hip::ORBmatcher mi(0.9f, true);
        const int ni = mi.SearchForInitialization(FA, F, prev, m12, 100);
        std::vector<int32_t> om12(kA.size());
        const int oni = omo_search_for_initialization(kA.data(), dA.data(), FA.N, kB.data(), dB.data(), F.N, 0.f, 0.f, 64.f / W, 48.f / H, oprev.data(), 100,
                                                      0.9f, 1, om12.data());
        CHECK(ni == oni && ni > 20);
        CHECK(std::memcmp(om12.data(), m12.data(), om12.size() * 4) == 0 && std::memcmp(oprev.data(), prev.data(), prev.size() * 4) == 0);
        // Fuse (KeyFrame overload gate) with the motion-model queries re-levelled to [L-1, L]
        std::vector<orbm_query> fq = q;
        float invS2[8];
        for (int l = 0; l < 8; l++) invS2[l] = 1.0f / (sf[l] * sf[l]);
        for (size_t i = 0; i < fq.size(); i++) {
            fq[i].radius = 3.0f * sf[kA[i].octave]; fq[i].min_level = (int16_t)(kA[i].octave - 1); fq[i].max_level = (int16_t)kA[i].octave;
            fq[i].u_right = fq[i].u - 12.f; fq[i].flags = ORBM_Q_VALID;
        }
        std::vector<int> bi, bd;
        const int nf = m.Fuse(F, fq, dA, invS2, 8, bi, bd);
        std::vector<int32_t> obi(fq.size()), obd(fq.size());
        const int onf = omo_fuse(kB.data(), dB.data(), nullptr, F.N, 0.f, 0.f, 64.f / W, 48.f / H, fq.data(), dA.data(), (int)fq.size(), 50, 1, invS2,
                                 obi.data(), obd.data());
        CHECK(nf == onf && nf > 20);
        CHECK(std::memcmp(obi.data(), bi.data(), obi.size() * 4) == 0 && std::memcmp(obd.data(), bd.data(), obd.size() * 4) == 0);
        // SearchBySim3: frame A's points searched in B (the fuse queries, no gate) and B's in A, then the agreement pass
        {
            std::vector<orbm_query> q21(kB.size());
            for (size_t i = 0; i < kB.size(); i++) {
                q21[i] = orbm_query{};
                q21[i].u = kB[i].x - (fq[0].u - kA[0].x); q21[i].v = kB[i].y - (fq[0].v - kA[0].y);
                q21[i].radius = 7.5f * sf[kB[i].octave]; q21[i].min_level = (int16_t)(kB[i].octave - 1); q21[i].max_level = (int16_t)kB[i].octave;
                q21[i].flags = (i % 7) ? ORBM_Q_VALID : 0;
            }
            std::vector<orbm_query> q12 = fq;
            for (auto& e : q12) e.radius *= 2.5f;
            std::vector<int> s12;
            const int ns = m.SearchBySim3(FA, F, q12, dA, q21, dB, s12);
            std::vector<int32_t> os12(kA.size());
            const int ons = omo_search_by_sim3(kA.data(), dA.data(), FA.N, 0.f, 0.f, 64.f / W, 48.f / H, kB.data(), dB.data(), F.N, 0.f, 0.f, 64.f / W, 48.f / H,
                                               q12.data(), dA.data(), q21.data(), dB.data(), os12.data());
            CHECK(ns == ons && ns > 20);
            CHECK(std::memcmp(os12.data(), s12.data(), os12.size() * 4) == 0);
        }
        // SearchForTriangulation: vocabulary node = a hash of descriptor bits; pure image-plane translation geometry
        orbslam3_hip::ORBmatcher::KeyFrameView K1, K2;
        std::vector<uint8_t> mp1(kA.size(), 0), mp2(kB.size(), 0);
        auto build = [](orbslam3_hip::ORBmatcher::KeyFrameView& K, const std::vector<orb_keypoint>& kp, const std::vector<uint8_t>& d, std::vector<uint8_t>& mp) {
            K.N = (int)kp.size(); K.keysUn = kp.data(); K.descriptors = d.data(); K.hasMapPoint = mp.data();
            std::vector<std::vector<int32_t>> nodes(40);
            for (int i = 0; i < K.N; i++) { nodes[((d[(size_t)i * 32] >> 4) * 7 + (d[(size_t)i * 32 + 9] >> 5) * 3) % 40].push_back(i); if (i % 5 == 0) mp[i] = 1; }
            K.nodeStart.push_back(0);
            for (int n = 0; n < 40; n++)
                if (!nodes[n].empty()) { K.nodeId.push_back(n); K.featIdx.insert(K.featIdx.end(), nodes[n].begin(), nodes[n].end()); K.nodeStart.push_back((int32_t)K.featIdx.size()); }
        };
        build(K1, kA, dA, mp1); build(K2, kB, dB, mp2);
        const float F12[9] = {0, 0, -0.03f, 0, 0, -0.05f, 0.03f, 0.05f, 0}, ep[2] = {-5000.f, -5000.f};   // image B = A shifted by (5,-3)
        float sig2[8];
        for (int l = 0; l < 8; l++) sig2[l] = sf[l] * sf[l];
        std::vector<std::pair<size_t, size_t>> pairs;
        const int nt = m.SearchForTriangulation(K1, K2, F12, ep, sig2, sf.data(), 8, pairs, false, false);
        OTriSide o1{kA.data(), dA.data(), nullptr, mp1.data(), K1.nodeId.data(), K1.nodeStart.data(), K1.featIdx.data(), (int)K1.nodeId.size(), K1.N};
        OTriSide o2{kB.data(), dB.data(), nullptr, mp2.data(), K2.nodeId.data(), K2.nodeStart.data(), K2.featIdx.data(), (int)K2.nodeId.size(), K2.N};
        std::vector<int32_t> ot(kA.size());
        const int ont = omo_search_for_triangulation(&o1, &o2, F12, ep, sig2, sf.data(), 0, 0, 1, ot.data());
        CHECK(nt == ont && nt > 10 && (int)pairs.size() == nt);
        size_t pi = 0;
        for (size_t i = 0; i < ot.size(); i++)
            if (ot[i] >= 0) { CHECK(pairs[pi].first == i && pairs[pi].second == (size_t)ot[i]); pi++; }
        // SearchByBoW on the same CSRs, single camera and as a rig frame (features >= Nleft = right camera)
        std::vector<float> angA(kA.size()), angB(kB.size());
        for (size_t i = 0; i < kA.size(); i++) angA[i] = kA[i].angle;
        for (size_t i = 0; i < kB.size(); i++) angB[i] = kB[i].angle;
        std::vector<uint8_t> valid(kA.size(), 1);
        K1.hasMapPoint = valid.data();
        for (int nl : {-1, (int)kB.size() / 2}) {
            std::vector<int> fm;
            orbslam3_hip::ORBmatcher mb(0.7f, true);
            const int nb = mb.SearchByBoW(K1, angA.data(), K2, angB.data(), nl, fm);
            std::vector<int32_t> ofm(kB.size());
            const int onb = omo_search_by_bow(dA.data(), angA.data(), valid.data(), K1.nodeId.data(), K1.nodeStart.data(), K1.featIdx.data(), (int)K1.nodeId.size(),
                                              dB.data(), angB.data(), K2.N, K2.nodeId.data(), K2.nodeStart.data(), K2.featIdx.data(), (int)K2.nodeId.size(),
                                              0.7f, 1, ofm.data(), nl);
            CHECK(nb == onb && nb > 10 && std::memcmp(ofm.data(), fm.data(), ofm.size() * 4) == 0);
        }
        {   // SearchByBoW(KeyFrame*, KeyFrame*) (ORBmatcher.cc:984-1124): map-point validity on both sides, result indexed by key frame 1
            std::vector<uint8_t> v1(kA.size()), v2(kB.size());
            for (size_t i = 0; i < v1.size(); i++) v1[i] = (i % 7) != 2;
            for (size_t i = 0; i < v2.size(); i++) v2[i] = (i % 5) != 4;
            K1.hasMapPoint = v1.data(); K2.hasMapPoint = v2.data();
            orbslam3_hip::ORBmatcher mk(0.8f, true);
            std::vector<int> m12;
            const int nk = mk.SearchByBoW(K1, angA.data(), K2, angB.data(), m12);
            std::vector<int32_t> om12(kA.size());
            const int onk = omo_search_by_bow_kf(dA.data(), angA.data(), v1.data(), K1.nodeId.data(), K1.nodeStart.data(), K1.featIdx.data(), (int)K1.nodeId.size(), K1.N,
                                                 dB.data(), angB.data(), v2.data(), K2.nodeId.data(), K2.nodeStart.data(), K2.featIdx.data(), (int)K2.nodeId.size(), K2.N,
                                                 0.8f, 1, om12.data());
            CHECK(nk == onk && nk > 10 && std::memcmp(om12.data(), m12.data(), om12.size() * 4) == 0);
            K1.hasMapPoint = valid.data(); K2.hasMapPoint = mp2.data();
        }
        std::printf("adapter N1: init %d, fuse %d, triangulation %d\n", ni, nf, nt);
    }
    // ---- N2: ORBVocabulary::transform through the adapter vs the oracle (a 5-ary, 3-level vocabulary built around the frame's descriptors)
    {
        std::vector<uint8_t> blob(24);
        const int K = 5, LV = 3;
        std::vector<int> parentOf; std::vector<std::vector<uint8_t>> nd; std::vector<int> depth;
        std::vector<int> cur(1, 0);
        std::vector<std::vector<uint8_t>> dsc(1, std::vector<uint8_t>(32, 0));
        for (int lev = 1; lev <= LV; lev++) {
            std::vector<int> nxt;
            for (int p : cur)
                for (int c = 0; c < K; c++) {
                    std::vector<uint8_t> d = dsc[p];
                    if (lev == 1) { const size_t q = rnd() % kA.size(); d.assign(dA.begin() + 32 * q, dA.begin() + 32 * q + 32); }
                    for (int fl = 0; fl < (40 >> lev); fl++) { const unsigned bp = rnd() % 256; d[bp >> 3] ^= (uint8_t)(1u << (bp & 7)); }
                    parentOf.push_back(p); depth.push_back(lev); dsc.push_back(d); nxt.push_back((int)dsc.size() - 1);
                }
            cur = nxt;
        }
        const uint32_t nbn = (uint32_t)parentOf.size(), szn = 41;
        const int32_t hdr[4] = {K, LV, 0, 0};
        std::memcpy(&blob[0], &nbn, 4); std::memcpy(&blob[4], &szn, 4); std::memcpy(&blob[8], hdr, 16);
        for (uint32_t i = 0; i < nbn; i++) {
            uint8_t rec[41];
            const int32_t par = parentOf[i];
            const float w = depth[i] == LV ? (i % 13 == 0 ? 0.f : 0.5f + (float)(rnd() % 800) / 100.f) : 0.f;
            std::memcpy(rec, &par, 4); std::memcpy(rec + 4, dsc[i + 1].data(), 32); std::memcpy(rec + 36, &w, 4); rec[40] = depth[i] == LV;
            blob.insert(blob.end(), rec, rec + 41);
        }
        orbslam3_hip::ORBVocabulary voc;
        CHECK(voc.loadFromMemory(blob.data(), blob.size()) && voc.getBranchingFactor() == K && voc.getDepthLevels() == LV);
        orbslam3_hip::BowVector bv; orbslam3_hip::FeatureVector fv;
        const int NF = (int)kA.size();
        voc.transform(dA.data(), NF, bv, fv, 1);
        void* ov = obw_load_binary(blob.data(), blob.size());
        CHECK(ov != nullptr);
        std::vector<int32_t> w1(NF), n1(NF), fn(NF), fs(NF + 1), ff(NF), bw(NF); std::vector<double> wt(NF), bvv(NF); int32_t nn = 0;
        const int nbv = obw_transform(ov, dA.data(), NF, 1, w1.data(), n1.data(), wt.data(), fn.data(), fs.data(), ff.data(), &nn, bw.data(), bvv.data());
        obw_destroy(ov);
        CHECK((int)bv.size() == nbv && (int)fv.size() == nn && nbv > 10 && nn > 5);
        int i = 0;
        for (auto& kv : bv) { CHECK((int)kv.first == bw[i] && kv.second == bvv[i]); i++; }
        i = 0;
        for (auto& kv : fv) {
            CHECK((int)kv.first == fn[i] && (int)kv.second.size() == fs[i + 1] - fs[i]);
            for (size_t j = 0; j < kv.second.size(); j++) CHECK((int)kv.second[j] == ff[fs[i] + j]);
            i++;
        }
        std::printf("adapter N2: %d words, %d feature-vector nodes\n", nbv, nn);
    }
    // ---- stage 3: a toy window (4 KFs, first fixed; 30 points) through LbaLinearizer
    orbslam3_hip::LbaLinearizer L;
    lba_camera cam{};
    cam.model = LBA_CAM_PINHOLE; cam.p[0] = 458.654f; cam.p[1] = 457.296f; cam.p[2] = 367.215f; cam.p[3] = 248.375f; cam.bf = 47.906f; cam.trl_q[3] = 1;
    L.addCamera(cam);
    std::vector<double> poses, points;
    std::vector<int32_t> hidx;
    for (int k = 0; k < 4; k++) {
        const float a = 0.05f * k;
        const float Tcw[12] = {std::cos(a), 0, std::sin(a), 0.1f * k, 0, 1, 0, 0.02f * k, -std::sin(a), 0, std::cos(a), 0.03f * k};
        double p7[7];
        orbslam3_hip::LbaLinearizer::poseFromTcw(Tcw, 4, p7);
        L.addPose(p7, k == 0);
        poses.insert(poses.end(), p7, p7 + 7);
        hidx.push_back(k == 0 ? -1 : k - 1);
    }
    std::vector<lba_edge> edges;
    for (int l = 0; l < 30; l++) {
        const double X[3] = {((int)(rnd() % 400) - 200) / 100.0, ((int)(rnd() % 300) - 150) / 100.0, 4.0 + (rnd() % 300) / 100.0};
        L.addPoint(X);
        points.insert(points.end(), X, X + 3);
        for (int k = 0; k < 4; k++) {
            if ((l + k) % 5 == 0) continue;
            const int kind = (l % 3 == 0) ? LBA_EDGE_STEREO : LBA_EDGE_MONO;
            const float u = 367.f + 80.f * (float)X[0] + (float)(rnd() % 7) - 3.f, v = 248.f + 80.f * (float)X[1] + (float)(rnd() % 7) - 3.f;
            const float s2 = 1.0f / (1.44f * (1 + k % 3));
            L.addEdge(k, l, kind, 0, u, v, u - 9.f, s2);
            edges.push_back(lba_edge{k, l, (int16_t)kind, 0, {u, v, u - 9.f}, s2});
        }
    }
    orbslam3_hip::LbaHostSystem S;
    L.buildSystem(S);
    const int ne = (int)edges.size();
    std::vector<double> Hpp(3 * 36), bp(3 * 6), Hll(30 * 9), bl(30 * 3), Hpl((size_t)ne * 18), err((size_t)ne * 3), chi2(ne), rho((size_t)ne * 2), depth(ne);
    double rs = 0;
    olb_build_system(poses.data(), hidx.data(), 4, points.data(), 30, edges.data(), ne, &cam, (double)std::sqrt(5.991f), (double)std::sqrt(7.815f),
                     Hpp.data(), bp.data(), Hll.data(), bl.data(), Hpl.data(), err.data(), chi2.data(), rho.data(), depth.data(), &rs);
    auto close = [](const std::vector<double>& a, const double* b, size_t n) {
        double mx = 1e-300, d = 0;
        for (size_t i = 0; i < n; i++) { mx = std::max(mx, std::fabs(b[i])); d = std::max(d, std::fabs(a[i] - b[i])); }
        return d / mx < 1e-10;
    };
    CHECK(close(S.Hpp, Hpp.data(), 3 * 36) && close(S.bp, bp.data(), 18) && close(S.Hll, Hll.data(), 270) && close(S.bl, bl.data(), 90));
    CHECK(close(S.Hpl, Hpl.data(), (size_t)ne * 18) && close(S.chi2, chi2.data(), ne) && close(S.err, err.data(), (size_t)ne * 3));
    orbslam3_hip::LbaHostSystem S2;
    L.computeErrors(S2);
    CHECK(std::fabs(S2.robustChi2 - rs) < 1e-9 * rs);
    // optimizer.optimize(5) through the adapter vs the oracle's LM
    std::vector<double> op = poses, ox = points;
    double ostats[4];
    olb_optimize(op.data(), hidx.data(), 4, ox.data(), 30, edges.data(), ne, &cam, (double)std::sqrt(5.991f), (double)std::sqrt(7.815f), 5, ostats);
    double chiFinal = 0;
    const int its = L.optimize(5, nullptr, &chiFinal);
    CHECK(its == (int)ostats[0] && std::fabs(chiFinal - ostats[1]) < 1e-6 * ostats[1] && ostats[1] < 0.99 * rs);
    for (int k = 0; k < 4; k++) for (int c = 0; c < 7; c++) CHECK(std::fabs(L.pose(k)[c] - op[k * 7 + c]) < 1e-7);
    // ---- N3: Optimizer::PoseOptimization through PoseOptimizer vs the oracle
    orbslam3_hip::PoseOptimizer PO;
    PO.addCamera(cam);
    std::vector<pose_edge> pe;
    for (int l = 0; l < 120; l++) {
        const float X[3] = {((int)(rnd() % 400) - 200) / 100.0f, ((int)(rnd() % 300) - 150) / 100.0f, 4.0f + (rnd() % 300) / 100.0f};
        float u = 367.215f + 458.654f * X[0] / X[2] + ((int)(rnd() % 200) - 100) / 100.0f, v = 248.375f + 457.296f * X[1] / X[2] + ((int)(rnd() % 200) - 100) / 100.0f;
        if (l % 11 == 0) { u += 35.f; v -= 20.f; }
        const float s2 = 1.0f / (1.44f * (1 + l % 3)), uR = u - 47.906f / X[2];
        if (l % 2) PO.addStereo(X, u, v, uR, s2); else PO.addMono(X, u, v, s2);
        pe.push_back(pose_edge{{X[0], X[1], X[2]}, {u, v, l % 2 ? uR : 0.f}, s2, (int16_t)(l % 2 ? LBA_EDGE_STEREO : LBA_EDGE_MONO), 0});
    }
    const float Tcw0[12] = {std::cos(0.01f), 0, std::sin(0.01f), 0.03f, 0, 1, 0, -0.02f, -std::sin(0.01f), 0, std::cos(0.01f), 0.04f};
    double pp[7], opp[7];
    orbslam3_hip::LbaLinearizer::poseFromTcw(Tcw0, 4, pp);
    std::vector<uint8_t> oout(pe.size());
    const int ogood = opo_pose_optimize(pp, pe.data(), (int)pe.size(), &cam, opp, oout.data());
    std::vector<bool> outl;
    const int good = PO.optimize(pp, outl);
    CHECK(good == ogood && good > 100 && good < 120);
    for (size_t i = 0; i < pe.size(); i++) CHECK(outl[i] == (oout[i] != 0));
    for (int c = 0; c < 7; c++) CHECK(std::fabs(pp[c] - opp[c]) < 1e-7);
    CHECK(std::fabs(pp[0]) < 0.01 && std::fabs(pp[1]) < 0.01 && std::fabs(pp[2]) < 0.02);   // pulled back to identity
    // ---- Frame constructor steps through FrameOps vs the oracle (EuRoC calibration, Examples/Monocular/EuRoC.yaml)
    {
        const float cam9[9] = {458.654f, 457.296f, 367.215f, 248.375f, -0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f, 0.0f};
        orbslam3_hip::FrameOps FO(cam9[0], cam9[1], cam9[2], cam9[3], {cam9[4], cam9[5], cam9[6], cam9[7]}, 752, 480);
        float ob[6];
        ofr_image_bounds(cam9, 752, 480, ob);
        CHECK(FO.mnMinX == ob[0] && FO.mnMaxX == ob[1] && FO.mnMinY == ob[2] && FO.mnMaxY == ob[3]);
        CHECK(FO.mfGridElementWidthInv == ob[4] && FO.mfGridElementHeightInv == ob[5] && FO.mnMinX < -30.f);
        std::vector<orb_keypoint> ks(500), un, oun(500);
        for (auto& k : ks) { k = orb_keypoint{(float)(16 + rnd() % 720), (float)(16 + rnd() % 448), 31.f, (float)(rnd() % 360), (float)(rnd() % 200), (int)(rnd() % 8), -1}; }
        FO.UndistortKeyPoints(ks, un);
        ofr_undistort_keypoints(ks.data(), 500, cam9, oun.data());
        CHECK(un.size() == 500 && std::memcmp(un.data(), oun.data(), 500 * sizeof(orb_keypoint)) == 0);
        std::vector<float> depth((size_t)752 * 480), ur, dz, our(500), odz(500);
        for (auto& d : depth) d = (rnd() % 5 == 0) ? 0.0f : (float)(rnd() % 6000) / 1000.0f;
        FO.ComputeStereoFromRGBD(ks, un, depth.data(), 752, 40.0f, ur, dz);
        ofr_stereo_from_rgbd(ks.data(), oun.data(), 500, depth.data(), 752, 40.0f, our.data(), odz.data());
        CHECK(std::memcmp(ur.data(), our.data(), 500 * 4) == 0 && std::memcmp(dz.data(), odz.data(), 500 * 4) == 0);
    }
    // ---- Frame::ComputeStereoFishEyeMatches through FisheyeStereoMatcher vs the oracle (two KB8 cameras 10 cm apart, planted correspondences)
    {
        orbf_fisheye_rig rg{};
        const float kb[8] = {190.978f, 190.973f, 254.932f, 256.897f, 0.0034823894f, 0.0007150348f, -0.0020532361f, 0.00020293673f};
        for (int i = 0; i < 8; i++) { rg.k_left[i] = kb[i]; rg.k_right[i] = kb[i]; }
        const float I9f[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int i = 0; i < 9; i++) rg.R_lr[i] = I9f[i];
        rg.t_lr[0] = 0.1f;
        for (int l = 0; l < 8; l++) rg.level_sigma2[l] = std::pow(1.44f, (float)l);
        auto kbproj = [&](double X, double Y, double Z, float& u, float& v) {
            const double r = std::sqrt(X * X + Y * Y), th = std::atan2(r, Z), psi = std::atan2(Y, X);
            const double th2 = th * th, rd = th * (1 + kb[4] * th2 + kb[5] * th2 * th2 + kb[6] * th2 * th2 * th2 + kb[7] * th2 * th2 * th2 * th2);
            u = (float)(kb[0] * rd * std::cos(psi) + kb[2]); v = (float)(kb[1] * rd * std::sin(psi) + kb[3]);
        };
        const int NS = 90, ML = 7, MR = 4;
        std::vector<orb_keypoint> fl(ML + NS), fr(MR + NS);
        std::vector<uint8_t> fdl((ML + NS) * 32), fdr((MR + NS) * 32);
        for (auto& b8 : fdl) b8 = (uint8_t)rnd();
        for (auto& b8 : fdr) b8 = (uint8_t)rnd();
        for (int i = 0; i < ML; i++) fl[i] = orb_keypoint{(float)(30 + rnd() % 400), (float)(30 + rnd() % 400), 31.f, 0.f, 50.f, (int)(rnd() % 8), -1};
        for (int i = 0; i < MR; i++) fr[i] = orb_keypoint{(float)(30 + rnd() % 400), (float)(30 + rnd() % 400), 31.f, 0.f, 50.f, (int)(rnd() % 8), -1};
        for (int i = 0; i < NS; i++) {
            const double X = ((int)(rnd() % 4000) - 2000) / 1000.0, Y = ((int)(rnd() % 3000) - 1500) / 1000.0, Z = 1.2 + (rnd() % 3000) / 1000.0;
            float u, v;
            kbproj(X, Y, Z, u, v);       fl[ML + i] = orb_keypoint{u, v, 31.f, 0.f, 50.f, (int)(rnd() % 8), -1};
            kbproj(X - 0.1, Y, Z, u, v); fr[MR + (i * 7) % NS] = orb_keypoint{u, v, 31.f, 0.f, 50.f, (int)(rnd() % 8), -1};
            std::memcpy(&fdr[(size_t)(MR + (i * 7) % NS) * 32], &fdl[(size_t)(ML + i) * 32], 32);
            fdr[(size_t)(MR + (i * 7) % NS) * 32 + (i % 32)] ^= (uint8_t)(1u << (i % 8));
        }
        orbslam3_hip::FisheyeStereoMatcher FM(rg);
        std::vector<int> l2r, r2l; std::vector<float> dep, p3;
        const int nm2 = FM.ComputeStereoFishEyeMatches(fl, fdl.data(), ML, fr, fdr.data(), MR, l2r, r2l, dep, p3);
        std::vector<int32_t> ol2r(fl.size()), or2l(fr.size()); std::vector<float> odep(fl.size()), op3(fl.size() * 3);
        float rigarr[28];
        for (int i = 0; i < 8; i++) { rigarr[i] = rg.k_left[i]; rigarr[8 + i] = rg.k_right[i]; }
        for (int i = 0; i < 9; i++) rigarr[16 + i] = rg.R_lr[i];
        for (int i = 0; i < 3; i++) rigarr[25 + i] = rg.t_lr[i];
        const int onm2 = ofr_stereo_fisheye(fl.data(), fdl.data(), (int)fl.size(), ML, fr.data(), fdr.data(), (int)fr.size(), MR, rigarr, rg.level_sigma2, ol2r.data(),
                                            or2l.data(), odep.data(), op3.data());
        CHECK(nm2 == onm2 && nm2 > 60);
        for (size_t i = 0; i < fl.size(); i++) { CHECK(l2r[i] == ol2r[i]); CHECK(std::fabs(dep[i] - odep[i]) <= 2e-6f * std::fabs(odep[i])); }
        for (size_t i = 0; i < fr.size(); i++) CHECK(r2l[i] == or2l[i]);
        for (int i = 0; i < ML; i++) CHECK(l2r[i] == -1);
    }
    // ---- N4 tail: Optimizer::LocalInertialBA through InertialBA vs the oracle
    {
        orbslam3_hip::InertialBA IB;
        liba_rig rig{};
        rig.n_cams = 1; rig.bf = 47.906;
        const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int i = 0; i < 9; i++) { rig.Rcb[0][i] = I9[i]; rig.Rbc[0][i] = I9[i]; }
        rig.tcb[0][0] = 0.05; rig.tbc[0][0] = -0.05;
        rig.model[0] = LBA_CAM_PINHOLE;
        const double fxd = 458.654, fyd = 457.296, cxd = 367.215, cyd = 248.375;
        rig.p[0][0] = fxd; rig.p[0][1] = fyd; rig.p[0][2] = cxd; rig.p[0][3] = cyd;
        IB.setRig(rig);
        const int NK = 6;
        const double dtk = 0.3;
        auto nz = [&](double s) { return s * ((int)(rnd() % 2001) - 1000) / 1000.0; };
        std::vector<liba_keyframe> okf;
        for (int k = 0; k < NK; k++) {
            const bool fixed = k == 0;
            double twb[3] = {0.3 * k + (fixed ? 0 : nz(0.02)), fixed ? 0 : nz(0.02), fixed ? 0 : nz(0.02)};
            double v[3] = {1.0 + (fixed ? 0 : nz(0.05)), fixed ? 0 : nz(0.05), fixed ? 0 : nz(0.05)};
            double bg[3] = {nz(0.001), nz(0.001), nz(0.001)}, ba[3] = {nz(0.01), nz(0.01), nz(0.01)};
            double tcw[3] = {-twb[0] + rig.tcb[0][0], -twb[1], -twb[2]};   // Rcw = Rcb * Rbw = I; tcw = Rcb * (-Rbw twb) + tcb
            IB.addKeyFrame(I9, twb, I9, tcw, nullptr, nullptr, v, bg, ba, fixed, true, fixed);
            okf.push_back(IB.keyFrame(k));
        }
        std::vector<liba_imu_edge> oimu;
        for (int i = 0; i < NK - 1; i++) {     // newest first
            liba_imu_edge e{};
            e.kf2 = NK - 1 - i; e.kf1 = e.kf2 - 1;
            e.dT = (float)dtk;
            for (int q = 0; q < 9; q++) { e.dR[q] = (float)I9[q]; e.JRg[q] = (float)(-dtk * I9[q]); e.JVa[q] = (float)(-dtk * I9[q]); e.JPa[q] = (float)(-0.5 * dtk * dtk * I9[q]); }
            e.dV[2] = (float)(9.81 * dtk); e.dP[2] = (float)(0.5 * 9.81 * dtk * dtk);      // exact deltas of the constant-velocity truth
            for (int q = 0; q < 9; q++) { e.info[q * 10] = q < 3 ? 3e4 : (q < 6 ? 2e3 : 8e3); }
            for (int q = 0; q < 3; q++) { e.info_g[q * 4] = 4e5; e.info_a[q * 4] = 2e3; }
            if (i == NK - 2) { e.huber = std::sqrt(16.92); for (int q = 0; q < 81; q++) e.info[q] *= 1e-2; }
            IB.addInertial(e);
            oimu.push_back(e);
        }
        std::vector<double> opts;
        std::vector<lba_edge> oedges;
        for (int l = 0; l < 80; l++) {
            const float X[3] = {(float)(0.75 + nz(3.0)), (float)nz(2.0), (float)(6.0 + nz(2.0))};
            const int pi = IB.addPoint(X);
            for (int c = 0; c < 3; c++) opts.push_back((double)X[c]);
            for (int k = 0; k < NK; k++) {
                const double xc = X[0] - 0.3 * k + 0.05, yc = X[1], zc = X[2];   // truth: twb = (0.3k, 0, 0)
                const float u = (float)(fxd * xc / zc + cxd + nz(0.7)), v = (float)(fyd * yc / zc + cyd + nz(0.7));
                if (u < 0 || u >= 752 || v < 0 || v >= 480) continue;
                if (l % 2) { IB.addStereo(k, pi, u, v, (float)(u - 47.906 / zc), 1.0f); oedges.push_back(lba_edge{k, pi, LBA_EDGE_STEREO, 0, {u, v, (float)(u - 47.906 / zc)}, 1.0f}); }
                else { IB.addMono(k, pi, u, v, 1.0f); oedges.push_back(lba_edge{k, pi, LBA_EDGE_MONO, 0, {u, v, 0.f}, 1.0f}); }
            }
        }
        double ostats[5], err = 0, errEnd = 0;
        oib_optimize(okf.data(), NK, &rig, opts.data(), 80, oedges.data(), (int)oedges.size(), oimu.data(), (int)oimu.size(), (double)std::sqrt(5.991f),
                     (double)std::sqrt(7.815f), 1.0, 10, ostats);
        const int its2 = IB.optimize(1.0, 10, &err, &errEnd);
        CHECK(its2 == (int)ostats[0] && its2 >= 3);
        CHECK(std::fabs(err - ostats[4]) < 1e-9 * ostats[4] && std::fabs(errEnd - ostats[1]) < 1e-6 * ostats[1] && errEnd < 0.5 * err);
        {
            double dt_ = 0, dv_ = 0, dbg_ = 0, dba_ = 0;
            for (int k = 0; k < NK; k++) for (int c = 0; c < 3; c++) {
                dt_ = std::max(dt_, std::fabs(IB.keyFrame(k).twb[c] - okf[k].twb[c])); dv_ = std::max(dv_, std::fabs(IB.keyFrame(k).v[c] - okf[k].v[c]));
                dbg_ = std::max(dbg_, std::fabs(IB.keyFrame(k).bg[c] - okf[k].bg[c])); dba_ = std::max(dba_, std::fabs(IB.keyFrame(k).ba[c] - okf[k].ba[c]));
            }
            std::printf("adapter inertial BA: its %d, max |d twb| %.3g, |d v| %.3g, |d bg| %.3g, |d ba| %.3g\n", its2, dt_, dv_, dbg_, dba_);
        }
        for (int k = 0; k < NK; k++)
            for (int c = 0; c < 3; c++) {
                // float-rounded ExpSO3 (rule R3) + device libm: a few 1e-6 after 10 LM iterations on this small window (bar: 1e-4)
                CHECK(std::fabs(IB.keyFrame(k).twb[c] - okf[k].twb[c]) < 5e-6 && std::fabs(IB.keyFrame(k).v[c] - okf[k].v[c]) < 5e-6);
                CHECK(std::fabs(IB.keyFrame(k).bg[c] - okf[k].bg[c]) < 5e-6 && std::fabs(IB.keyFrame(k).ba[c] - okf[k].ba[c]) < 5e-6);
            }
        CHECK(std::fabs(IB.keyFrame(0).twb[0]) == 0.0 && std::fabs(IB.keyFrame(3).twb[0] - 0.9) < 0.02);   // fixed KF untouched, others pulled to the truth
        for (int l = 0; l < 80; l++) for (int c = 0; c < 3; c++) CHECK(std::fabs(IB.point(l)[c] - opts[(size_t)l * 3 + c]) < 5e-5);
        int nout = 0;
        for (size_t e = 0; e < oedges.size(); e++) { CHECK(IB.depthPositive((int)e)); nout += IB.visualChi2((int)e) > 5.991; }
        CHECK(nout < (int)oedges.size() / 10);
    }
    // ---- Optimizer::PoseInertialOptimizationLastKeyFrame through PoseInertialOptimizer vs the oracle
    {
        liba_rig rig{};
        rig.n_cams = 1; rig.bf = 47.906;
        const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int i = 0; i < 9; i++) { rig.Rcb[0][i] = I9[i]; rig.Rbc[0][i] = I9[i]; }
        rig.model[0] = LBA_CAM_PINHOLE;
        rig.p[0][0] = 458.654; rig.p[0][1] = 457.296; rig.p[0][2] = 367.215; rig.p[0][3] = 248.375;
        auto nz = [&](double s2) { return s2 * ((int)(rnd() % 2001) - 1000) / 1000.0; };
        liba_keyframe kf{}, fr{};
        for (int i = 0; i < 9; i++) { kf.Rwb[i] = I9[i]; kf.Rcw[0][i] = I9[i]; fr.Rwb[i] = I9[i]; fr.Rcw[0][i] = I9[i]; }
        kf.v[0] = 1.0; kf.has_imu = 1; kf.pose_fixed = 1; kf.imu_fixed = 1;
        fr.twb[0] = 0.3 + 0.02; fr.twb[1] = -0.015; fr.twb[2] = 0.01; fr.v[0] = 1.03; fr.v[1] = -0.02; fr.has_imu = 1;
        for (int i = 0; i < 3; i++) { fr.tcw[0][i] = -fr.twb[i]; fr.bg[i] = nz(0.001); fr.ba[i] = nz(0.01); kf.bg[i] = nz(0.001); kf.ba[i] = nz(0.01); }
        liba_imu_edge pe2{};
        pe2.kf1 = 0; pe2.kf2 = 1; pe2.dT = 0.3f;
        for (int q = 0; q < 9; q++) { pe2.dR[q] = (float)I9[q]; pe2.JRg[q] = (float)(-0.3 * I9[q]); pe2.JVa[q] = (float)(-0.3 * I9[q]); pe2.JPa[q] = (float)(-0.045 * I9[q]); }
        pe2.dV[2] = (float)(9.81 * 0.3); pe2.dP[2] = (float)(0.5 * 9.81 * 0.09);
        for (int q = 0; q < 9; q++) pe2.info[q * 10] = q < 3 ? 3e4 : (q < 6 ? 2e3 : 8e3);
        for (int q = 0; q < 3; q++) { pe2.info_g[q * 4] = 4e5; pe2.info_a[q * 4] = 2e3; }
        orbslam3_hip::PoseInertialOptimizer PIO;
        PIO.setRig(rig);
        std::vector<pose_edge> oe;
        for (int l = 0; l < 150; l++) {
            const float X[3] = {(float)(0.3 + nz(3.0)), (float)nz(2.0), (float)(5.0 + nz(2.0))};
            float u = (float)(458.654 * (X[0] - 0.3) / X[2] + 367.215 + nz(0.8)), v = (float)(457.296 * X[1] / X[2] + 248.375 + nz(0.8));   // truth: twb = (0.3, 0, 0)
            if (l % 13 == 0) { u += 30.f; v -= 18.f; }
            const bool close = l % 3 != 0;
            if (l % 2) { PIO.addStereo(X, u, v, (float)(u - 47.906 / X[2]), 1.0f); oe.push_back(pose_edge{{X[0], X[1], X[2]}, {u, v, (float)(u - 47.906 / X[2])}, 1.0f, LBA_EDGE_STEREO, 0}); }
            else { PIO.addMono(X, u, v, 1.0f, close); oe.push_back(pose_edge{{X[0], X[1], X[2]}, {u, v, 0.f}, 1.0f, (int16_t)(LBA_EDGE_MONO | (close ? LIBA_EDGE_CLOSE : 0)), 0}); }
        }
        liba_keyframe ofr = fr;
        std::vector<uint8_t> oout(oe.size());
        double oH[225], H15[225];
        const int ogood = oib_pose_inertial_kf(&ofr, &kf, &rig, oe.data(), (int)oe.size(), &pe2, 0, oout.data(), oH);
        std::vector<bool> outl2;
        const int good2 = PIO.optimize(fr, kf, pe2, false, outl2, H15);
        CHECK(good2 == ogood && good2 > 120 && good2 < 150);
        for (size_t i = 0; i < oe.size(); i++) CHECK(outl2[i] == (oout[i] != 0));
        for (int c = 0; c < 3; c++) { CHECK(std::fabs(fr.twb[c] - ofr.twb[c]) < 5e-6 && std::fabs(fr.v[c] - ofr.v[c]) < 5e-6 && std::fabs(fr.bg[c] - ofr.bg[c]) < 5e-6); }
        CHECK(std::fabs(fr.twb[0] - 0.3) < 0.01 && std::fabs(fr.twb[1]) < 0.01);
        double hmax = 0;
        for (int i = 0; i < 225; i++) hmax = std::max(hmax, std::fabs(oH[i]));
        for (int i = 0; i < 225; i++) CHECK(std::fabs(H15[i] - oH[i]) <= 1e-6 * hmax);
        // ... and PoseInertialOptimizationLastFrame: the key frame plays the previous frame, with a diagonal prior at its state
        liba_prior pr{};
        for (int i = 0; i < 9; i++) pr.Rwb[i] = kf.Rwb[i];
        for (int i = 0; i < 3; i++) { pr.twb[i] = kf.twb[i] + 0.002 * (i + 1); pr.vwb[i] = kf.v[i] - 0.004; pr.bg[i] = kf.bg[i]; pr.ba[i] = kf.ba[i] + 0.001; }
        for (int i = 0; i < 15; i++) pr.H[i * 16] = i < 3 ? 2e4 : (i < 6 ? 5e3 : (i < 9 ? 1e3 : (i < 12 ? 3e5 : 2e3)));
        liba_keyframe f2 = ofr, p2 = kf, of2 = ofr, op2 = kf;
        f2.twb[0] += 0.01; of2.twb[0] += 0.01;
        const int og2 = oib_pose_inertial_lastframe(&of2, &op2, &rig, oe.data(), (int)oe.size(), &pe2, &pr, 0, oout.data(), oH);
        const int g2 = PIO.optimizeLastFrame(f2, p2, pr, pe2, false, outl2, H15);
        CHECK(g2 == og2 && g2 > 100);
        for (size_t i = 0; i < oe.size(); i++) CHECK(outl2[i] == (oout[i] != 0));
        for (int c = 0; c < 3; c++) { CHECK(std::fabs(f2.twb[c] - of2.twb[c]) < 5e-6 && std::fabs(p2.twb[c] - op2.twb[c]) < 5e-6 && std::fabs(p2.v[c] - op2.v[c]) < 5e-6); }
        hmax = 0;
        for (int i = 0; i < 225; i++) hmax = std::max(hmax, std::fabs(oH[i]));
        for (int i = 0; i < 225; i++) CHECK(std::fabs(H15[i] - oH[i]) <= 1e-5 * hmax);
    }
    std::printf("adapter_test OK: %d keypoints, %d matches, %d LBA edges, LM chi2 %.1f -> %.1f\n", n, nm, ne, rs, chiFinal);
    return 0;
}
