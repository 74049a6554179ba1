// orbslam3_hip/ORBVocabulary.h — adapter for ORB_SLAM3::ORBVocabulary (reference include/ORBVocabulary.h:31-32 =
// DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) over liborbhip.so (include/orbhip.h, "SURVEY.md N2").
// Covers what the hot path uses: loadFromBinaryFile (System.cc:83) and transform(features, BowVector&, FeatureVector&, levelsup)
// (Frame::ComputeBoW, Frame.cc:865-872; KeyFrame::ComputeBoW KeyFrame.cc:88-97).  BowVector / FeatureVector keep DBoW2's
// container types (std::map<WordId, WordValue>, std::map<NodeId, std::vector<unsigned int>>).
#ifndef ORBSLAM3_HIP_ORBVOCABULARY_H
#define ORBSLAM3_HIP_ORBVOCABULARY_H
#include <cstdint>
#include <fstream>
#include <iterator>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../orbhip.h"
#include "ORBmatcher.h"  // detail::DevBuf

namespace orbslam3_hip {

typedef std::map<unsigned int, double> BowVector;                          // DBoW2::BowVector     (BowVector.h:58-60)
typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector;   // DBoW2::FeatureVector (FeatureVector.h:21-23)

class ORBVocabulary {
public:
    ORBVocabulary() = default;
    ORBVocabulary(const ORBVocabulary&) = delete;
    ORBVocabulary& operator=(const ORBVocabulary&) = delete;
    ~ORBVocabulary() { if (h_) bow_vocab_destroy(h_); }

    // TemplatedVocabulary::loadFromBinaryFile (TemplatedVocabulary.h:1442-1480)
    bool loadFromBinaryFile(const std::string& filename, int device = 0) {
        std::ifstream f(filename.c_str(), std::ios::in | std::ios::binary);
        if (!f) return false;
        std::vector<char> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        return loadFromMemory(bytes.data(), bytes.size(), device);
    }
    bool loadFromMemory(const void* bytes, size_t n, int device = 0) {
        if (h_) { bow_vocab_destroy(h_); h_ = nullptr; }
        if (bow_vocab_load_binary(bytes, n, device, &h_) != ORB_OK) return false;
        int32_t info[6];
        bow_vocab_info(h_, info);
        k_ = info[0]; L_ = info[1];
        return true;
    }
    bool empty() const { return h_ == nullptr; }
    int getBranchingFactor() const { return k_; }
    int getDepthLevels() const { return L_; }

    // transform(features, v, fv, levelsup): `descriptors` = N x 32 bytes (mDescriptors rows)
    void transform(const uint8_t* descriptors, int N, BowVector& v, FeatureVector& fv, int levelsup) {
        v.clear();
        fv.clear();
        if (!h_ || N <= 0) return;
        if (N > 4096) throw std::runtime_error("ORBVocabulary::transform: more than 4096 features");
        const uint8_t* dd = desc_.upload(descriptors, (size_t)N * 32);
        const int32_t* dn = n_.upload(&N, 1);
        bow_result R;
        R.word_id = (int32_t*)b_[0].ensure((size_t)N * 4); R.node_id = (int32_t*)b_[1].ensure((size_t)N * 4); R.weight = (double*)b_[2].ensure((size_t)N * 8);
        R.fv_node_id = (int32_t*)b_[3].ensure((size_t)N * 4); R.fv_node_start = (int32_t*)b_[4].ensure((size_t)(N + 1) * 4);
        R.fv_feat_idx = (int32_t*)b_[5].ensure((size_t)N * 4); R.fv_n_nodes = (int32_t*)b_[6].ensure(4);
        R.bv_word = (int32_t*)b_[7].ensure((size_t)N * 4); R.bv_value = (double*)b_[8].ensure((size_t)N * 8); R.bv_n = (int32_t*)b_[9].ensure(4);
        if (bow_transform(h_, dd, dn, 1, N, 1, levelsup, &R, nullptr) != ORB_OK) throw std::runtime_error("bow_transform");
        std::vector<int32_t> nid(N), nst(N + 1), fid(N), bw(N);
        std::vector<double> bv(N);
        int32_t nn = 0, nb = 0;
        orb_memcpy_d2h(nid.data(), R.fv_node_id, (size_t)N * 4, nullptr); orb_memcpy_d2h(nst.data(), R.fv_node_start, (size_t)(N + 1) * 4, nullptr);
        orb_memcpy_d2h(fid.data(), R.fv_feat_idx, (size_t)N * 4, nullptr); orb_memcpy_d2h(&nn, R.fv_n_nodes, 4, nullptr);
        orb_memcpy_d2h(bw.data(), R.bv_word, (size_t)N * 4, nullptr); orb_memcpy_d2h(bv.data(), R.bv_value, (size_t)N * 8, nullptr);
        orb_memcpy_d2h(&nb, R.bv_n, 4, nullptr);
        if (orb_stream_sync(nullptr) != ORB_OK) throw std::runtime_error("orb_stream_sync");
        for (int i = 0; i < nb; i++) v.insert(v.end(), std::make_pair((unsigned)bw[i], bv[i]));
        for (int k = 0; k < nn; k++) {
            std::vector<unsigned int>& feats = fv[(unsigned)nid[k]];
            feats.assign(fid.begin() + nst[k], fid.begin() + nst[k + 1]);
        }
    }

private:
    bow_vocab_handle h_ = nullptr;
    int k_ = 0, L_ = 0;
    detail::DevBuf desc_, n_, b_[10];
};

}  // namespace orbslam3_hip
#endif
