// bow_oracle.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
// CPU restatement of the DBoW2 pieces on the path between ORBextractor and ORBmatcher::SearchByBoW (SURVEY.md "next" row N2):
//   TemplatedVocabulary::loadFromBinaryFile   Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1442-1480  (System.cc:83 loads this format)
//   TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)   :1137-1206   (Frame::ComputeBoW, Frame.cc:865-872)
//   TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)          :1231-1272
//   FORB::distance  FORB.cpp:81-101 ; BowVector::addWeight / addIfNotExist / normalize  BowVector.cpp:34-84 ; FeatureVector::addFeature
//   ScoringObject.h:53-89 (which scorings normalise, and with which norm)
// DBoW2 is vendored in the reference tree but needs OpenCV (cv::Mat descriptors) -> unbuildable here; the data structures below use
// std::map / std::vector exactly like the original so that insertion and summation order are the reference's.
// PARITY: pinned only by this restatement's own invariants (tests/test_bow_parity.py); the reference ships no vocabulary blob
// (.MISSING_LARGE_BLOBS) and no tests -> "parity unpinned" against upstream outputs.
// Note on loadFromBinaryFile's `while(!f.eof())` loop: after the last record one more iteration runs with a failed read; it re-adds the
// last record as node nb_nodes+1 (an out-of-bounds write in the original).  That phantom child carries the same descriptor as the real
// last node and comes later in its parent's list, so the strict `d < best_d` scan can never select it: results are unaffected, and the
// restatement (like the product) simply stops at nb_nodes records.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

namespace {
enum LNorm { L1, L2 };
enum WeightingType { TF_IDF, TF, IDF, BINARY };
enum ScoringType { L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT };

struct Node {
    unsigned id = 0;
    double weight = 0;
    std::vector<unsigned> children;
    unsigned parent = 0;
    uint8_t descriptor[32] = {0};
    unsigned word_id = 0;
    bool isLeaf() const { return children.empty(); }
};

struct Vocabulary {
    int m_k = 0, m_L = 0, m_scoring = 0, m_weighting = 0;
    std::vector<Node> m_nodes;
    std::vector<unsigned> m_words;   // word id -> node id
    bool mustNormalize(LNorm& norm) const {   // ScoringObject.h:74-89
        norm = m_scoring == L2_NORM ? L2 : L1;
        return m_scoring != DOT_PRODUCT;
    }
};

int distance(const uint8_t* a, const uint8_t* b) {   // FORB.cpp:81-101
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

void transformOne(const Vocabulary& V, const uint8_t* feature, unsigned& word_id, double& weight, unsigned* nid, int levelsup) {   // :1231-1272
    std::vector<unsigned> nodes;
    const int nid_level = V.m_L - levelsup;
    if (nid_level <= 0 && nid != nullptr) *nid = 0;
    unsigned final_id = 0;
    int current_level = 0;
    do {
        ++current_level;
        nodes = V.m_nodes[final_id].children;
        final_id = nodes[0];
        double best_d = distance(feature, V.m_nodes[final_id].descriptor);
        for (auto nit = nodes.begin() + 1; nit != nodes.end(); ++nit) {
            unsigned id = *nit;
            double d = distance(feature, V.m_nodes[id].descriptor);
            if (d < best_d) { best_d = d; final_id = id; }
        }
        if (nid != nullptr && current_level == nid_level) *nid = final_id;
    } while (!V.m_nodes[final_id].isLeaf());
    word_id = V.m_nodes[final_id].word_id;
    weight = V.m_nodes[final_id].weight;
}
}  // namespace

extern "C" {
// Returns an opaque vocabulary or nullptr (short / inconsistent file).
void* obw_load_binary(const uint8_t* bytes, size_t n) {
    if (n < 24) return nullptr;
    unsigned nb_nodes, size_node;
    Vocabulary* V = new Vocabulary();
    memcpy(&nb_nodes, bytes, 4); memcpy(&size_node, bytes + 4, 4);
    memcpy(&V->m_k, bytes + 8, 4); memcpy(&V->m_L, bytes + 12, 4); memcpy(&V->m_scoring, bytes + 16, 4); memcpy(&V->m_weighting, bytes + 20, 4);
    if (size_node < 41 || n < 24 + (size_t)nb_nodes * size_node) { delete V; return nullptr; }
    V->m_nodes.resize((size_t)nb_nodes + 1);
    V->m_nodes[0].id = 0;
    const uint8_t* buf = bytes + 24;
    for (unsigned nid = 1; nid <= nb_nodes; nid++, buf += size_node) {
        Node& nd = V->m_nodes[nid];
        nd.id = nid;
        int parent;
        memcpy(&parent, buf, 4);
        if (parent < 0 || (unsigned)parent > nb_nodes || (unsigned)parent == nid) { delete V; return nullptr; }
        nd.parent = (unsigned)parent;
        V->m_nodes[nd.parent].children.push_back(nid);
        memcpy(nd.descriptor, buf + 4, 32);
        float w;
        memcpy(&w, buf + 4 + 32, 4);
        nd.weight = w;
        if (buf[8 + 32]) {   // is leaf
            nd.word_id = (unsigned)V->m_words.size();
            V->m_words.push_back(nid);
        }
    }
    return V;
}
void obw_destroy(void* v) { delete (Vocabulary*)v; }
void obw_info(void* v, int* out6) {
    const Vocabulary* V = (const Vocabulary*)v;
    out6[0] = V->m_k; out6[1] = V->m_L; out6[2] = V->m_scoring; out6[3] = V->m_weighting; out6[4] = (int)V->m_nodes.size() - 1; out6[5] = (int)V->m_words.size();
}

// transform(features, v, fv, levelsup): per-feature word / node / weight, FeatureVector as CSR (node ids ascending = std::map order,
// feature indices in insertion order), BowVector as (word ascending, value).  Returns the BowVector size.
int obw_transform(void* voc, const uint8_t* desc, int n, int levelsup, int32_t* word_id, int32_t* node_id, double* weight, int32_t* fv_node_id,
                  int32_t* fv_node_start, int32_t* fv_feat_idx, int32_t* fv_n_nodes, int32_t* bv_word, double* bv_value) {
    const Vocabulary& V = *(const Vocabulary*)voc;
    std::map<unsigned, double> v;                      // BowVector
    std::map<unsigned, std::vector<unsigned>> fv;      // FeatureVector
    *fv_n_nodes = 0;
    if (V.m_nodes.size() <= 1) return 0;   // empty()
    LNorm norm;
    const bool must = V.mustNormalize(norm);
    if (V.m_weighting == TF || V.m_weighting == TF_IDF) {
        for (unsigned i_feature = 0; i_feature < (unsigned)n; ++i_feature) {
            unsigned id, nid = 0;
            double w;
            transformOne(V, desc + (size_t)i_feature * 32, id, w, &nid, levelsup);
            word_id[i_feature] = (int32_t)id; node_id[i_feature] = (int32_t)nid; weight[i_feature] = w;
            if (w > 0) {
                auto vit = v.lower_bound(id);                                    // BowVector::addWeight
                if (vit != v.end() && !(v.key_comp()(id, vit->first))) vit->second += w;
                else v.insert(vit, std::make_pair(id, w));
                fv[nid].push_back(i_feature);                                    // FeatureVector::addFeature
            }
        }
        if (!v.empty() && !must) {
            const double nd = (double)v.size();
            for (auto vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
        }
    } else {
        for (unsigned i_feature = 0; i_feature < (unsigned)n; ++i_feature) {
            unsigned id, nid = 0;
            double w;
            transformOne(V, desc + (size_t)i_feature * 32, id, w, &nid, levelsup);
            word_id[i_feature] = (int32_t)id; node_id[i_feature] = (int32_t)nid; weight[i_feature] = w;
            if (w > 0) {
                auto vit = v.lower_bound(id);                                    // BowVector::addIfNotExist
                if (vit == v.end() || v.key_comp()(id, vit->first)) v.insert(vit, std::make_pair(id, w));
                fv[nid].push_back(i_feature);
            }
        }
    }
    if (must) {                                                                  // BowVector::normalize
        double nrm = 0.0;
        if (norm == L1) { for (auto it = v.begin(); it != v.end(); ++it) nrm += std::fabs(it->second); }
        else { for (auto it = v.begin(); it != v.end(); ++it) nrm += it->second * it->second; nrm = std::sqrt(nrm); }
        if (nrm > 0.0) for (auto it = v.begin(); it != v.end(); ++it) it->second /= nrm;
    }
    int k = 0, pos = 0;
    for (auto& kv : fv) {
        fv_node_id[k] = (int32_t)kv.first; fv_node_start[k] = pos;
        for (unsigned f : kv.second) fv_feat_idx[pos++] = (int32_t)f;
        k++;
    }
    fv_node_start[k] = pos;
    *fv_n_nodes = k;
    int m = 0;
    for (auto& kv : v) { bv_word[m] = (int32_t)kv.first; bv_value[m] = kv.second; m++; }
    return m;
}

// L1Scoring::score (ScoringObject.cpp:23-68), used by tests to sanity-check the BowVectors of two views of one scene.
double obw_score_l1(const int32_t* w1, const double* v1, int n1, const int32_t* w2, const double* v2, int n2) {
    double score = 0;
    int i = 0, j = 0;
    while (i < n1 && j < n2) {
        if (w1[i] == w2[j]) { score += std::fabs(v1[i] - v2[j]) - std::fabs(v1[i]) - std::fabs(v2[j]); i++; j++; }
        else if (w1[i] < w2[j]) i++;
        else j++;
    }
    return -score / 2.0;
}
}  // extern "C"
