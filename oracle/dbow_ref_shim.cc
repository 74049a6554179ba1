// oracle/dbow_ref_shim.cc — C entry points over the REFERENCE's own DBoW2::BowVector / DBoW2::FeatureVector, compiled from
// /root/reference/Thirdparty/DBoW2/DBoW2/{BowVector,FeatureVector}.cpp where they lie (recipe: `make -C oracle _ref`, output oracle/_ref/libdbow_ref.so;
// nothing of the reference is copied into this repository, no stand-in header is written: the two files need only the C++ standard library).
// TEST INFRASTRUCTURE: tests/test_dbow_ref.py drives these with the word / weight / node sequences oracle/bow_oracle.cpp produces and compares the
// doubles bit for bit — the one part of row N2 that is pinned to the reference compiled here.  (.cc, not .cpp: oracle/Makefile's
// liboracle.so rule takes *.cpp, and this file needs the reference's header on the include path.)
#include <cstdint>

#include "BowVector.h"        // -I /root/reference/Thirdparty/DBoW2/DBoW2
#include "FeatureVector.h"

extern "C" {
// n operations (ids[i], vals[i]) applied in order with addWeight (mode 0) or addIfNotExist (mode 1), then normalize(L1) (norm 0), normalize(L2) (norm 1)
// or nothing (norm -1) -> the map's (word, value) pairs in iteration order; returns the map size (or -1 if it exceeds cap)
int dbr_bowvector(const uint32_t* ids, const double* vals, int n, int mode, int norm, uint32_t* out_ids, double* out_vals, int cap) {
    DBoW2::BowVector v;
    for (int i = 0; i < n; i++) {
        if (mode == 0) v.addWeight(ids[i], vals[i]);
        else v.addIfNotExist(ids[i], vals[i]);
    }
    if (norm == 0) v.normalize(DBoW2::L1);
    else if (norm == 1) v.normalize(DBoW2::L2);
    if ((int)v.size() > cap) return -1;
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) { out_ids[k] = it->first; out_vals[k] = it->second; }
    return k;
}
// n calls addFeature(node_ids[i], feat[i]) -> CSR in the map's iteration order; returns the number of nodes (or -1 if it exceeds cap_nodes)
int dbr_featurevector(const uint32_t* node_ids, const uint32_t* feat, int n, uint32_t* out_nodes, int32_t* out_start, uint32_t* out_feat, int cap_nodes) {
    DBoW2::FeatureVector fv;
    for (int i = 0; i < n; i++) fv.addFeature(node_ids[i], feat[i]);
    if ((int)fv.size() > cap_nodes) return -1;
    int k = 0, pos = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++k) {
        out_nodes[k] = it->first; out_start[k] = pos;
        for (unsigned f : it->second) out_feat[pos++] = f;
    }
    out_start[k] = pos;
    return k;
}
}
