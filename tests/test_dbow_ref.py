"""The one part of row N2 that is pinned to the REFERENCE compiled here (round-4 verdict, item 7): oracle/_ref/libdbow_ref.so =
/root/reference/Thirdparty/DBoW2/DBoW2/{BowVector,FeatureVector}.cpp compiled as they lie (`make -C oracle _ref`, shim oracle/dbow_ref_shim.cc).
oracle/bow_oracle.cpp restates BowVector::addWeight / addIfNotExist / normalize (BowVector.cpp:34-84) and FeatureVector::addFeature
(FeatureVector.cpp:28-41) inside its transform(); here the per-feature (word, weight, node) sequences the oracle's descent produces are replayed
through the reference's own classes and the resulting doubles / index lists must equal the oracle's outputs bit for bit.  What this does NOT pin:
the descent (FORB::distance, the tree walk of TemplatedVocabulary.h) and the size-division of the not-normalised scorings — both live in headers
that need OpenCV."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from orbhip.bow import BINARY, DOT_PRODUCT, IDF, L1_NORM, L2_NORM, TF, TF_IDF, synth_vocabulary
from test_matcher_parity import scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libdbow_ref.so")


def _ref():
    if os.path.isdir("/root/reference/Thirdparty/DBoW2/DBoW2"):
        subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref"], stdout=subprocess.DEVNULL)   # (a failed build leaves the skip below)
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libdbow_ref.so absent and /root/reference not here to build it from")
    L = C.CDLL(REF_SO)
    L.dbr_bowvector.restype = C.c_int
    L.dbr_bowvector.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.dbr_featurevector.restype = C.c_int
    L.dbr_featurevector.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _bow(L, ids, vals, mode, norm):
    ids = np.ascontiguousarray(ids, np.uint32); vals = np.ascontiguousarray(vals, np.float64)
    oi = np.zeros(len(ids) + 1, np.uint32); ov = np.zeros(len(ids) + 1)
    n = L.dbr_bowvector(_p(ids), _p(vals), len(ids), mode, norm, _p(oi), _p(ov), len(oi))
    assert n >= 0
    return oi[:n], ov[:n]


def test_reference_bowvector_known_answers():
    """the compiled reference itself on hand-made sequences (so a broken build of it cannot hide behind an equally broken restatement)"""
    L = _ref()
    i, v = _bow(L, [5, 2, 5, 9, 2], [1.0, 2.0, 0.5, 4.0, 0.25], 0, -1)              # addWeight accumulates, std::map order
    assert i.tolist() == [2, 5, 9] and v.tolist() == [2.25, 1.5, 4.0]
    i, v = _bow(L, [5, 2, 5, 9, 2], [1.0, 2.0, 0.5, 4.0, 0.25], 1, -1)              # addIfNotExist keeps the first
    assert i.tolist() == [2, 5, 9] and v.tolist() == [2.0, 1.0, 4.0]
    i, v = _bow(L, [1, 2], [3.0, -1.0], 0, 0)                                          # L1: sum of |v|
    assert v.tolist() == [0.75, -0.25]
    i, v = _bow(L, [1, 2], [3.0, 4.0], 0, 1)                                           # L2
    assert v.tolist() == [3.0 / 5.0, 4.0 / 5.0]
    i, v = _bow(L, [1, 2], [0.0, 0.0], 0, 0)                                           # norm 0: left alone (BowVector.cpp:79)
    assert v.tolist() == [0.0, 0.0]
    nodes = np.array([7, 3, 7, 3, 1], np.uint32); feat = np.array([0, 1, 2, 3, 4], np.uint32)
    on = np.zeros(6, np.uint32); os_ = np.zeros(7, np.int32); of = np.zeros(5, np.uint32)
    k = L.dbr_featurevector(_p(nodes), _p(feat), 5, _p(on), _p(os_), _p(of), 6)
    assert k == 3 and on[:3].tolist() == [1, 3, 7] and os_[:4].tolist() == [0, 1, 3, 5] and of.tolist() == [4, 1, 3, 0, 2]


@pytest.mark.parametrize("scoring,weighting", [(L1_NORM, TF_IDF), (L2_NORM, TF_IDF), (L1_NORM, TF), (L1_NORM, IDF), (L2_NORM, BINARY), (DOT_PRODUCT, TF_IDF)])
def test_oracle_accumulation_equals_compiled_reference(scoring, weighting):
    """ORBvoc's own combination (TF_IDF + L1_NORM, TemplatedVocabulary.h:1442-1480 header of the stock file) and the other weighting / norm branches:
    the oracle's per-feature sequence through the reference's BowVector / FeatureVector == the oracle's own BowVector / FeatureVector, bitwise."""
    L = _ref()
    S = scene()
    for seed, desc in ((0, S["da"]), (1, S["db"]), (2, np.concatenate([S["da"], S["db"], S["da"][:50]]))):
        blob = synth_vocabulary(seed, 6, 3, scoring, weighting, stop_frac=0.05, sample_desc=np.concatenate([S["da"], S["db"]]))
        ov = O.OracleVocabulary(blob)
        o = ov.transform(desc, levelsup=2)
        n = len(desc)
        keep = o["weight"][:n] > 0                                                      # TemplatedVocabulary.h:1167 `if(w > 0)`
        ids, w, nid = o["word_id"][:n][keep], o["weight"][:n][keep], o["node_id"][:n][keep]
        feat = np.nonzero(keep)[0].astype(np.uint32)
        assert keep.sum() > 50 and (~keep).sum() > 0                                    # stop words occur
        mode = 0 if weighting in (TF, TF_IDF) else 1                                    # TemplatedVocabulary.h:1160 / :1189
        norm = {L1_NORM: 0, L2_NORM: 1}.get(scoring, -1)                                # mustNormalize (ScoringObject.h)
        ri, rv = _bow(L, ids, w, mode, norm)
        m = o["bv_n"]
        if norm < 0 and mode == 0:                                                      # the division by v.size() of TemplatedVocabulary.h:1177-1184 is not in BowVector.cpp
            rv = rv / float(len(rv))
        assert m == len(ri) and np.array_equal(o["bv_word"][:m].astype(np.uint32), ri)
        assert np.array_equal(o["bv_value"][:m].view(np.uint64), rv.view(np.uint64)), "BowVector doubles differ from the compiled reference"
        nn = np.ascontiguousarray(nid, np.uint32)
        on = np.zeros(len(nn) + 1, np.uint32); os_ = np.zeros(len(nn) + 2, np.int32); of = np.zeros(len(nn) + 1, np.uint32)
        k = L.dbr_featurevector(_p(nn), _p(feat), len(nn), _p(on), _p(os_), _p(of), len(on))
        assert k == o["fv_n_nodes"] and np.array_equal(on[:k], o["fv_node_id"][:k].astype(np.uint32))
        assert np.array_equal(os_[:k + 1], o["fv_node_start"][:k + 1]) and np.array_equal(of[:len(nn)], o["fv_feat_idx"][:len(nn)].astype(np.uint32))
