"""SURVEY N2 parity: DBoW2 vocabulary (binary format) + Frame::ComputeBoW on the device vs the oracle's restatement
(oracle/bow_oracle.cpp).  Word / node ids, the FeatureVector CSR and the BowVector doubles must be bit-identical; the device CSR then
drives SearchByBoW end to end."""
import numpy as np
import pytest

import oracle_lib as O
import orbhip
from orbhip.bow import (BINARY, DOT_PRODUCT, IDF, L1_NORM, L2_NORM, TF, TF_IDF, ORBVocabulary, synth_vocabulary, write_binary_vocabulary)
from test_matcher_parity import scene, to_dev, to_host


def test_oracle_loader_and_descent_known_answers():
    """A hand-built 2-level tree (k=2): the descent takes the first minimum in child order, node ids follow file order, word ids leaf
    order, and the level-(L - levelsup) ancestor is reported."""
    z = np.zeros(32, np.uint8)
    f = np.full(32, 255, np.uint8)
    h = z.copy(); h[:16] = 255
    # nodes 1,2: children of the root; 3,4 under 1; 5,6 under 2
    parent = [0, 0, 1, 1, 2, 2]
    desc = np.stack([z, f, z, h, f, h])
    weight = [0, 0, 1.5, 2.5, 0.0, 4.0]
    leaf = [0, 0, 1, 1, 1, 1]
    V = O.OracleVocabulary(write_binary_vocabulary(parent, desc, weight, leaf, 2, 2))
    assert (V.k, V.L, V.n_nodes, V.n_words) == (2, 2, 6, 4)
    q = np.stack([z, h, f, f])
    q[1, 0] = 0x7F   # one bit towards zero: level 1 ties are impossible here (129 vs 127 bits) -> child 1 (all zero) loses
    o = V.transform(q, levelsup=1)
    # q0 -> node 1 -> node 3 (word 0); q1: dist to z = 127, to f = 129 -> node 1; then z:127 vs h:1 -> node 4 (word 1)
    # q2, q3 -> node 2 -> node 5 (word 2, weight 0: stopped)
    assert o["word_id"][:4].tolist() == [0, 1, 2, 2] and o["node_id"][:4].tolist() == [1, 1, 2, 2]
    assert o["weight"][:4].tolist() == [1.5, 2.5, 0.0, 0.0]
    assert o["fv_n_nodes"] == 1 and o["fv_node_id"][0] == 1 and o["fv_feat_idx"][:2].tolist() == [0, 1]
    assert o["bv_n"] == 2 and o["bv_word"][:2].tolist() == [0, 1]
    assert o["bv_value"][:2].tolist() == [1.5 / 4.0, 2.5 / 4.0]   # L1-normalised
    # exact tie at level 1 (128 vs 128 bits): strict '<' keeps the first child
    t = z.copy(); t[:16] = 255
    assert V.transform(t[None], levelsup=1)["node_id"][0] == 1
    with pytest.raises(ValueError):
        O.OracleVocabulary(write_binary_vocabulary(parent, desc, weight, leaf, 2, 2)[:-5])   # truncated file


def _frames(B, cap):
    S = scene()
    ds = [S["da"], S["db"], S["da"][::-1].copy(), S["db"][:37]][:B]
    desc = np.zeros((B, cap, 32), np.uint8)
    n = np.zeros(B, np.int32)
    for b, d in enumerate(ds):
        desc[b, :len(d)] = d; n[b] = len(d)
    return S, ds, desc, n


def _check(lib, backend, scoring, weighting, k, L, levelsup, seed=0):
    S = scene()
    blob = synth_vocabulary(seed, k, L, scoring, weighting, stop_frac=0.05, sample_desc=np.concatenate([S["da"], S["db"]]))
    ov = O.OracleVocabulary(blob)
    B, cap = 4, max(len(S["da"]), len(S["db"])) + 11
    _, ds, desc, n = _frames(B, cap)
    V = ORBVocabulary(blob, lib=lib)
    assert (V.k, V.L, V.scoring, V.weighting, V.n_nodes, V.n_words) == (ov.k, ov.L, ov.scoring, ov.weighting, ov.n_nodes, ov.n_words)
    r = {kk: to_host(v) for kk, v in V.transform(to_dev(desc, backend), to_dev(n, backend), levelsup).items()}
    stopped = 0
    for b, d in enumerate(ds):
        o = ov.transform(d, levelsup)
        m = len(d)
        assert np.array_equal(r["word_id"][b, :m], o["word_id"][:m]) and np.array_equal(r["node_id"][b, :m], o["node_id"][:m])
        assert np.array_equal(r["weight"][b, :m], o["weight"][:m])
        nn = o["fv_n_nodes"]
        assert r["fv_n_nodes"][b] == nn
        assert np.array_equal(r["fv_node_id"][b, :nn], o["fv_node_id"][:nn]) and np.array_equal(r["fv_node_start"][b, :nn + 1], o["fv_node_start"][:nn + 1])
        nf = o["fv_node_start"][nn]
        assert np.array_equal(r["fv_feat_idx"][b, :nf], o["fv_feat_idx"][:nf])
        nb = o["bv_n"]
        assert r["bv_n"][b] == nb and np.array_equal(r["bv_word"][b, :nb], o["bv_word"][:nb])
        assert np.array_equal(r["bv_value"][b, :nb].view(np.uint64), o["bv_value"][:nb].view(np.uint64)), "BowVector doubles must be bit-identical"
        assert nb > 20
        stopped += int((o["weight"][:m] == 0).sum())
    assert stopped > 0   # the vocabulary really has stopped words that the scene hits
    return S, ov, r


CASES = [(L1_NORM, TF_IDF, 10, 3, 2), (L2_NORM, TF, 6, 4, 3), (DOT_PRODUCT, TF_IDF, 10, 3, 4), (L1_NORM, BINARY, 17, 2, 1), (DOT_PRODUCT, IDF, 8, 3, 1)]


@pytest.mark.parametrize("scoring,weighting,k,L,levelsup", CASES[:4])
def test_emu_bow_transform_matches_oracle(emu_lib, scoring, weighting, k, L, levelsup):
    _check(emu_lib, "emu", scoring, weighting, k, L, levelsup)


@pytest.mark.gpu
@pytest.mark.parametrize("scoring,weighting,k,L,levelsup", CASES)
def test_hip_bow_transform_matches_oracle(hip_lib, scoring, weighting, k, L, levelsup):
    _check(hip_lib, "hip", scoring, weighting, k, L, levelsup)


def _bow_to_search(lib, backend):
    """extract (oracle keypoints) -> ComputeBoW on the device -> SearchByBoW on the device CSR == the oracle chain."""
    S = scene()
    blob = synth_vocabulary(3, 10, 3, sample_desc=np.concatenate([S["da"], S["db"]]))
    ov = O.OracleVocabulary(blob)
    V = ORBVocabulary(blob, lib=lib)
    ka, da, kb, db = S["ka"], S["da"], S["kb"], S["db"]
    cap = max(len(ka), len(kb)) + 5
    oa, ob = ov.transform(da, 1), ov.transform(db, 1)   # levelsup=1 on a 3-level tree: level-2 nodes, ~10 features per node
    side = lambda o, k, d: dict(desc=d, angle=np.ascontiguousarray(k["angle"]), node_id=o["fv_node_id"][:o["fv_n_nodes"]],
                                node_start=o["fv_node_start"][:o["fv_n_nodes"] + 1], feat_idx=o["fv_feat_idx"][:o["fv_node_start"][o["fv_n_nodes"]]],
                                n_nodes=o["fv_n_nodes"])
    kvalid = np.ones(len(ka), np.uint8)
    om, on = O.search_by_bow(side(oa, ka, da), kvalid, side(ob, kb, db), 0.7, True)
    d = lambda a: to_dev(a, backend)
    slab = lambda a, dt=None: np.concatenate([a, np.zeros((cap - len(a),) + a.shape[1:], a.dtype)])[None]
    ra = V.transform(d(slab(da)), d(np.array([len(ka)], np.int32)), 1)
    rb = V.transform(d(slab(db)), d(np.array([len(kb)], np.int32)), 1)
    m = orbhip.ORBmatcher(0.7, True, lib=lib)
    mk = lambda r, k, dsc: dict(desc=d(slab(dsc)), angle=d(slab(np.ascontiguousarray(k["angle"]))), node_id=r["fv_node_id"], node_start=r["fv_node_start"],
                                feat_idx=r["fv_feat_idx"], n_nodes=r["fv_n_nodes"])
    fm, nm = [to_host(x) for x in m.SearchByBoW(mk(ra, ka, da), d(slab(kvalid)), mk(rb, kb, db))]
    assert nm[0] == on and np.array_equal(fm[0, :len(kb)], om) and on > 30
    # two views of one scene score far higher than unrelated descriptor sets (L1 score in [0, 1])
    s_same = O.bow_score_l1(oa["bv_word"][:oa["bv_n"]], oa["bv_value"][:oa["bv_n"]], ob["bv_word"][:ob["bv_n"]], ob["bv_value"][:ob["bv_n"]])
    rnd = ov.transform(np.random.default_rng(1).integers(0, 256, (len(da), 32), dtype=np.uint8), 1)
    s_rand = O.bow_score_l1(oa["bv_word"][:oa["bv_n"]], oa["bv_value"][:oa["bv_n"]], rnd["bv_word"][:rnd["bv_n"]], rnd["bv_value"][:rnd["bv_n"]])
    assert 0 <= s_rand < s_same <= 1 and s_same > 1.5 * s_rand


def test_emu_compute_bow_feeds_search_by_bow(emu_lib):
    _bow_to_search(emu_lib, "emu")


@pytest.mark.gpu
def test_hip_compute_bow_feeds_search_by_bow(hip_lib):
    _bow_to_search(hip_lib, "hip")


def test_loader_rejects_inconsistent_files(emu_lib):
    S = scene()
    blob = bytearray(synth_vocabulary(0, 4, 2))
    with pytest.raises(orbhip._lib.OrbHipError):
        ORBVocabulary(bytes(blob[:100]), lib=emu_lib)          # truncated
    bad = bytearray(blob)
    bad[24 + 40] = 1                                           # node 1 (an inner node) flagged as a leaf
    with pytest.raises(orbhip._lib.OrbHipError):
        ORBVocabulary(bytes(bad), lib=emu_lib)
    bad = bytearray(blob)
    bad[24:28] = np.array([99999], "<i4").tobytes()            # parent out of range
    with pytest.raises(orbhip._lib.OrbHipError):
        ORBVocabulary(bytes(bad), lib=emu_lib)
