"""Host-side mirror of the DBoW2 step of the hot path (include/orbhip.h "SURVEY.md N2"): ORBVocabulary in the binary format
System.cc:83 loads, and Frame::ComputeBoW == vocabulary.transform(descriptors, BowVector, FeatureVector, levelsup=4)
(reference src/Frame.cc:865-872, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1137-1272, 1442-1480)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import OrbHipError
from .matcher import _like, _ptr, _stream

L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = range(6)   # DBoW2::ScoringType  (BowVector.h:45-53)
TF_IDF, TF, IDF, BINARY = range(4)                                        # DBoW2::WeightingType (BowVector.h:36-42)


class BowResult(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("word_id", "node_id", "weight", "fv_node_id", "fv_node_start", "fv_feat_idx", "fv_n_nodes", "bv_word",
                                          "bv_value", "bv_n")]


def bind(lib):
    lib.bow_vocab_load_binary.restype = C.c_int
    lib.bow_vocab_load_binary.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
    lib.bow_vocab_info.restype = C.c_int
    lib.bow_vocab_info.argtypes = [C.c_void_p, C.c_void_p]
    lib.bow_vocab_destroy.restype = None
    lib.bow_vocab_destroy.argtypes = [C.c_void_p]
    lib.bow_transform.restype = C.c_int
    lib.bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(BowResult), C.c_void_p]
    return lib


def write_binary_vocabulary(parent, desc, weight, is_leaf, k, L, scoring=L1_NORM, weighting=TF_IDF):
    """The file image TemplatedVocabulary::saveToBinaryFile / loadFromBinaryFile exchange: 24-byte header + 41-byte node records."""
    nb = len(parent)
    rec = np.zeros(nb, np.dtype([("parent", "<i4"), ("desc", "u1", (32,)), ("weight", "<f4"), ("leaf", "u1")]))
    assert rec.dtype.itemsize == 41
    rec["parent"], rec["desc"], rec["weight"], rec["leaf"] = parent, desc, weight, is_leaf
    hdr = np.array([nb, 41], "<u4").tobytes() + np.array([k, L, scoring, weighting], "<i4").tobytes()
    return hdr + rec.tobytes()


def synth_vocabulary(seed=0, k=10, L=3, scoring=L1_NORM, weighting=TF_IDF, stop_frac=0.02, sample_desc=None):
    """A k-ary, L-level tree in creation (breadth-first) order.  Node descriptors: hierarchical perturbations of the parent (or of real
    descriptors when `sample_desc` is given), so that descriptors of one scene point usually reach the same word.  Leaf weights:
    idf-like positive floats, a few exactly 0 ("stopped" words)."""
    rng = np.random.default_rng(seed)
    parent, desc, depth = [], [], []
    level_nodes = [0]
    root_desc = {0: rng.integers(0, 256, 32, dtype=np.uint8)}
    pool = None if sample_desc is None else np.ascontiguousarray(sample_desc)
    for lev in range(1, L + 1):
        nxt = []
        for p in level_nodes:
            for _ in range(k if lev < L or True else k):
                nid = len(parent) + 1
                if pool is not None and lev == 1:
                    d = pool[rng.integers(0, len(pool))].copy()
                else:
                    d = root_desc[p].copy()
                flips = rng.integers(0, 256, max(2, 48 >> lev))
                for bpos in flips:
                    d[bpos >> 3] ^= np.uint8(1 << (bpos & 7))
                parent.append(p); desc.append(d); depth.append(lev)
                root_desc[nid] = d
                nxt.append(nid)
        level_nodes = nxt
    parent = np.array(parent, np.int32); desc = np.stack(desc); depth = np.array(depth)
    is_leaf = (depth == L).astype(np.uint8)
    weight = np.zeros(len(parent), np.float32)
    nl = int(is_leaf.sum())
    w = rng.uniform(0.5, 9.0, nl).astype(np.float32)
    w[rng.random(nl) < stop_frac] = 0.0
    weight[is_leaf == 1] = w
    return write_binary_vocabulary(parent, desc, weight, is_leaf, k, L, scoring, weighting)


def synth_vocabulary_fast(seed=0, k=10, L=6, scoring=L1_NORM, weighting=TF_IDF, stop_frac=0.001, sample_desc=None):
    """Vectorised variant of synth_vocabulary for big trees (k=10, L=6 is the shape of the stock ORBvoc: 1 111 110 nodes).
    Breadth-first node order, children of one parent contiguous."""
    rng = np.random.default_rng(seed)
    if sample_desc is not None:
        pool = np.ascontiguousarray(sample_desc)
        cur = pool[rng.integers(0, len(pool), k)].copy()
    else:
        cur = rng.integers(0, 256, (k, 32), dtype=np.uint8)
    cur_ids = np.arange(1, k + 1, dtype=np.int64)
    parents, descs, depths = [np.zeros(k, np.int32)], [cur], [np.full(k, 1, np.int8)]
    next_id = k + 1
    for lev in range(2, L + 1):
        nflip = max(2, 48 >> lev)
        child = np.repeat(cur, k, axis=0)
        bits = np.unpackbits(child, axis=1)
        idx = rng.integers(0, 256, (len(child), nflip))
        bits[np.arange(len(child))[:, None], idx] ^= 1
        child = np.packbits(bits, axis=1)
        par = np.repeat(cur_ids, k).astype(np.int32)
        parents.append(par); descs.append(child); depths.append(np.full(len(child), lev, np.int8))
        cur = child
        cur_ids = np.arange(next_id, next_id + len(child), dtype=np.int64)
        next_id += len(child)
    parent = np.concatenate(parents); desc = np.concatenate(descs); depth = np.concatenate(depths)
    is_leaf = (depth == L).astype(np.uint8)
    weight = np.zeros(len(parent), np.float32)
    nl = int(is_leaf.sum())
    w = rng.uniform(0.5, 9.0, nl).astype(np.float32)
    w[rng.random(nl) < stop_frac] = 0.0
    weight[is_leaf == 1] = w
    return write_binary_vocabulary(parent, desc, weight, is_leaf, k, L, scoring, weighting)


class ORBVocabulary:
    """ORB_SLAM3::ORBVocabulary (include/ORBVocabulary.h) resident on the device."""

    def __init__(self, file_bytes, device=0, lib=None):
        self._L = bind(lib if lib is not None else _lib.load())
        self._h = C.c_void_p()
        buf = np.frombuffer(file_bytes, np.uint8)
        rc = self._L.bow_vocab_load_binary(buf.ctypes.data_as(C.c_void_p), buf.size, device, C.byref(self._h))
        if rc != 0:
            raise OrbHipError(rc, "bow_vocab_load_binary failed")
        info = np.zeros(6, np.int32)
        self._L.bow_vocab_info(self._h, info.ctypes.data_as(C.c_void_p))
        self.k, self.L, self.scoring, self.weighting, self.n_nodes, self.n_words = [int(x) for x in info]

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.bow_vocab_destroy(self._h)
            self._h = C.c_void_p()

    def transform(self, desc, n, levelsup=4):
        """desc [B,cap,32] u8, n [B] int32 (device tensors, or numpy with the emulated build) -> dict of the bow_result slabs.
        The fv_* arrays have exactly the layout ORBmatcher.SearchByBoW / SearchForTriangulation take (with cap_nodes = cap)."""
        B, cap = desc.shape[0], desc.shape[1]
        o = dict(word_id=_like(desc, (B, cap), np.int32), node_id=_like(desc, (B, cap), np.int32), weight=_like64f(desc, (B, cap)),
                 fv_node_id=_like(desc, (B, cap), np.int32), fv_node_start=_like(desc, (B, cap + 1), np.int32),
                 fv_feat_idx=_like(desc, (B, cap), np.int32), fv_n_nodes=_like(desc, (B,), np.int32),
                 bv_word=_like(desc, (B, cap), np.int32), bv_value=_like64f(desc, (B, cap)), bv_n=_like(desc, (B,), np.int32))
        R = BowResult(*[_ptr(o[k]).value for k in ("word_id", "node_id", "weight", "fv_node_id", "fv_node_start", "fv_feat_idx", "fv_n_nodes",
                                                   "bv_word", "bv_value", "bv_n")])
        rc = self._L.bow_transform(self._h, _ptr(desc), _ptr(n), 1, cap, B, int(levelsup), C.byref(R), _stream(desc))
        if rc != 0:
            raise OrbHipError(rc, "bow_transform failed")
        return o


def _like64f(a, shape):
    if isinstance(a, np.ndarray):
        return np.zeros(shape, np.float64)
    import torch
    return torch.zeros(shape, dtype=torch.float64, device=a.device)
