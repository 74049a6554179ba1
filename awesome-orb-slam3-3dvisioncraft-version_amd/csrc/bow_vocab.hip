// bow_vocab.hip — SURVEY.md N2 on gfx950: DBoW2 vocabulary (binary format) + Frame::ComputeBoW
//   (reference src/Frame.cc:865-872 -> Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1137-1206, 1231-1272, 1442-1480; FORB.cpp:81-101;
//    BowVector.cpp:34-84).
//   k_bow_descend   one 16-lane group per descriptor walks the tree: lane j takes child j of the current node (256-bit Hamming against
//                   the children's descriptors, stored contiguously per parent), first-minimum over the group, repeat until a leaf
//   k_bow_build     one workgroup per frame turns the per-feature (node, word, weight) into the FeatureVector CSR and the BowVector:
//                   two LDS bitonic sorts of (key << 32 | feature index); the BowVector's L1/L2 norm is accumulated serially in
//                   ascending word order, exactly like std::map iteration, so the normalised doubles are bit-identical
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/orbhip.h"

struct bow_vocab {
    int device = 0, k = 0, L = 0, scoring = 0, weighting = 0, nb = 0, nwords = 0;
    int32_t* d_child_start = nullptr;   // [nb + 2] CSR over parents (node ids 0..nb)
    int32_t* d_child_id = nullptr;      // [nb]
    uint8_t* d_child_desc = nullptr;    // [nb][32] in CSR order
    double* d_weight = nullptr;         // [nb + 1]
    int32_t* d_word = nullptr;          // [nb + 1]
};

extern "C" void bow_vocab_destroy(bow_vocab_handle v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    void* bufs[] = {v->d_child_start, v->d_child_id, v->d_child_desc, v->d_weight, v->d_word};
    for (void* p : bufs) if (p) (void)hipFree(p);
    delete v;
}

extern "C" int bow_vocab_load_binary(const void* file_bytes, size_t n_bytes, int device, bow_vocab_handle* out) {
    if (!file_bytes || !out) return ORB_E_INVALID;
    *out = nullptr;
    const uint8_t* bytes = (const uint8_t*)file_bytes;
    if (n_bytes < 24) return ORB_E_INVALID;
    uint32_t nb, size_node;
    int32_t hdr[4];
    memcpy(&nb, bytes, 4); memcpy(&size_node, bytes + 4, 4); memcpy(hdr, bytes + 8, 16);
    if (size_node < 41 || nb < 1 || nb > 0x7FFFFFF0u / 41u || n_bytes < 24 + (size_t)nb * size_node) return ORB_E_INVALID;
    // same traversal as loadFromBinaryFile: node ids in file order, children appended in file order, word ids in leaf order
    std::vector<int32_t> parent(nb + 1, 0), word(nb + 1, -1), cnt(nb + 2, 0);
    std::vector<double> weight(nb + 1, 0.0);
    std::vector<uint8_t> leaf(nb + 1, 0);
    int nwords = 0;
    const uint8_t* buf = bytes + 24;
    for (uint32_t nid = 1; nid <= nb; nid++, buf += size_node) {
        int32_t p;
        memcpy(&p, buf, 4);
        if (p < 0 || (uint32_t)p > nb || (uint32_t)p == nid) return ORB_E_INVALID;
        parent[nid] = p;
        cnt[p + 1]++;
        float w;
        memcpy(&w, buf + 36, 4);
        weight[nid] = (double)w;   // Node::weight is a double (WordValue) assigned from the file's float
        leaf[nid] = buf[40] ? 1 : 0;
        if (leaf[nid]) word[nid] = nwords++;
    }
    for (uint32_t i = 1; i <= nb + 1; i++) cnt[i] += cnt[i - 1];   // cnt[p] = start of p's children
    std::vector<int32_t> child_start(cnt), fill(cnt.begin(), cnt.end() - 1), child_id(nb);
    std::vector<uint8_t> child_desc((size_t)nb * 32);
    buf = bytes + 24;
    for (uint32_t nid = 1; nid <= nb; nid++, buf += size_node) {
        const int pos = fill[parent[nid]]++;
        child_id[pos] = (int32_t)nid;
        memcpy(&child_desc[(size_t)pos * 32], buf + 4, 32);
    }
    // a vocabulary the descent can walk: the root has children, leaves (by flag) have none, inner nodes have some
    if (child_start[1] - child_start[0] < 1) return ORB_E_INVALID;
    for (uint32_t nid = 1; nid <= nb; nid++) {
        const int nc = child_start[nid + 1] - child_start[nid];
        if ((leaf[nid] && nc != 0) || (!leaf[nid] && nc == 0)) return ORB_E_INVALID;
    }
    bow_vocab* v = new (std::nothrow) bow_vocab();
    if (!v) return ORB_E_NOMEM;
    v->device = device; v->k = hdr[0]; v->L = hdr[1]; v->scoring = hdr[2]; v->weighting = hdr[3]; v->nb = (int)nb; v->nwords = nwords;
    if (hipSetDevice(device) != hipSuccess) { delete v; return ORB_E_HIP; }
#define BK(call) do { if ((call) != hipSuccess) { bow_vocab_destroy(v); return ORB_E_HIP; } } while (0)
    BK(hipMalloc((void**)&v->d_child_start, (size_t)(nb + 2) * 4));
    BK(hipMalloc((void**)&v->d_child_id, (size_t)nb * 4));
    BK(hipMalloc((void**)&v->d_child_desc, (size_t)nb * 32));
    BK(hipMalloc((void**)&v->d_weight, (size_t)(nb + 1) * 8));
    BK(hipMalloc((void**)&v->d_word, (size_t)(nb + 1) * 4));
    BK(hipMemcpy(v->d_child_start, child_start.data(), (size_t)(nb + 2) * 4, hipMemcpyHostToDevice));
    BK(hipMemcpy(v->d_child_id, child_id.data(), (size_t)nb * 4, hipMemcpyHostToDevice));
    BK(hipMemcpy(v->d_child_desc, child_desc.data(), (size_t)nb * 32, hipMemcpyHostToDevice));
    BK(hipMemcpy(v->d_weight, weight.data(), (size_t)(nb + 1) * 8, hipMemcpyHostToDevice));
    BK(hipMemcpy(v->d_word, word.data(), (size_t)(nb + 1) * 4, hipMemcpyHostToDevice));
#undef BK
    *out = v;
    return ORB_OK;
}

extern "C" int bow_vocab_info(bow_vocab_handle v, int32_t* out6) {
    if (!v || !out6) return ORB_E_INVALID;
    out6[0] = v->k; out6[1] = v->L; out6[2] = v->scoring; out6[3] = v->weighting; out6[4] = v->nb; out6[5] = v->nwords;
    return ORB_OK;
}

struct BowArgsT {
    const int32_t* child_start; const int32_t* child_id; const uint8_t* child_desc; const double* weight; const int32_t* word;
    int L, scoring, weighting;
    const uint8_t* desc; const int32_t* n; int cstride, cap_f, levelsup, sortN;
    bow_result out;
};

static __device__ __forceinline__ int hamming32(const uint4& a0, const uint4& a1, const uint8_t* p) {
    const uint4 b0 = *(const uint4*)p, b1 = *(const uint4*)(p + 16);
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) +
           __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// transform(feature, word_id, weight, nid, levelsup)  TemplatedVocabulary.h:1231-1272
static __global__ __launch_bounds__(256) void k_bow_descend(BowArgsT A) {
    const int b = blockIdx.y, sub = threadIdx.x & 15;
    const int f = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int n = min(A.n[(size_t)b * A.cstride], A.cap_f);
    const bool on = f < n;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    if (on) {
        const uint8_t* p = A.desc + ((size_t)b * A.cap_f + f) * 32;
        a0 = *(const uint4*)p; a1 = *(const uint4*)(p + 16);
    }
    const int nid_level = A.L - A.levelsup;
    int node = 0, nid = 0, level = 0;
    bool active = on;
    while (__any(active)) {   // groups of one wave may finish at different depths
        int cs = 0, ce = 0;
        if (active) { cs = A.child_start[node]; ce = A.child_start[node + 1]; }
        uint32_t key = 0xFFFFFFFFu;   // dist << 20 | child position: first minimum == the reference's strict '<' scan in child order
        for (int c = cs + sub; c < ce; c += 16) {
            const uint32_t k = ((uint32_t)hamming32(a0, a1, A.child_desc + (size_t)c * 32) << 20) | (uint32_t)(c - cs);
            key = min(key, k);
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, off));
        if (active) {
            level++;
            node = A.child_id[cs + (int)(key & 0xFFFFFu)];
            if (level == nid_level) nid = node;
            active = A.child_start[node + 1] > A.child_start[node];   // !isLeaf()
        }
    }
    if (on && sub == 0) {
        const size_t o = (size_t)b * A.cap_f + f;
        A.out.word_id[o] = A.word[node];
        A.out.node_id[o] = nid;
        A.out.weight[o] = A.weight[node];
    }
}

// block-wide exclusive scan of one int per thread (256 threads); returns the total
static __device__ __forceinline__ int scan256(int v, int* scratch, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    __syncthreads();
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    const int w0 = scratch[0], w1 = scratch[1], w2 = scratch[2], w3 = scratch[3];
    total = w0 + w1 + w2 + w3;
    return (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0) + incl - v;
}

static __device__ void bitonic_sort_u64(unsigned long long* k, int N) {   // ascending, N a power of two, whole block
    for (int size = 2; size <= N; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (N >> 1); t += 256) {
                const int i = ((t / stride) * stride * 2) + (t % stride), j = i + stride;
                const bool up = ((i & size) == 0);
                const unsigned long long a = k[i], c = k[j];
                if ((a > c) == up) { k[i] = c; k[j] = a; }
            }
        }
    __syncthreads();
}

// transform(features, BowVector&, FeatureVector&, levelsup)  TemplatedVocabulary.h:1137-1206, one workgroup per frame
static __global__ __launch_bounds__(256) void k_bow_build(BowArgsT A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orb_smem[];
    unsigned long long* keys = (unsigned long long*)orb_smem;   // [sortN]; later reused as the BowVector values (doubles)
    int* scratch = (int*)(keys + A.sortN);                      // [8]
    const int b = blockIdx.x, tid = threadIdx.x, N = A.sortN;
    const int n = min(A.n[(size_t)b * A.cstride], A.cap_f);
    const size_t fo = (size_t)b * A.cap_f;
    const int32_t* word = A.out.word_id + fo; const int32_t* nodeid = A.out.node_id + fo; const double* wgt = A.out.weight + fo;
    // ---- FeatureVector: features with w > 0 sorted by (node id, feature index) — std::map order, insertion order inside a node
    for (int i = tid; i < N; i += 256) keys[i] = (i < n && wgt[i] > 0) ? (((unsigned long long)(uint32_t)nodeid[i] << 32) | (uint32_t)i) : ~0ull;
    bitonic_sort_u64(keys, N);
    int32_t* fv_node = A.out.fv_node_id + fo; int32_t* fv_feat = A.out.fv_feat_idx + fo;
    int32_t* fv_start = A.out.fv_node_start + (size_t)b * (A.cap_f + 1);
    int base = 0, m = 0;
    for (int c0 = 0; c0 < N; c0 += 256) {   // chunks of 256 sorted positions
        const int pos = c0 + tid;
        const unsigned long long kk = pos < N ? keys[pos] : ~0ull;
        const bool valid = kk != ~0ull;
        const bool head = valid && (pos == 0 || (keys[pos - 1] >> 32) != (kk >> 32));
        int tot;
        const int idx = base + scan256(head ? 1 : 0, scratch, tot);
        if (valid) fv_feat[pos] = (int32_t)(kk & 0xFFFFFFFFu);
        if (head) { fv_node[idx] = (int32_t)(kk >> 32); fv_start[idx] = pos; }
        base += tot;
        int vt;
        scan256(valid ? 1 : 0, scratch, vt);
        m += vt;
    }
    if (tid == 0) { fv_start[base] = m; A.out.fv_n_nodes[b] = base; }
    __syncthreads();
    // ---- BowVector: (word id, feature index) sorted; a run of one word = the features that hit it
    for (int i = tid; i < N; i += 256) keys[i] = (i < n && wgt[i] > 0) ? (((unsigned long long)(uint32_t)word[i] << 32) | (uint32_t)i) : ~0ull;
    bitonic_sort_u64(keys, N);
    int32_t* bv_word = A.out.bv_word + fo; double* bv_val = A.out.bv_value + fo;
    const bool tf = A.weighting == 0 || A.weighting == 1;   // TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
    base = 0;
    for (int c0 = 0; c0 < N; c0 += 256) {
        const int pos = c0 + tid;
        const unsigned long long kk = pos < N ? keys[pos] : ~0ull;
        const bool valid = kk != ~0ull;
        const bool head = valid && (pos == 0 || (keys[pos - 1] >> 32) != (kk >> 32));
        int tot;
        const int idx = base + scan256(head ? 1 : 0, scratch, tot);
        if (head) {
            const double w = wgt[(int)(kk & 0xFFFFFFFFu)];   // every feature of the run carries the word's weight
            double v = w;
            if (tf) {   // v.addWeight(id, w) once per feature, in feature order: w + w + ... (all addends equal)
                for (int q = pos + 1; q < N && (keys[q] >> 32) == (kk >> 32); q++) v += w;
            }
            bv_word[idx] = (int32_t)(kk >> 32);
            bv_val[idx] = v;
        }
        base += tot;
    }
    const int nb = base;
    __threadfence_block();
    __syncthreads();
    // normalisation (BowVector::normalize, or the `/= size` of the non-normalising scorings), map order = ascending word id
    double* vals = (double*)keys;
    for (int i = tid; i < nb; i += 256) vals[i] = bv_val[i];
    __syncthreads();
    const bool must = A.scoring != 5;        // every scoring but DOT_PRODUCT normalises (ScoringObject.h:74-89)
    const bool l2 = A.scoring == 1;          // L2Scoring -> L2, the others L1
    double* nrmp = (double*)(scratch + 4);   // 8-byte aligned (scratch follows an 8-byte array)
    if (tid == 0) {
        double nrm = 0.0;
        if (must) {
            if (!l2) { for (int i = 0; i < nb; i++) nrm += fabs(vals[i]); }
            else { for (int i = 0; i < nb; i++) nrm += vals[i] * vals[i]; nrm = sqrt(nrm); }
        } else if (tf && nb > 0) {
            nrm = (double)nb;
        }
        *nrmp = nrm;
        A.out.bv_n[b] = nb;
    }
    __syncthreads();
    const double nrm = *nrmp;
    if (nrm > 0.0)
        for (int i = tid; i < nb; i += 256) bv_val[i] = vals[i] / nrm;
}

extern "C" int bow_transform(bow_vocab_handle v, const uint8_t* d_desc, const int32_t* d_n, int count_stride, int cap_f, int batch, int levelsup,
                             const bow_result* out, void* stream) {
    if (!v || !d_desc || !d_n || !out || count_stride < 1 || cap_f < 1 || cap_f > 4096 || batch < 1) return ORB_E_INVALID;
    if (!out->word_id || !out->node_id || !out->weight || !out->fv_node_id || !out->fv_node_start || !out->fv_feat_idx || !out->fv_n_nodes ||
        !out->bv_word || !out->bv_value || !out->bv_n)
        return ORB_E_INVALID;
    int sortN = 256;
    while (sortN < cap_f) sortN <<= 1;
    BowArgsT A;
    A.child_start = v->d_child_start; A.child_id = v->d_child_id; A.child_desc = v->d_child_desc; A.weight = v->d_weight; A.word = v->d_word;
    A.L = v->L; A.scoring = v->scoring; A.weighting = v->weighting;
    A.desc = d_desc; A.n = d_n; A.cstride = count_stride; A.cap_f = cap_f; A.levelsup = levelsup; A.sortN = sortN; A.out = *out;
    hipLaunchKernelGGL(k_bow_descend, dim3((cap_f + 15) / 16, batch), dim3(256), 0, (hipStream_t)stream, A);
    hipLaunchKernelGGL(k_bow_build, dim3(batch), dim3(256), (size_t)sortN * 8 + 64, (hipStream_t)stream, A);
    return hipGetLastError() == hipSuccess ? ORB_OK : ORB_E_HIP;
}
