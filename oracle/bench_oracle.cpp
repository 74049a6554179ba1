// bench_oracle.cpp — CPU baseline driver for bench.py's `cpu_baseline` leg.  TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/README.md):
// it times the oracle (the reference algorithm restated, oracle/orb_oracle.cpp + frame_oracle.cpp + match_oracle.cpp) in native threads on the
// GPU box's host cores; nothing in the product path links or calls it.
//
// One unit = what Tracking does per frame on the hot path (reference call order): ORBextractor::operator() (Frame.cc:111-137 constructor),
// Frame::UndistortKeyPoints (Frame.cc:874), AssignFeaturesToGrid (Frame.cc:444, inside the search call of the restatement) and
// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:2244-2509) with prepared projection records — the same
// unit bench.py's GPU step processes.  The reference runs one frame on one thread; here every native thread owns one extractor and walks its
// own share of the frames.
#include <malloc.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {
void* oro_create(int, float, int, int, int);
void oro_destroy(void*);
int oro_extract(void*, const uint8_t*, int, int, int, int, int, void*, uint8_t*, int, int*);
void ofr_undistort_keypoints(const void*, int, const float*, void*);
int omo_search_by_projection(const void*, const uint8_t*, const float*, const uint8_t*, int, float, float, float, float, const void*,
                             const uint8_t*, int, int, int, float, int, int32_t*, int32_t*);

// glibc hands every large vector (pyramid planes, candidate lists) to mmap/munmap by default: with dozens of threads the kernel's mm lock
// serialises them (round 1: 826 frames/s on 32 of 256 threads where 1 700 were ideal).  Keep those blocks on the per-thread arenas instead.
void oro_bench_tune_allocator(int nthreads) {
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    mallopt(M_ARENA_MAX, nthreads > 0 ? 2 * nthreads : 8);
}

// frames [B][H][W]; queries [B][cap_q] 28-byte records, qdesc [B][cap_q][32], nq [B]: the projection records of frame b's partner (prepared
// once, like the GPU leg's resident query slabs).  do_match = 0 times the extraction alone.
// Returns seconds; kp_total / match_total = checksums of the work done.
double oro_bench_extract_match_mt(const uint8_t* frames, int B, int W, int H, int nfeatures, float scaleFactor, int nlevels, int iniTh,
                                  int minTh, int lap0, int lap1, const float cam9[9], const float grid4[4], const uint8_t* queries,
                                  const uint8_t* qdesc, const int32_t* nq, int cap_q, int th_dist, float nnratio, int checkOri,
                                  int do_match, int nthreads, int frames_per_thread, long* kp_total, long* match_total) {
    oro_bench_tune_allocator(nthreads);
    std::vector<std::thread> th;
    std::vector<long> kp(nthreads, 0), mt(nthreads, 0);
    const int cap = 4 * nfeatures + 64;
    auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([&, t] {
            void* o = oro_create(nfeatures, scaleFactor, nlevels, iniTh, minTh);
            std::vector<uint8_t> k((size_t)cap * 28), ku((size_t)cap * 28), d((size_t)cap * 32);
            std::vector<int32_t> qm(cap_q > 0 ? cap_q : 1), km(cap);
            for (int i = 0; i < frames_per_thread; i++) {
                const int b = (t * frames_per_thread + i) % B;
                int n = 0;
                oro_extract(o, frames + (size_t)b * W * H, W, H, W, lap0, lap1, k.data(), d.data(), cap, &n);
                kp[t] += n;
                if (do_match) {
                    ofr_undistort_keypoints(k.data(), n, cam9, ku.data());
                    mt[t] += omo_search_by_projection(ku.data(), d.data(), nullptr, nullptr, n, grid4[0], grid4[1], grid4[2], grid4[3],
                                                      queries + (size_t)b * cap_q * 28, qdesc + (size_t)b * cap_q * 32, nq[b], 1, th_dist, nnratio,
                                                      checkOri, qm.data(), km.data());
                }
            }
            oro_destroy(o);
        });
    for (auto& x : th) x.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (kp_total) { *kp_total = 0; for (long v : kp) *kp_total += v; }
    if (match_total) { *match_total = 0; for (long v : mt) *match_total += v; }
    return s;
}
}  // extern "C"

// ---- the checker for the benchmark's own configuration (bench.py `parity_checked_frames`, tests/test_bench_config_parity.py) --------------
// The same per-frame unit as above for the frames `sel[0..nsel)` of the batch, in native threads, but KEEPING every output so that the GPU
// step can be compared frame by frame: key points (28-byte records, reference output order), descriptors, {n, monoIndex}
// (ORBextractor.cc:1074-1156), the undistorted records (Frame.cc:874-924), and the search's per-query match, mvpMapPoints image and match
// count (ORBmatcher.cc:2244-2509).  Output slabs are [nsel][cap] (queries: [nsel][cap_q]); rows beyond a frame's count are left untouched.
// Returns 0, or -(1 + i) if frame sel[i] produced more than `cap` key points.
extern "C" int oro_extract_match_frames_mt(const uint8_t* frames, int B, int W, int H, int nfeatures, float scaleFactor, int nlevels, int iniTh,
                                           int minTh, int lap0, int lap1, const float cam9[9], const float grid4[4], const uint8_t* queries,
                                           const uint8_t* qdesc, const int32_t* nq, int cap_q, int mode, int th_dist, float nnratio, int checkOri,
                                           int do_match, int nthreads, const int32_t* sel, int nsel, int cap, uint8_t* out_kps, uint8_t* out_desc,
                                           int32_t* out_counts, uint8_t* out_un, int32_t* out_qmatch, int32_t* out_kpmatch, int32_t* out_nm) {
    oro_bench_tune_allocator(nthreads);
    std::vector<std::thread> th;
    std::vector<int> bad(nthreads, 0);
    const int ocap = 4 * nfeatures + 64;
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([&, t] {
            void* o = oro_create(nfeatures, scaleFactor, nlevels, iniTh, minTh);
            std::vector<uint8_t> k((size_t)ocap * 28), ku((size_t)ocap * 28), d((size_t)ocap * 32);
            std::vector<int32_t> qm(cap_q > 0 ? cap_q : 1), km(ocap);
            for (int i = t; i < nsel; i += nthreads) {
                const int b = sel[i];
                if (b < 0 || b >= B) { bad[t] = -(1 + i); break; }
                int n = 0;
                const int mono = oro_extract(o, frames + (size_t)b * W * H, W, H, W, lap0, lap1, k.data(), d.data(), ocap, &n);
                if (n > cap) { bad[t] = -(1 + i); break; }
                out_counts[2 * i] = n; out_counts[2 * i + 1] = mono;
                std::memcpy(out_kps + (size_t)i * cap * 28, k.data(), (size_t)n * 28);
                std::memcpy(out_desc + (size_t)i * cap * 32, d.data(), (size_t)n * 32);
                if (!do_match) continue;
                ofr_undistort_keypoints(k.data(), n, cam9, ku.data());
                std::memcpy(out_un + (size_t)i * cap * 28, ku.data(), (size_t)n * 28);
                out_nm[i] = omo_search_by_projection(ku.data(), d.data(), nullptr, nullptr, n, grid4[0], grid4[1], grid4[2], grid4[3],
                                                     queries + (size_t)b * cap_q * 28, qdesc + (size_t)b * cap_q * 32, nq[b], mode, th_dist, nnratio,
                                                     checkOri, qm.data(), km.data());
                std::memcpy(out_qmatch + (size_t)i * cap_q, qm.data(), (size_t)nq[b] * 4);
                std::memcpy(out_kpmatch + (size_t)i * cap, km.data(), (size_t)n * 4);
            }
            oro_destroy(o);
        });
    for (auto& x : th) x.join();
    for (int v : bad) if (v) return v;
    return 0;
}
