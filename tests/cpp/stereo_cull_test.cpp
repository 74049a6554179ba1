// CPU unit test of k_stereo_cull (test infrastructure; the product source against the HIP emulator): the two-level-histogram median must be the value at
// rank size / 2 of the ascending order of the valid SADs (Frame.cc:1120-1123), and the cull 1.5 * 1.4 * median, on adversarial inputs: ties, a single
// valid entry, all equal, values on bin boundaries, the largest possible SAD (120 * 510 = 61 200: centre-subtracted patches), medians above 32 767,
// mostly-invalid frames.
#include "../../awesome-orb-slam3-3dvisioncraft-version_amd/csrc/orbx_extractor.hip"

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

int main() {
    std::mt19937 rng(7);
    const int cap = 1300, frames = 64;
    std::vector<int32_t> cnt(2 * frames), sad((size_t)frames * cap);
    std::vector<float> ur((size_t)frames * cap), dp((size_t)frames * cap);
    for (int f = 0; f < frames; f++) {
        const int n = f == 0 ? 0 : f == 1 ? 1 : 2 + (int)(rng() % (cap - 2));
        cnt[2 * f] = n;
        const int kind = f % 8;
        for (int i = 0; i < cap; i++) {
            int s;
            switch (kind) {
                case 0: s = (int)(rng() % 61201); break;                                   // whole range: 0 .. 120 * 510
                case 1: s = 777; break;                                                    // all equal
                case 2: s = (rng() % 3) ? -1 : (int)(rng() % 4000); break;                 // mostly invalid
                case 3: s = 32 * (int)(rng() % 1912) + ((rng() & 1) ? 31 : 0); break;      // bin boundaries
                case 4: s = (rng() % 5) ? 1000 + (int)(rng() % 8) : 61200; break;          // ties + the maximum
                case 5: s = (f & 8) ? 61137 + (int)(rng() % 64) : (int)(rng() % 64); break;  // two bins: the lowest two, or the highest two (median >= 32 768)
                case 6: s = i < 3 ? 5 * i : -1; break;                                     // three valid entries
                default: s = 2000 + (int)(rng() % 33); break;                              // straddles one boundary
            }
            sad[(size_t)f * cap + i] = s;
            ur[(size_t)f * cap + i] = s >= 0 ? 10.0f + i : -1.0f;
            dp[(size_t)f * cap + i] = s >= 0 ? 2.0f : -1.0f;
        }
    }
    std::vector<float> ur0 = ur;
    StereoParams P;
    memset(&P, 0, sizeof(P));
    P.cntL = cnt.data(); P.cap = cap; P.sad = sad.data(); P.uRight = ur.data(); P.depth = dp.data();
    hipLaunchKernelGGL(k_stereo_cull, dim3(frames), dim3(256), (size_t)(STEREO_CULL_BINS + 32 + 256 + 4) * 4, (hipStream_t)0, P);
    long bad = 0, culled = 0;
    for (int f = 0; f < frames; f++) {
        const int n = std::min(cnt[2 * f], cap);
        std::vector<int> v;
        for (int i = 0; i < n; i++) if (sad[(size_t)f * cap + i] >= 0) v.push_back(sad[(size_t)f * cap + i]);
        std::sort(v.begin(), v.end());
        const float th = v.empty() ? 0.f : 1.5f * 1.4f * (float)v[v.size() / 2];
        for (int i = 0; i < cap; i++) {
            const int s = sad[(size_t)f * cap + i];
            const bool cull = !v.empty() && i < n && s >= 0 && !((float)s < th);
            const float want = cull ? -1.0f : ur0[(size_t)f * cap + i];
            culled += cull;
            if (ur[(size_t)f * cap + i] != want || (cull && dp[(size_t)f * cap + i] != -1.0f)) {
                if (bad++ < 5) fprintf(stderr, "frame %d entry %d: sad %d threshold %g: uRight %g, expected %g\n", f, i, s, th, ur[(size_t)f * cap + i], want);
            }
        }
    }
    printf("stereo cull: %d frames, %ld entries culled, %ld mismatches\n", frames, culled, bad);
    return bad ? 1 : 0;
}
