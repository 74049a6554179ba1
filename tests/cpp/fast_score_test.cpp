// CPU unit test of k_fast's score functions (test infrastructure): the product source is compiled against the HIP emulator header and
// fast_S_pk — ONE polarity picked by a count of the brighter ring pixels, arc minima by running minima on two 16-bit lanes per word —
// is compared with fast_S, the straightforward both-polarity form, on random and adversarial 7 x 7 patches.
// Contract (orbx_extractor.hip): fast_S_pk == fast_S wherever fast_S > 0; some value <= 0 elsewhere.
#include "../../awesome-orb-slam3-3dvisioncraft-version_amd/csrc/orbx_extractor.hip"

#include <cstdio>
#include <cstdlib>
#include <random>

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    std::mt19937_64 rng(12345);
    uint8_t p[7 * 7];
    long bad = 0, pos = 0;
    for (long it = 0; it < n; it++) {
        const int kind = (int)(rng() % 6);
        const int base = (int)(rng() % 256), amp = 1 + (int)(rng() % 255);
        for (int i = 0; i < 49; i++) {
            int v;
            switch (kind) {
                case 0: v = (int)(rng() % 256); break;                                          // noise
                case 1: v = base + (int)(rng() % (2 * amp + 1)) - amp; break;                   // noise around a level
                case 2: v = ((i % 7) + (int)(rng() % 3) > 3 + (int)(it % 3) - 1) ? base + amp : base; break;   // vertical edge, jittered
                case 3: v = ((i / 7) * (1 + (int)(it % 3)) + (i % 7) > 6 + (int)(rng() % 2)) ? base : base + amp; break;   // slanted edge / corner
                case 4: v = (rng() & 1) ? base + amp : base - amp; break;                       // two levels around the centre
                default: v = base + ((i * 37 + (int)(it % 11)) % 5 - 2) * (amp / 8 + 1); break; // ramps
            }
            p[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
        if (kind == 4) p[24] = (uint8_t)base;
        if (it % 7 == 0) p[24] = (rng() & 1) ? 0 : 255;                                          // extreme centres
        const int a = fast_S(p + 24, 7), b = fast_S_pk(p + 24, 7);
        pos += a > 0;
        if (a > 0 ? a != b : b > 0) {
            if (bad++ < 5) fprintf(stderr, "mismatch: fast_S %d fast_S_pk %d (case %ld, kind %d)\n", a, b, it, kind);
        }
    }
    printf("fast score: %ld cases, %ld with a positive score, %ld mismatches\n", n, pos, bad);
    return bad ? 1 : 0;
}
