// CPU unit test of k_fast's score functions (test infrastructure): the product source is compiled against the HIP emulator header and
// fast_S_pk — ONE polarity picked by a count of the brighter ring pixels, arc minima by running minima on two 16-bit lanes per word —
// is compared with fast_S, the straightforward both-polarity form, on random and adversarial 7 x 7 patches.
// Contract (orbx_extractor.hip): fast_S_pk == fast_S wherever fast_S > 0; some value <= 0 elsewhere.
// Second part: the byte-parallel pre-test (fast_pretest4, four pixels per call) is a NECESSARY condition at every threshold 0 .. 255: whenever a pixel
// is a FAST-9 corner at threshold t (fast_S > t), its bit is set — the pre-test may let non-corners through, never drop a corner.
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
    // ---- pre-test: a 7-row strip at the kernel's LDS pitch, the centre dword = pixels (row 3, bytes 4 .. 7)
    static uint8_t strip[7 * FAST_PITCH + 16] __attribute__((aligned(16)));
    long dropped = 0, corners = 0, passed = 0, tested = 0;
    for (long it = 0; it < n / 8; it++) {
        const int kind = (int)(rng() % 4), base = (int)(rng() % 256), amp = 1 + (int)(rng() % 255);
        for (int r = 0; r < 7; r++)
            for (int c = 0; c < 16; c++) {
                int v;
                switch (kind) {
                    case 0: v = (int)(rng() % 256); break;
                    case 1: v = base + (int)(rng() % (2 * amp + 1)) - amp; break;
                    case 2: v = (c + (int)(rng() % 2) > 5 + (int)(it % 4)) ? base + amp : base; break;
                    default: v = (r * (1 + (int)(it % 3)) + c > 9 + (int)(rng() % 2)) ? base : base + amp; break;
                }
                strip[r * FAST_PITCH + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        int S[4];
        for (int t = 0; t < 4; t++) S[t] = fast_S(strip + 3 * FAST_PITCH + 4 + t, FAST_PITCH);
        for (int th = 0; th < 256; th += (it & 1) ? 1 : 7) {
            uint32_t KB, KG;
            fast_pretest_consts(th, &KB, &KG);
            const uint32_t bits = fast_pretest4((const uint32_t*)strip, KB, KG);
            for (int t = 0; t < 4; t++) {
                const bool corner = S[t] > th, pass = (bits >> (8 * t + 7)) & 1u;
                tested++; corners += corner; passed += pass;
                if (corner && !pass && dropped++ < 5) fprintf(stderr, "pre-test dropped a corner: S %d threshold %d (case %ld, pixel %d)\n", S[t], th, it, t);
            }
        }
    }
    printf("pre-test: %ld (pixel, threshold) pairs, %ld corners, %ld passed, %ld corners dropped\n", tested, corners, passed, dropped);
    return (bad || dropped) ? 1 : 0;
}
