// CPU unit test of the round-4 arithmetic of k_describe2 (test infrastructure): the product source is compiled against the HIP emulator header.
//  (1) rint_bits(x) - RINT_BIAS == lrintf(x) (cvRound: ties to even) for every float the rotation can produce and well beyond (|x| < 2^22);
//  (2) the sampled point's byte address inside the row-pass buffer, built from the RAW rint_bits of (r, c) by one 24-bit multiply with every
//      bias folded into one constant, equals the plain form 2 * ((18 + c) * DRP + ((18 + r) & ~1)), and the realignment shift equals 2 * ((18 + r) & 1);
//  (3) fast_atan2_deg as selects around one division == the reference's two-branch form (ORBextractor.cc:91-93 -> cv::fastAtan2), bit for bit.
#include "../../awesome-orb-slam3-3dvisioncraft-version_amd/csrc/orbx_extractor.hip"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

static float atan2_two_branch(float y, float x) {
    const float s = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s, p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

int main() {
    long bad = 0, n = 0;
    std::mt19937_64 rng(99);
    // (1) every half-integer and its float neighbours in [-64, 64], then random values up to 2^22
    for (int h = -128; h <= 128; h++) {
        const float c = 0.5f * (float)h;
        const float xs[5] = {c, nextafterf(c, 1e9f), nextafterf(c, -1e9f), c + 0.25f, c - 0.25f};
        for (float x : xs) { n++; if ((int)(rint_bits(x) - RINT_BIAS) != (int)lrintf(x)) { if (bad++ < 5) fprintf(stderr, "rint %a\n", x); } }
    }
    for (long i = 0; i < 4000000; i++) {
        const int e = (int)(rng() % 46) - 24;                      // 2^-24 .. 2^21
        float x = ldexpf((float)((rng() >> 40) | (1u << 23)) / 8388608.0f, e);
        if (rng() & 1) x = -x;
        if (fabsf(x) >= 4194304.0f) continue;
        n++;
        if ((int)(rint_bits(x) - RINT_BIAS) != (int)lrintf(x)) { if (bad++ < 5) fprintf(stderr, "rint %a\n", x); }
    }
    // (2) the address arithmetic of k_describe2's blurred() for every (r, c) the pattern can reach
    const uint32_t K = (uint32_t)(18 * 2 * DRP + 36) - 0x400000u * (uint32_t)(2 * DRP) - 2u * RINT_BIAS;
    for (int r = -19; r <= 19; r++)
        for (int c = -19; c <= 19; c++) {
            const uint32_t rb = rint_bits((float)r), cb = rint_bits((float)c);
            const uint32_t r2 = rb + rb;
            const uint32_t off = (uint32_t)imul24((int)cb, 2 * DRP) + K + (r2 & ~3u), sh = r2 & 2u;
            const uint32_t want = 2u * (uint32_t)((18 + c) * DRP + ((18 + r) & ~1)), wsh = 2u * (uint32_t)((18 + r) & 1);
            n++;
            if (off != want || sh != wsh) { if (bad++ < 5) fprintf(stderr, "addr r %d c %d: %u / %u, shift %u / %u\n", r, c, off, want, sh, wsh); }
        }
    // (3) fastAtan2: integer moments as IC_Angle produces them, plus ties |x| == |y| and zeros
    for (long i = 0; i < 3000000; i++) {
        int mx = (int)(rng() % 2000001) - 1000000, my = (int)(rng() % 2000001) - 1000000;
        if (i % 7 == 0) my = (rng() & 1) ? mx : -mx;
        if (i % 11 == 0) mx = 0;
        if (i % 13 == 0) my = 0;
        const float a = fast_atan2_deg((float)my, (float)mx), b = atan2_two_branch((float)my, (float)mx);
        n++;
        if (memcmp(&a, &b, 4) != 0) { if (bad++ < 5) fprintf(stderr, "atan2 %d %d: %a %a\n", my, mx, a, b); }
    }
    printf("%ld cases, %ld mismatches\n", n, bad);
    return bad ? 1 : 0;
}
