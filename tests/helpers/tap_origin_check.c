/* The tiled scaler computes its LDS window origin with an f64 formula (mx_k_video.hip: sc_first_tap) instead of a
 * 64-bit integer division; this checks the formula against the integer tap spec (DESIGN.md "Scaler"). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
static int64_t floor_div(int64_t a, int64_t b) { int64_t q = a / b; if ((a % b) < 0) --q; return q; }
static int ref_first(uint32_t o, uint32_t src, uint32_t dst) {
    const int64_t pos = floor_div((2 * (int64_t)o + 1) * (int64_t)src * 65536, 2 * (int64_t)dst) - 32768;
    return (int)(pos >> 16) - 1;
}
static int f64_first(uint32_t o, uint32_t src, uint32_t dst) {
    const double n = (double)(2u * o + 1u) * ((double)src * 65536.0), d = 2.0 * (double)dst;
    double q = floor(n / d);
    const double r = fma(-q, d, n);
    q += (r >= d) ? 1.0 : ((r < 0.0) ? -1.0 : 0.0);
    return (int)floor((q - 32768.0) * (1.0 / 65536.0)) - 1;
}
int main() {
    uint64_t bad = 0, n = 0;
    srand(1);
    for (int it = 0; it < 200000; ++it) {
        uint32_t src = 1 + rand() % 65535, dst = 1 + rand() % 65535;
        for (int k = 0; k < 64; ++k) { uint32_t o = rand() % dst; ++n; if (ref_first(o, src, dst) != f64_first(o, src, dst)) ++bad; }
        uint32_t o = dst - 1; ++n; if (ref_first(o, src, dst) != f64_first(o, src, dst)) ++bad;
        ++n; if (ref_first(0, src, dst) != f64_first(0, src, dst)) ++bad;
    }
    // exhaustive for common sizes
    uint32_t sz[] = {120, 180, 212, 320, 360, 540, 640, 720, 960, 1080, 1280, 1920, 2160, 3840};
    for (unsigned i = 0; i < sizeof sz / sizeof sz[0]; ++i) for (unsigned j = 0; j < sizeof sz / sizeof sz[0]; ++j) for (uint32_t o = 0, s = sz[i], d = sz[j]; o < d; ++o) { ++n; if (ref_first(o, s, d) != f64_first(o, s, d)) ++bad; }
    printf("checked %llu, mismatches %llu\n", (unsigned long long)n, (unsigned long long)bad);
    return bad != 0;
}
