/* The tiled / inline scalers compute their LDS window origin as floor(((2o+1) src + dst) / (2 dst)) - 2 in 32 bits, the quotient from
 * an f32 estimate corrected with the exact remainder (mx_k_video.hip: sc_first_tap), instead of a 64-bit integer division; this checks
 * it against the integer tap spec (DESIGN.md "Scaler") for every frame size the library accepts (<= 16384). */
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
    const float rcp2d = 1.0f / (float)(2u * dst);
    const uint32_t n = (2u * o + 1u) * src + dst, d = 2u * dst;
    uint32_t q = (uint32_t)((float)n * rcp2d);
    const int32_t r = (int32_t)(n - q * d);
    q += (r >= (int32_t)d) ? 1u : 0u;
    q -= (r < 0) ? 1u : 0u;
    return (int)q - 2;
}
int main() {
    uint64_t bad = 0, n = 0;
    srand(1);
    for (int it = 0; it < 200000; ++it) {
        uint32_t src = 1 + rand() % 16384, dst = 1 + rand() % 16384;
        for (int k = 0; k < 64; ++k) { uint32_t o = rand() % dst; ++n; if (ref_first(o, src, dst) != f64_first(o, src, dst)) ++bad; }
        uint32_t o = dst - 1; ++n; if (ref_first(o, src, dst) != f64_first(o, src, dst)) ++bad;
        ++n; if (ref_first(0, src, dst) != f64_first(0, src, dst)) ++bad;
    }
    // exhaustive for common sizes
    uint32_t sz[] = {1, 2, 3, 120, 180, 212, 320, 360, 540, 640, 720, 960, 1080, 1280, 1920, 2160, 3840, 8192, 16383, 16384};
    for (unsigned i = 0; i < sizeof sz / sizeof sz[0]; ++i) for (unsigned j = 0; j < sizeof sz / sizeof sz[0]; ++j) for (uint32_t o = 0, s = sz[i], d = sz[j]; o < d; ++o) { ++n; if (ref_first(o, s, d) != f64_first(o, s, d)) ++bad; }
    printf("checked %llu, mismatches %llu\n", (unsigned long long)n, (unsigned long long)bad);
    return bad != 0;
}
