/* Exhaustive check of the envelope kernel's division shortcut (mx_audio_kernels.hip: seq_ms):
 *   q = a*y; r = fma(-q, b, a); q' = fma(r, y, q)   with y = RN(1/b)
 * must equal the IEEE quotient a / b bit-for-bit for every integer a in [lo, hi).
 * Build: gcc -O2 -mfma -fopenmp -ffp-contract=off fastdiv_check.c -o fastdiv_check -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const double b = atof(argv[1]);
    const uint64_t lo = strtoull(argv[2], 0, 10), hi = strtoull(argv[3], 0, 10);
    const double y = 1.0 / b;
    uint64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (uint64_t i = lo; i < hi; i++) {
        const double a = (double)i;
        double q = a * y;
        const double r = fma(-q, b, a);
        q = fma(r, y, q);
        const double t = a / b;
        uint64_t qb, tb;
        memcpy(&qb, &q, 8); memcpy(&tb, &t, 8);
        bad += (qb != tb);
    }
    printf("%llu\n", (unsigned long long)bad);
    return bad ? 1 : 0;
}
