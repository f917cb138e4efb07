// mx_sin_f32.hpp -- `sin(x) as f32` for the Sine oscillator and FmSine (src/module/oscillator.rs:25-27,77-80, src/module/fm_sine.rs:44-52), made
// independent of whose libm computed the f64 sine.
//
// The reference takes the host libm's f64 sin and casts it to f32.  glibc's sin is within 0.55 f64 ulp of the real sine, the device library's within a
// couple of ulp: both carry 29 bits more than the f32 keeps, so the two casts agree unless the real sine lies within those few f64 ulp of an f32 ROUNDING
// BOUNDARY (the midpoint of two neighbouring floats) -- about once in 2^26 samples.  Ziv's strategy: take the fast f64 result when the whole interval
// [y - E, y + E] rounds to one float; otherwise redo that lane in double-double arithmetic (a 161-bit reduction by pi/2 and a Taylor series to ~2^-100)
// and round the real sine correctly.  What is left is the case where the REAL sine lies within glibc's own error (0.55 f64 ulp) of a boundary -- there the
// reference's float depends on glibc's last-bit error and nothing short of glibc's own code reproduces it: <= 1.1 / 2^29, about 2 samples in 10^9
// (counted on 50 M module-shaped arguments: DESIGN.md "Sine").
//
// Plain C++ (fma, rint): compiled by hipcc for the device and by g++ for the CPU check of the slow path (tests/helpers/sin_f32_check.cpp).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define MX_SIN_HD __host__ __device__
#define MX_SIN_NOINLINE __noinline__
#else
#define MX_SIN_HD
#define MX_SIN_NOINLINE __attribute__((noinline))
#endif

namespace mx {

struct dd_t { double h, l; };   // h + l, |l| <= ulp(h) / 2

MX_SIN_HD inline dd_t dd_two_sum(double a, double b) { const double s = a + b, bb = s - a; return {s, (a - (s - bb)) + (b - bb)}; }
MX_SIN_HD inline dd_t dd_fast_two_sum(double a, double b) { const double s = a + b; return {s, b - (s - a)}; }   // |a| >= |b|
MX_SIN_HD inline dd_t dd_add(dd_t a, dd_t b) {
    dd_t s = dd_two_sum(a.h, b.h);
    const dd_t t = dd_two_sum(a.l, b.l);
    s.l += t.h;
    s = dd_fast_two_sum(s.h, s.l);
    s.l += t.l;
    return dd_fast_two_sum(s.h, s.l);
}
MX_SIN_HD inline dd_t dd_mul(dd_t a, dd_t b) {
    const double p = a.h * b.h;
    double e = fma(a.h, b.h, -p);
    e = fma(a.h, b.l, e);
    e = fma(a.l, b.h, e);
    return dd_fast_two_sum(p, e);
}

// x - m pi/2 as a double-double, |x| < 2^40, m = rint(x 2/pi).  pi/2 = HI + MID + LO (161 bits).  fma(-m, HI, x) is exact: the difference is a multiple
// of 2^-53 below 1 in magnitude (m = 0: it is x).  m MID is split exactly (two-product), joined with a two-sum; what is left out is below m 2^-163.
MX_SIN_HD inline dd_t sin_reduce_pio2(double x, long long* quadrant) {
    const double HI = 0x1.921fb54442d18p+0, MID = 0x1.1a62633145c07p-54, LO = -0x1.f1976b7ed8fbcp-110;
    const double m = rint(x * 0x1.45f306dc9c883p-1);
    *quadrant = (long long)m;
    const double r1 = fma(-m, HI, x);
    const double p2 = m * MID, e2 = fma(m, MID, -p2);
    const dd_t s = dd_two_sum(r1, -p2);
    const double c = (s.l - e2) - m * LO;
    return dd_two_sum(s.h, c);
}

// the real sin(x), |x| < 2^40, as a double-double good to ~2^-100 of its magnitude (2^-56 of it in the one corner where x lies within 1e-20 of a multiple of pi / 2)
MX_SIN_HD MX_SIN_NOINLINE inline dd_t sin_dd(double x) {
    static const double S[13][2] = {   // (-1)^k / (2k + 1)!, k = 1 .. 13
        {-0x1.5555555555555p-3, -0x1.5555555555555p-57}, {0x1.1111111111111p-7, 0x1.1111111111111p-63}, {-0x1.a01a01a01a01ap-13, -0x1.a01a01a01a01ap-73},
        {0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73}, {-0x1.ae64567f544e4p-26, 0x1.c062e06d1f209p-80}, {0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87},
        {-0x1.ae7f3e733b81fp-41, -0x1.1d8656b0ee8cbp-97}, {0x1.952c77030ad4ap-49, 0x1.ac981465ddc6cp-103}, {-0x1.2f49b46814157p-57, -0x1.2650f61dbdcb4p-112},
        {0x1.71b8ef6dcf572p-66, -0x1.d043ae40c4647p-120}, {-0x1.761b41316381ap-75, 0x1.3423c7d91404fp-130}, {0x1.3f3ccdd165fa9p-84, -0x1.58ddadf344487p-139},
        {-0x1.d1ab1c2dccea3p-94, -0x1.054d0c78aea14p-149}};
    static const double Cc[14][2] = {   // (-1)^k / (2k)!, k = 1 .. 14
        {-0x1.0000000000000p-1, 0.0}, {0x1.5555555555555p-5, 0x1.5555555555555p-59}, {-0x1.6c16c16c16c17p-10, 0x1.f49f49f49f49fp-65},
        {0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76}, {-0x1.27e4fb7789f5cp-22, -0x1.cbbc05b4fa99ap-76}, {0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83},
        {-0x1.93974a8c07c9dp-37, -0x1.05d6f8a2efd1fp-92}, {0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101}, {-0x1.6827863b97d97p-53, -0x1.eec01221a8b0bp-107},
        {0x1.e542ba4020225p-62, 0x1.ea72b4afe3c2fp-120}, {-0x1.0ce396db7f853p-70, 0x1.aebcdbd20331cp-124}, {0x1.f2cf01972f578p-80, -0x1.9ada5fcc1ab14p-135},
        {-0x1.88e85fc6a4e5ap-89, 0x1.71c37ebd16540p-143}, {0x1.0a18a2635085dp-98, 0x1.b9e2e28e1aa54p-153}};
    if (x == 0.0) return {x, 0.0};   // sin(+-0) = +-0
    long long q;
    const dd_t r = sin_reduce_pio2(x, &q);
    const dd_t z = dd_mul(r, r);
    dd_t v;
    if ((q & 1LL) == 0) {   // sin r = r + r z (S1 + z (S2 + ...))
        dd_t p = {S[12][0], S[12][1]};
        for (int k = 11; k >= 0; --k) p = dd_add(dd_mul(p, z), dd_t{S[k][0], S[k][1]});
        v = dd_add(r, dd_mul(dd_mul(r, z), p));
    } else {                // cos r = 1 + z (C1 + z (C2 + ...))
        dd_t p = {Cc[13][0], Cc[13][1]};
        for (int k = 12; k >= 0; --k) p = dd_add(dd_mul(p, z), dd_t{Cc[k][0], Cc[k][1]});
        v = dd_add(dd_t{1.0, 0.0}, dd_mul(z, p));
    }
    if (q & 2LL) { v.h = -v.h; v.l = -v.l; }
    return v;
}

// h + l rounded to the nearest float (ties to even) without the double rounding of (float)(h + l).  h is the f64 nearest to the sum and an f32 rounding boundary
// is itself an f64, so (float)h is right unless h IS a boundary -- then the sign of l says on which side of it the sum lies.
MX_SIN_HD inline float dd_to_f32(dd_t v) {
    const float f = (float)v.h;
    const double d = v.h - (double)f;                                        // exact
    if (d != 0.0 && v.l != 0.0) {
        const float n = nextafterf(f, d > 0.0 ? INFINITY : -INFINITY);       // the float on the other side of h
        if (d == ((double)n - (double)f) * 0.5) return ((v.l > 0.0) == (d > 0.0)) ? n : f;   // h is the midpoint of f and n (f was the tie-to-even pick)
    }
    return f;
}

// `y` is an f64 sine of x good to `ulps` f64 ulp: the float every value of [y - E, y + E] rounds to, or the correctly rounded real sine where that interval
// straddles a rounding boundary.  E also covers glibc's 0.55 ulp, so a result taken on the fast path is the reference's float.
MX_SIN_HD inline float sin_f32_from(double x, double y, double ulps_plus) {
    const double e = fabs(y) * (ulps_plus * 0x1p-52) + 0x1p-1070;
    const float lo = (float)(y - e), hi = (float)(y + e);
    if (lo == hi || !(fabs(x) < 1099511627776.0)) return (float)y;   // (also NaN: both compare false, x fails the range test; 2^40 and beyond, inf: the library's own)
    return dd_to_f32(sin_dd(x));
}

}  // namespace mx
