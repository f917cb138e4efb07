"""The correctly rounded f32 of the REAL sine of an f64 argument, in pure Python (decimal, 130 digits) -- TEST INFRASTRUCTURE, independent of any libm: what
mixlab_amd/csrc/mx_sin_f32.hpp's double-double slow path has to produce.  Slow (about a millisecond per value): thousands of arguments, not millions."""
from decimal import Decimal, getcontext
from fractions import Fraction

import numpy as np

getcontext().prec = 130
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899862803482534211706798214808651328230664709384460955058223172535940812848111745")


def sin_decimal(x: float) -> Decimal:
    d = Decimal(x)                                   # exact
    k = (d / (2 * PI)).to_integral_value()
    r = d - k * 2 * PI                               # |r| <= pi, absolute error ~ |k| 1e-129
    term, total, n = r, r, 1
    r2 = r * r
    while abs(term) > Decimal(10) ** -125:
        term = -term * r2 / ((n + 1) * (n + 2))
        total += term
        n += 2
    return total


def round_to_f32(v: Decimal) -> np.float32:
    """nearest float32, ties to even, from the exact rational value"""
    fr = Fraction(v)
    f = np.float32(float(v))                         # a candidate; the true nearest is it or a neighbour
    best, best_err = None, None
    for c in (np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))):
        err = abs(Fraction(float(c)) - fr)
        even = (int(np.float32(c).view(np.uint32)) & 1) == 0
        if best is None or err < best_err or (err == best_err and even):
            best, best_err = np.float32(c), err
    return best


def sin_f32(x: float) -> np.float32:
    if x == 0.0:
        return np.float32(x)                         # sin(+-0) = +-0
    return round_to_f32(sin_decimal(float(x)))
