#!/usr/bin/env python
"""Tap tables of the build-specified bicubic scaler, derived from the TEXT of DESIGN.md section 6 with exact rational arithmetic
(fractions.Fraction) -- an implementation that shares no code with oracle/mixlab_oracle_video.c or mixlab_amd/csrc/mx_video.cpp.
tests/test_cpu_oracle_and_abi.py checks BOTH of those against the files this script writes (tests/golden/bicubic_taps_*.json).

    python tests/golden/make_bicubic_taps.py

Spec.  Output sample o of `dst` from `src` samples:
    pos = floor(((2o + 1) * src * 65536) / (2 * dst)) - 32768          (Q16 position of the output sample's centre)
    ip = floor(pos / 65536),  d = pos - ip * 65536
  cubic kernel, B = 0, C = 0.6, of a distance x >= 0:
    x < 1:      (7 x^3 - 12 x^2 + 5) / 5
    1 <= x < 2: (-3 x^3 + 15 x^2 - 24 x + 12) / 5
    else 0
  Q14(w) = floor(w * 16384 + 1/2), evaluated at x = X / 65536 for an integer Q16 distance X.
  src <= dst (up-scaling, 1:1): four taps, first = ip - 1, at distances 65536 + d, d, 65536 - d, 131072 - d;
    coefficients Q14(cubic(.)); the residual 16384 - sum goes to tap 2 if it is larger than tap 1, else to tap 1.
  src > dst (down-scaling): N = 2 * ceil(2 * src / dst) + 2 taps, first = ip - N / 2 + 1; tap k sits at source index i = first + k,
    raw_k = Q14(cubic(floor(|i * 65536 - pos| * dst / src) / 65536));  coefficient_k = floor(raw_k * 16384 / sum(raw) + 1/2);
    the residual 16384 - sum goes to the first largest coefficient.
"""
import json
import pathlib
from fractions import Fraction
from math import floor

HERE = pathlib.Path(__file__).resolve().parent
GEOMETRIES = [(720, 1080), (1080, 1080), (1000, 1001), (1080, 635), (1280, 100), (959, 539), (360, 540)]


def cubic(x: Fraction) -> Fraction:
    if x < 1:
        return (7 * x ** 3 - 12 * x ** 2 + 5) / 5
    if x < 2:
        return (-3 * x ** 3 + 15 * x ** 2 - 24 * x + 12) / 5
    return Fraction(0)


def q14(X: int) -> int:
    return floor(cubic(Fraction(X, 65536)) * 16384 + Fraction(1, 2))


def taps(o: int, src: int, dst: int):
    pos = floor(Fraction((2 * o + 1) * src * 65536, 2 * dst)) - 32768
    ip = pos // 65536
    d = pos - ip * 65536
    if src <= dst:
        c = [q14(65536 + d), q14(d), q14(65536 - d), q14(131072 - d)]
        resid = 16384 - sum(c)
        if c[2] > c[1]:
            c[2] += resid
        else:
            c[1] += resid
        return ip - 1, c
    n = 2 * -(-2 * src // dst) + 2
    first = ip - n // 2 + 1
    raw = [q14(floor(Fraction(abs((first + k) * 65536 - pos) * dst, src))) for k in range(n)]
    s = sum(raw)
    c = [floor(Fraction(r * 16384, s) + Fraction(1, 2)) for r in raw]
    c[c.index(max(c))] += 16384 - sum(c)
    return first, c


def main():
    for src, dst in GEOMETRIES:
        t = [taps(o, src, dst) for o in range(dst)]
        assert all(sum(c) == 16384 for _f, c in t)
        out = {"src": src, "dst": dst, "n_taps": len(t[0][1]), "first": [f for f, _c in t], "coef": [c for _f, c in t]}
        (HERE / f"bicubic_taps_{src}_to_{dst}.json").write_text(json.dumps(out, separators=(",", ":")) + "\n")
        print(src, "->", dst, ":", len(t[0][1]), "taps")


if __name__ == "__main__":
    main()
