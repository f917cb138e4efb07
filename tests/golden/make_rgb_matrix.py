#!/usr/bin/env python
"""The integer coefficients of the build-specified packed-RGB -> YUV conversion (DESIGN.md "Pixel formats"), derived from the BT.709 primaries
with exact rationals -- independent of the oracle's and the kernel's hard-coded constants, which tests/test_cpu_oracle_and_abi.py checks
against this file.   Y = 16 + 219 (Kr R + Kg G + Kb B) / 255,  Cb = 128 + 224 (B' - Y') / (2 (1 - Kb)) / 255,  Cr likewise with Kr;
coefficients scaled by 256 and rounded to nearest.   usage: python tests/golden/make_rgb_matrix.py > tests/golden/rgb_matrix_bt709.json"""
import json
from fractions import Fraction as F


def rnd(x: F) -> int:            # round half away from zero (no tie occurs for these values)
    return int(x + F(1, 2)) if x >= 0 else -int(-x + F(1, 2))


Kr, Kb = F(2126, 10000), F(722, 10000)
Kg = 1 - Kr - Kb
y = [rnd(F(219, 255) * k * 256) for k in (Kr, Kg, Kb)]
cb = [rnd(F(224, 255) * k * 256) for k in (-Kr / (2 * (1 - Kb)), -Kg / (2 * (1 - Kb)), F(1, 2))]
cr = [rnd(F(224, 255) * k * 256) for k in (F(1, 2), -Kg / (2 * (1 - Kr)), -Kb / (2 * (1 - Kr)))]
print(json.dumps({"spec": "Y = ((y . rgb + 128) >> 8) + 16, U = ((cb . rgb + 128) >> 8) + 128, V = ((cr . rgb + 128) >> 8) + 128 (arithmetic shifts)",
                  "y": y, "cb": cb, "cr": cr}, indent=1))
