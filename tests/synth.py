"""Synthetic inputs shared by tests and bench: SplitMix64 noise, uniform f32 in [-1, 1)."""
from __future__ import annotations

import numpy as np

SEED_BASE = 0x4D58_0000
_M = (1 << 64) - 1


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n successive SplitMix64 outputs for `seed` (vectorised: state_i = seed + (i+1)*golden)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed & _M) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def noise(index: int, n: int) -> np.ndarray:
    """uniform f32 in [-1, 1): top 24 bits of SplitMix64(seed = SEED_BASE + index)."""
    u = splitmix64(SEED_BASE + index, n) >> np.uint64(40)
    return (u.astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)).astype(np.float32)


def uniform(index: int, n: int, lo: float, hi: float) -> np.ndarray:
    u = (splitmix64(SEED_BASE + 0x100000 + index, n) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
    return lo + (hi - lo) * u


def ulp_diff(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """distance in f32 ULPs between two float32 arrays (monotone integer mapping of the bit patterns)."""
    def key(x):
        i = np.ascontiguousarray(x, dtype=np.float32).view(np.int32).astype(np.int64)
        return np.where(i < 0, np.int64(-0x80000000) - i, i)
    return np.abs(key(a) - key(b))


def yuv_pattern(w: int, h: int, layer: int, seed: int = 0, fmt: int = 0):
    """SURVEY.md section 8d config 4 pattern, planar YUV (fmt 0 yuv420p, 1 yuv422p, 2 yuv444p): Y(x,y) = (x + 2y + 31*layer + LCG
    noise) mod 256, U/V alike at the format's chroma resolution.  Pure numpy -- bench.py and the tests' HostFrame.fill share it."""
    planes = []
    cw, ch = (0 if fmt in (2, 8) else (2 if fmt in (6, 7) else 1)), (1 if fmt in (0, 8) else (2 if fmt == 6 else 0))   # 6 yuv410p, 7 yuv411p, 8 yuv440p
    for p in range(3):
        hh, ww = h >> (ch if p else 0), w >> (cw if p else 0)
        yy, xx = np.mgrid[0:hh, 0:ww].astype(np.uint32)
        lcg = ((xx * np.uint32(1664525) + yy * np.uint32(1013904223) + np.uint32((seed * 7919 + layer * 104729 + p * 31337) & 0xffffffff)) >> np.uint32(13)) & np.uint32(15)
        planes.append(((xx + 2 * yy + 31 * layer + 57 * p + lcg) & np.uint32(255)).astype(np.uint8))
    return planes
