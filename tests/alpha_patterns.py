"""Coverage (alpha) planes for the per-pixel alpha tests: numpy only, shared by the CPU and GPU suites and bench.py."""
from __future__ import annotations

import numpy as np

PATTERNS = ("opaque", "random", "soft-disc")


def alpha_plane(w: int, h: int, pattern: str, seed: int = 0) -> np.ndarray:
    """(h, w) uint8 coverage: `opaque` = 255 everywhere (must reproduce the alpha-free picture bit for bit), `random` = every byte value with exact 0 and 255
    present, `soft-disc` = a disc with a soft edge over a fully transparent surround (what a keyed or titled layer looks like)."""
    if pattern == "opaque":
        return np.full((h, w), 255, np.uint8)
    if pattern == "random":
        rng = np.random.default_rng(0xA1FA + seed * 7919 + w * 31 + h)
        a = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        a.flat[: min(8, a.size)] = [0, 255, 1, 254, 127, 128, 0, 255][: min(8, a.size)]
        a[h // 2, : w // 3] = 0
        a[h // 3, w // 2:] = 255
        return a
    if pattern == "soft-disc":
        yy, xx = np.mgrid[0:h, 0:w]
        cx, cy, r = w * (0.45 + 0.02 * (seed % 5)), h * 0.55, min(w, h) * 0.38
        d = np.hypot(xx - cx, (yy - cy) * 1.1)
        edge = max(2.0, r * 0.15)
        return np.clip((r - d) / edge * 255.0 + 128.0, 0, 255).astype(np.uint8)
    raise ValueError(pattern)


def crossfade_alpha_numpy(a, b, aa, ab, fade, cw=0, ch=0):
    """The build-specified rule on one plane in plain numpy (an independent restatement for the oracle's own test): a, b (ph, pw) uint8, aa / ab (h, w) uint8 luma-resolution
    coverage or None, fade = (fader * 255) as u8, cw / ch the plane's log2 chroma subsampling."""
    a16, b16 = a.astype(np.uint32), b.astype(np.uint32)
    al_a = np.full(a.shape, 255, np.uint32) if aa is None else aa[:: 1 << ch, :: 1 << cw][: a.shape[0], : a.shape[1]].astype(np.uint32)
    al_b = np.full(a.shape, 255, np.uint32) if ab is None else ab[:: 1 << ch, :: 1 << cw][: a.shape[0], : a.shape[1]].astype(np.uint32)
    wa = (al_a * fade) // 255
    wb = (al_b * (255 - wa)) // 255
    return ((a16 * (255 - wb) + b16 * wb) // 255).astype(np.uint8)
