"""Row-band sharding of ONE composited picture over N ranks (mixlab_amd/shard.py): every rank composites
only its band of rows -- its own rows of the same-size layers, and for a layer that is scaled into the picture only the source
SLICE shard.band_source_rows() says its vertical taps reach (the halo).  Stitched, the bands must be the unsharded picture bit
for bit: cascade of reference cross-fades (src/module/video_mixer.rs:150-239) over the DynamicScaler's letterboxed scale
(src/video/encode.rs:338-397), then the build-specified RGBA conversion.  CPU only: the oracle is the compositor here."""
import ctypes as C

import numpy as np
import pytest

import oracle_video as ov
from mixlab_amd import shard
from oracle import OFrame, lib

lib.orc_dynamic_scale_band.argtypes = [C.POINTER(OFrame), C.c_uint32, C.c_uint32, C.POINTER(OFrame), C.c_uint32, C.c_uint32, C.c_uint32]
lib.orc_dynamic_scale_band.restype = C.c_int

FADERS = [1.0, 0.75, 0.5, 0.5, 0.25, 0.9, 0.1]
MATRIX = [3900, 150, 46, 4096, 60, 3980, 56, -2048, 20, 120, 3956, 0]


def rows_of(f: ov.HostFrame, row0: int, rows: int) -> ov.HostFrame:
    """A frame holding ONLY luma rows [row0, row0 + rows) of f (and the matching chroma rows): what a rank is sent."""
    b = ov.HostFrame(f.w, rows)
    for p in range(3):
        c = 1 if p else 0
        b.planes[p][:, :] = f.planes[p][row0 >> c:(row0 + rows) >> c, :]
    return b


def cascade(layers):
    prev = layers[0]
    for k in range(1, len(layers)):
        out = ov.HostFrame(prev.w, prev.h)
        ov.blank(out)
        ov.crossfade(out, prev, layers[k], FADERS[k - 1])
        prev = out
    return prev


@pytest.mark.parametrize("full,small,world", [((320, 180), (212, 120), 4), ((1920, 1080), (1280, 720), 8), ((640, 360), (640, 180), 3), ((256, 144), (96, 144), 2)])
def test_band_wise_compositing_with_halo_slices_equals_the_unsharded_picture(full, small, world):
    W, H = full
    layers = [ov.HostFrame(W, H).fill(k, seed=4) for k in range(6)] + [ov.HostFrame(*small).fill(k, seed=4) for k in (6, 7)]
    # unsharded: scale the small layers into the picture, cascade, convert
    whole = []
    for f in layers:
        if (f.w, f.h) == (W, H):
            whole.append(f)
        else:
            o = ov.HostFrame(W, H)
            ov.dynamic_scale(f, o)
            whole.append(o)
    want = cascade(whole)
    want_rgba = ov.to_rgba(want, MATRIX)

    got = ov.HostFrame(W, H)
    got_rgba = np.zeros_like(want_rgba)
    for (row0, rows) in shard.row_bands(H, world):
        band_layers = []
        for f in layers:
            if (f.w, f.h) == (W, H):
                band_layers.append(rows_of(f, row0, rows))
                continue
            o = ov.HostFrame(W, rows)
            need = shard.band_source_rows((row0, rows), f.w, f.h, W, H)
            if need is None:                                   # the band lies in the letterbox bars: blank
                ov.blank(o)
            else:
                sl = rows_of(f, need[0], need[1])             # ONLY the halo slice exists on this rank
                assert lib.orc_dynamic_scale_band(C.byref(sl.c), f.h, need[0], C.byref(o.c), W, H, row0) == 0, "the halo slice misses a row the band needs"
            band_layers.append(o)
        b = cascade(band_layers)
        for p in range(3):
            c = 1 if p else 0
            got.planes[p][row0 >> c:(row0 + rows) >> c, :] = b.planes[p]
        got_rgba[row0:row0 + rows] = ov.to_rgba(b, MATRIX)
    for p, (a, b) in enumerate(zip(got.visible(), want.visible())):
        assert np.array_equal(a, b), f"plane {p}: stitched bands differ from the unsharded picture"
    assert np.array_equal(got_rgba, want_rgba)


def test_row_bands_are_whole_chroma_rows_and_cover_the_picture():
    assert shard.row_bands(1080, 8) == [(0, 136), (136, 136), (272, 136), (408, 136), (544, 134), (678, 134), (812, 134), (946, 134)]
    for h, w in [(1080, 8), (720, 7), (2, 1), (350, 4)]:
        b = shard.row_bands(h, w)
        assert b[0][0] == 0 and sum(r for _s, r in b) == h and all(s % 2 == 0 and r % 2 == 0 for s, r in b)
        assert all(b[i][0] + b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
    with pytest.raises(ValueError):
        shard.row_bands(1081, 8)


def test_a_slice_one_chroma_row_short_is_refused():
    # the halo is what the taps reach: drop its last chroma row and the band scaler must say so (not read past the slice)
    W, H, sw, sh = 320, 180, 212, 120
    f = ov.HostFrame(sw, sh).fill(3, seed=1)
    band = shard.row_bands(H, 4)[1]
    need = shard.band_source_rows(band, sw, sh, W, H)
    o = ov.HostFrame(W, band[1])
    sl = rows_of(f, need[0], need[1] - 2)
    assert lib.orc_dynamic_scale_band(C.byref(sl.c), sh, need[0], C.byref(o.c), W, H, band[0]) == -1
