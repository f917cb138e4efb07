"""Per-pixel alpha composite (BUILD-SPECIFIED: the reference's only "alpha" is the VideoMixer's global fader, src/module/video_mixer.rs:168,211-235): the oracle's
rule against an independent numpy restatement, and the properties the specification promises -- opaque coverage IS the reference's cross-fade, a transparent sample
of one layer hands its share to the other, the fader scales coverage by the reference's own truncation rule.  No GPU."""
import numpy as np
import pytest

import alpha_patterns as ap
import oracle_video as ov


def _frames(w, h, seed):
    return ov.HostFrame(w, h).fill(1, seed=seed), ov.HostFrame(w, h).fill(5, seed=seed + 1)


@pytest.mark.parametrize("size", [(64, 36), (66, 38), (322, 182), (34, 2)])
@pytest.mark.parametrize("fader", [0.0, 0.1, 0.5, 0.999, 1.0])
@pytest.mark.parametrize("who", ["a", "b", "both"])
def test_oracle_alpha_crossfade_equals_the_numpy_restatement(size, fader, who):
    w, h = size
    A, B = _frames(w, h, 3)
    aa = ap.alpha_plane(w, h, "random", 1) if who in ("a", "both") else None
    ab = ap.alpha_plane(w, h, "soft-disc", 2) if who in ("b", "both") else None
    if aa is not None:
        A.set_alpha(aa)
    if ab is not None:
        B.set_alpha(ab)
    out = ov.HostFrame(w, h); ov.blank(out)
    ov.crossfade(out, A, B, fader)
    fade = ov.lib.orc_crossfade_factor(fader)
    for p, (o, a, b) in enumerate(zip(out.visible(), A.visible(), B.visible())):
        c = 1 if p else 0
        assert np.array_equal(o, ap.crossfade_alpha_numpy(a, b, aa, ab, fade, c, c)), f"plane {p}"


@pytest.mark.parametrize("fader", [0.0, 0.3, 0.75, 1.0])
def test_opaque_coverage_is_the_reference_crossfade_bit_for_bit(fader):
    w, h = 130, 74
    A, B = _frames(w, h, 9)
    want = ov.HostFrame(w, h); ov.blank(want); ov.crossfade(want, A, B, fader)          # fade_line, video_mixer.rs:211-235
    for who in ("a", "b", "both"):
        A2, B2 = _frames(w, h, 9)
        if who in ("a", "both"):
            A2.set_alpha()
        if who in ("b", "both"):
            B2.set_alpha()
        got = ov.HostFrame(w, h); ov.blank(got); ov.crossfade(got, A2, B2, fader)
        for x, y in zip(got.planes, want.planes):
            assert np.array_equal(x, y), who


def test_a_transparent_sample_hands_its_share_to_the_other_layer():
    w, h = 64, 36
    A, B = _frames(w, h, 4)
    A.set_alpha(np.zeros((h, w), np.uint8))
    out = ov.HostFrame(w, h); ov.blank(out); ov.crossfade(out, A, B, 1.0)                # fader full on A, A fully transparent: B
    for o, b in zip(out.visible(), B.visible()):
        assert np.array_equal(o, b)
    A2, B2 = _frames(w, h, 4)
    B2.set_alpha(np.zeros((h, w), np.uint8))
    out = ov.HostFrame(w, h); ov.blank(out); ov.crossfade(out, A2, B2, 0.0)              # fader full on B, B fully transparent: A
    for o, a in zip(out.visible(), A2.visible()):
        assert np.array_equal(o, a)


def test_scaled_coverage_plane_follows_the_luma_geometry_and_bars_are_opaque():
    src = ov.HostFrame(160, 120).fill(2, seed=1).set_alpha(ap.alpha_plane(160, 120, "soft-disc"))
    dst = ov.HostFrame(320, 180).set_alpha(np.zeros((180, 320), np.uint8))
    ov.dynamic_scale(src, dst)
    sw, sh, lx, ly = ov.scaler_geometry(160, 120, 320, 180)
    al = dst.visible_alpha()
    assert (al[:, :lx] == 255).all() and (al[:, lx + sw:] == 255).all()                  # pillar bars
    # the coverage inside the picture is the luma scaler's output of the coverage plane
    as_luma = ov.HostFrame(160, 120); as_luma.planes[0][:, :160] = src.visible_alpha()
    want = ov.HostFrame(320, 180); ov.dynamic_scale(as_luma, want)
    assert np.array_equal(al[ly:ly + sh, lx:lx + sw], want.visible()[0][ly:ly + sh, lx:lx + sw])


def test_packed_rgba_carries_its_a_byte_as_coverage():
    rng = np.random.default_rng(5)
    pix = rng.integers(0, 256, size=(18, 32, 4), dtype=np.uint8)
    for fmt, ai in ((5, 3), (24, 3), (25, 0), (26, 0)):
        f = ov.packed_rgb_to_yuv444(pix, fmt)
        assert np.array_equal(f.visible_alpha(), pix[..., ai])
    assert not hasattr(ov.packed_rgb_to_yuv444(pix[..., :3], 4), "alpha")
