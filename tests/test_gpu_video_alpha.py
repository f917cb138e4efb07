"""Per-pixel alpha composite on the device (BUILD-SPECIFIED, include/mixlab_gpu.h MX_PIXFMT_YUVA420P; DESIGN.md "Per-pixel alpha") against the oracle's rule
(oracle/mixlab_oracle.h orc_video_crossfade; tests/test_cpu_video_alpha.py pins that rule to an independent numpy restatement).  Bit-exact: integer work.
Reference side of the rule: the cross-fade it degenerates to for opaque layers is src/module/video_mixer.rs:168,211-235."""
import numpy as np
import pytest

import alpha_patterns as ap
import oracle_video as ov
from mixlab_amd import abi, video
from mixlab_amd.workspace import Workspace
from test_gpu_video_graph import FADERS, MATRIX, cascade

pytestmark = pytest.mark.gpu

GEOMETRIES = [(64, 36), (66, 38), (322, 182), (1280, 720), (1920, 1080)]


def upload(hf, with_alpha=None):
    """HostFrame -> device frame; a HostFrame with a coverage plane becomes yuva420p"""
    has = hasattr(hf, "alpha") if with_alpha is None else with_alpha
    y, u, v = hf.visible()
    d = video.DFrame(hf.w, hf.h, fmt=video.PIXFMT_YUVA420P if has else video.PIXFMT_YUV420P).upload(y, u, v)
    if has:
        d.upload_alpha(hf.visible_alpha())
    return d


def assert_frame_equal(d, hf, what):
    for p, (x, y) in enumerate(zip(d.download(), hf.visible())):
        bad = np.argwhere(x != y)
        assert bad.size == 0, f"{what}: plane {p} differs at {bad[:4].tolist()} ({len(bad)} samples)"


@pytest.mark.parametrize("size", GEOMETRIES, ids=[f"{w}x{h}" for w, h in GEOMETRIES])
@pytest.mark.parametrize("pattern", ap.PATTERNS)
@pytest.mark.parametrize("who", ["a", "b", "both"])
def test_crossfade_with_coverage_planes_bit_exact(size, pattern, who):
    """The stateless compose step (mx_video_crossfade) over five geometries x three coverage patterns x which layer carries the plane, four faders each.
    Constant 255 must ALSO equal the alpha-free device picture bit for bit."""
    w, h = size
    A = ov.HostFrame(w, h).fill(1, seed=2); B = ov.HostFrame(w, h).fill(6, seed=3)
    if who in ("a", "both"):
        A.set_alpha(ap.alpha_plane(w, h, pattern, 1))
    if who in ("b", "both"):
        B.set_alpha(ap.alpha_plane(w, h, pattern if who == "b" else "random", 2))
    dA, dB = upload(A), upload(B)
    assert dA.fmt == (video.PIXFMT_YUVA420P if who in ("a", "both") else video.PIXFMT_YUV420P)
    if who in ("a", "both"):
        assert np.array_equal(dA.download_alpha(), A.visible_alpha())
    plainA, plainB = upload(A, False), upload(B, False)
    for fader in (0.0, 0.37, 0.9, 1.0):
        want = ov.HostFrame(w, h); ov.blank(want); ov.crossfade(want, A, B, fader)
        out = video.DFrame(w, h)
        video.crossfade(out, dA, dB, fader)
        assert_frame_equal(out, want, f"fader {fader}")
        if pattern == "opaque" and who != "both":
            plain = video.DFrame(w, h)
            video.crossfade(plain, plainA, plainB, fader)
            for x, y in zip(out.download(), plain.download()):
                assert np.array_equal(x, y), "opaque coverage must be today's cross-fade"
    # a missing layer reads the blank plane (video_mixer.rs:180-188), opaque
    want = ov.HostFrame(w, h); ov.blank(want); ov.crossfade(want, A, None, 0.6)
    out = video.DFrame(w, h); video.crossfade(out, dA, None, 0.6)
    assert_frame_equal(out, want, "B missing")


@pytest.mark.parametrize("geom", [((320, 180), (320, 180)), ((212, 120), (320, 180)), ((160, 120), (322, 182)), ((1280, 720), (1920, 1080)), ((640, 360), (320, 180))],
                         ids=["same-size", "upscale", "pillarbox", "720p-to-1080p", "downscale"])
def test_scaler_resamples_the_coverage_plane_like_luma_and_mixer_honours_it(geom):
    """A yuva420p layer smaller (or larger) than the picture: the DynamicScaler's output carries the coverage resampled with the luma taps into the same letterboxed
    rectangle, bars opaque; the VideoMixer then composes with it.  Persistent scaler (twice: ring frames are reused) and VideoMixer against the oracle."""
    (iw, ih), (ow, oh) = geom
    src = ov.HostFrame(iw, ih).fill(3, seed=7).set_alpha(ap.alpha_plane(iw, ih, "soft-disc", 3))
    d = upload(src)
    want = ov.HostFrame(ow, oh).set_alpha(); ov.blank(want); ov.dynamic_scale(src, want)
    sc = video.Scaler(ow, oh)
    for rnd in range(2):
        res = sc.scale(d)
        assert res.has_alpha()
        assert_frame_equal(res, want, f"round {rnd}")
        assert np.array_equal(res.download_alpha(), want.visible_alpha()), f"round {rnd}: coverage plane"
        del res
    other = ov.HostFrame(ow, oh).fill(4, seed=5)
    for (a, b, fader) in ((0, 1, 0.8), (1, 0, 0.35)):
        m = video.VideoMixer(a=a, b=b, fader=fader)
        om = ov.OracleVideoMixer(a=a, b=b, fader=fader)
        prog, _a, _b = m.run_tick(0, [(d, (1, 30), (0, 1)), (upload(other), (1, 30), (0, 1)), None, None])
        want_prog = om.run_tick(0, [(src, (1, 30), (0, 1)), (other, (1, 30), (0, 1)), None, None])
        assert_frame_equal(prog, want_prog, f"VideoMixer a={a} b={b}")
        assert not prog.has_alpha()                      # the composite is opaque yuv420p (video_mixer.rs:282-283)


@pytest.mark.parametrize("fmt", [video.PIXFMT_BGRA, video.PIXFMT_RGBA, video.PIXFMT_ARGB, video.PIXFMT_ABGR], ids=["bgra", "rgba", "argb", "abgr"])
@pytest.mark.parametrize("src,dst", [((320, 180), (480, 270)), ((64, 64), (320, 180)), ((320, 180), (320, 180))], ids=["up-1.5x", "pillarbox", "same-size"])
def test_packed_rgba_a_byte_is_the_layers_coverage(fmt, src, dst):
    rng = np.random.default_rng(src[0] + dst[0] + fmt)
    w, h = src
    pix = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    ai = 0 if fmt in (video.PIXFMT_ARGB, video.PIXFMT_ABGR) else 3
    pix[..., ai] = ap.alpha_plane(w, h, "soft-disc", 1)
    pix[: h // 4, :, ai] = rng.integers(0, 256, size=(h // 4, w), dtype=np.uint8)
    d = video.DFrame(w, h, fmt=fmt).upload_packed(pix)
    as444 = ov.packed_rgb_to_yuv444(pix, fmt)
    assert np.array_equal(as444.visible_alpha(), pix[..., ai])
    other = ov.HostFrame(*dst).fill(3, seed=5)
    for (a, b, fader) in ((0, 1, 1.0), (0, 1, 0.6), (1, 0, 0.25)):
        m = video.VideoMixer(a=a, b=b, fader=fader)
        om = ov.OracleVideoMixer(a=a, b=b, fader=fader)
        prog, _a, _b = m.run_tick(0, [(d, (1, 30), (0, 1)), (upload(other), (1, 30), (0, 1)), None, None])
        want_prog = om.run_tick(0, [(as444, (1, 30), (0, 1)), (other, (1, 30), (0, 1)), None, None])
        assert_frame_equal(prog, want_prog, f"a={a} b={b} fader={fader}")


def _alpha_layers(sizes, which, patterns, seed=3):
    layers = []
    for k, (w, h) in enumerate(sizes):
        hf = ov.HostFrame(w, h).fill(k, seed=seed)
        if k in which:
            hf.set_alpha(ap.alpha_plane(w, h, patterns[k % len(patterns)], k))
        layers.append(hf)
    return layers


def _oracle_cascade(layers, faders, ticks=1):
    oms = [ov.OracleVideoMixer(a=0, b=1, fader=faders[k]) for k in range(len(layers) - 1)]
    want = None
    for tick in range(ticks):
        prev = (layers[0], (1, 60), (0, 1))
        for k in range(len(layers) - 1):
            out = oms[k].run_tick(tick * 735, [prev, (layers[k + 1], (1, 60), (0, 1)), None, None])
            prev = (out, (1, 60), (0, 1))
        want = prev[0]
    return want


@pytest.mark.parametrize("sizes", [[(320, 180)] * 6 + [(212, 120)] * 2, [(1920, 1080)] * 6 + [(1280, 720)] * 2], ids=["320x180", "1080p"])
@pytest.mark.parametrize("which", [(1, 3, 6), (0, 2, 7), tuple(range(8))], ids=["layers-1-3-6", "base-2-7", "all"])
def test_config4_cascade_with_coverage_layers_bit_exact(sizes, which):
    """BASELINE configs[3] ("scale + alpha composite + colour-matrix") as a graph: the 8-layer cascade with a mix of opaque layers and layers that carry coverage --
    among them the BASE layer, a scaled 720p layer, and a layer whose fader rests at 1.0 -- through the fused RGBA sink (k_video_batch's coverage instantiation,
    several ticks per launch) and as the last mixer's YUV program, against the oracle cascade."""
    ws, srcs, mixers, rgba = cascade(sizes, MATRIX)
    g = ws.build(max_ticks_per_run=4)
    layers = _alpha_layers(sizes, which, ("soft-disc", "random", "opaque"))
    dl = [upload(l) for l in layers]
    for s, d in zip(srcs, dl):
        video.graph_set_video_source(g, s, d, dur=(1, 60), off=(0, 1), repeat=True)
    g.run_ticks(0, 3)
    want = _oracle_cascade(layers, FADERS, 3)
    got = video.graph_video_output(g, mixers[-1], 0)
    assert_frame_equal(got, want, "final composite")
    assert np.array_equal(video.graph_rgba_output(g, rgba), ov.to_rgba(want, MATRIX))


def test_cascade_whose_layers_all_carry_opaque_coverage_equals_the_alpha_free_cascade():
    sizes = [(1920, 1080)] * 6 + [(1280, 720)] * 2
    outs = []
    for which in ((), tuple(range(8))):
        ws, srcs, mixers, rgba = cascade(sizes, MATRIX)
        g = ws.build(max_ticks_per_run=2)
        layers = _alpha_layers(sizes, which, ("opaque",))
        keep = [upload(l) for l in layers]
        for s, d in zip(srcs, keep):
            video.graph_set_video_source(g, s, d, dur=(1, 60), off=(0, 1), repeat=True)
        g.run_ticks(0, 2)
        outs.append(video.graph_rgba_output(g, rgba).copy())
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("faders", [[1.0, 0.0, 1.0, 0.5, 0.0, 1.0, 0.3], [0.0, 1.0, 0.0, 0.0, 1.0, 0.0, 1.0]], ids=["resting-mixed", "resting-all"])
def test_faders_resting_at_their_ends_with_coverage_layers(faders):
    """The launcher drops steps whose fader rests at an end of its travel (mx_k_video.hip chain_matrix_mode); with coverage a step at 0.0 no longer returns the other
    layer exactly, and a step at 1.0 over a bare base layer with coverage is not a no-op: both must still be the oracle's picture."""
    sizes = [(322, 182)] * 8
    ws = Workspace(44100, 60)
    srcs = [ws.source_video() for _ in sizes]
    prev, mixers = srcs[0], []
    for k in range(1, 8):
        m = ws.video_mixer(a=0, b=1, fader=faders[k - 1])
        ws.connect(prev, 0, m, 0); ws.connect(srcs[k], 0, m, 1)
        mixers.append(m); prev = m
    rgba = ws.video_to_rgba(MATRIX)
    ws.connect(prev, 0, rgba, 0)
    g = ws.build(max_ticks_per_run=2)
    layers = _alpha_layers(sizes, (0, 1, 2, 4, 5, 7), ("random", "soft-disc"))
    keep = [upload(l) for l in layers]
    for s, d in zip(srcs, keep):
        video.graph_set_video_source(g, s, d, dur=(1, 60), off=(0, 1), repeat=True)
    g.run_ticks(0, 2)
    want = _oracle_cascade(layers, faders, 2)
    assert_frame_equal(video.graph_video_output(g, mixers[-1], 0), want, "final composite")
    assert np.array_equal(video.graph_rgba_output(g, rgba), ov.to_rgba(want, MATRIX))


def test_layer_on_input_a_over_the_running_composite():
    """The other wiring: every new layer enters on input A (it is FADED IN over the composite so far), so the running composite is B."""
    sizes = [(320, 180)] * 5
    faders = [0.4, 0.9, 1.0, 0.15]
    ws = Workspace(44100, 60)
    srcs = [ws.source_video() for _ in sizes]
    prev, mixers = srcs[0], []
    for k in range(1, 5):
        m = ws.video_mixer(a=0, b=1, fader=faders[k - 1])
        ws.connect(srcs[k], 0, m, 0); ws.connect(prev, 0, m, 1)
        mixers.append(m); prev = m
    rgba = ws.video_to_rgba(MATRIX)
    ws.connect(prev, 0, rgba, 0)
    g = ws.build()
    layers = _alpha_layers(sizes, (0, 1, 3, 4), ("soft-disc", "random"))
    keep = [upload(l) for l in layers]
    for s, d in zip(srcs, keep):
        video.graph_set_video_source(g, s, d, dur=(1, 60), off=(0, 1), repeat=True)
    g.run_ticks(0, 1)
    oms = [ov.OracleVideoMixer(a=0, b=1, fader=f) for f in faders]
    prevo = (layers[0], (1, 60), (0, 1))
    for k in range(4):
        out = oms[k].run_tick(0, [(layers[k + 1], (1, 60), (0, 1)), prevo, None, None])
        prevo = (out, (1, 60), (0, 1))
    assert_frame_equal(video.graph_video_output(g, mixers[-1], 0), prevo[0], "final composite")
    assert np.array_equal(video.graph_rgba_output(g, rgba), ov.to_rgba(prevo[0], MATRIX))


@pytest.mark.parametrize("world", [4, 8])
def test_row_bands_of_a_cascade_with_coverage_layers_equal_the_unsharded_picture(world):
    """Row-band sharding (mixlab_amd/shard.py) with coverage: every band gets its rows of every layer AND of the layer's coverage plane; stitched, the bands
    are the unsharded picture.  (Layers of the picture's own size: a band-scaled smaller layer cannot carry coverage and says so.)"""
    from mixlab_amd import shard
    W, H = 1920, 1080
    sizes = [(W, H)] * 8
    layers = _alpha_layers(sizes, (1, 2, 5, 7), ("soft-disc", "random"), seed=4)
    want = _oracle_cascade(layers, FADERS, 1)
    want_rgba = ov.to_rgba(want, MATRIX)
    got_rgba = np.zeros_like(want_rgba)
    for (row0, rows) in shard.row_bands(H, world):
        ws, srcs, mixers, rgba = cascade([(W, rows)] * 8, MATRIX)
        g = ws.build()
        keep = []
        for k, f in enumerate(layers):
            y, u, v = f.visible()
            has = hasattr(f, "alpha")
            d = video.DFrame(W, rows, fmt=video.PIXFMT_YUVA420P if has else video.PIXFMT_YUV420P).upload(y[row0:row0 + rows], u[row0 // 2:(row0 + rows) // 2], v[row0 // 2:(row0 + rows) // 2])
            if has:
                d.upload_alpha(f.visible_alpha()[row0:row0 + rows])
            keep.append(d)
            video.graph_set_video_source(g, srcs[k], d, dur=(1, 60), off=(0, 1), repeat=True)
        g.run_ticks(0, 1)
        got_rgba[row0:row0 + rows] = video.graph_rgba_output(g, rgba)
    assert np.array_equal(got_rgba, want_rgba)
    # a band-scaled layer with coverage is refused, not composited wrong
    whole = video.DFrame(1280, 720, fmt=video.PIXFMT_YUVA420P)
    with pytest.raises(abi.MxError, match="coverage"):
        video.scale_band(whole, 720, 0, video.DFrame(W, 136), W, H, 0)
    video.scale_band(video.DFrame(1280, 720), 720, 0, video.DFrame(W, 136), W, H, 0)      # the same call without the plane is fine


def test_alpha_entry_points_reject_frames_without_a_plane():
    d = video.DFrame(64, 36)
    assert not d.has_alpha()
    with pytest.raises(abi.MxError):
        d.upload_alpha(np.zeros((36, 64), np.uint8))
    with pytest.raises(abi.MxError):
        d.download_alpha()
    fresh = video.DFrame(64, 36, fmt=video.PIXFMT_YUVA420P)
    assert fresh.has_alpha() and (fresh.download_alpha() == 255).all()   # blank = opaque
    y, u, v = fresh.download()
    assert not y.any() and (u == 0x80).all() and (v == 0x80).all()
