"""Video nodes inside the graph executor: SURVEY.md section 8d config 4 -- an 8-layer composite expressed in
reference semantics as a cascade of 7 VideoMixer cross-fades (each truncating to u8), then the
build-specified YUV420P->RGBA + colour matrix.  Bit-exact vs the oracle."""
import numpy as np
import pytest

import oracle_video as ov
from mixlab_amd import abi, video
from mixlab_amd.workspace import Workspace

pytestmark = pytest.mark.gpu

FADERS = [1.0, 0.75, 0.5, 0.5, 0.25, 0.9, 0.1]
MATRIX = [3900, 150, 46, 4096, 60, 3980, 56, -2048, 20, 120, 3956, 0]   # Q12 3x4, mild cross-talk + offsets


def cascade(sizes, matrix):
    ws = Workspace(44100, 60)
    srcs = [ws.source_video() for _ in sizes]
    prev, mixers = srcs[0], []
    for k in range(1, len(sizes)):
        m = ws.video_mixer(a=0, b=1, fader=FADERS[k - 1])
        ws.connect(prev, 0, m, 0)      # A = running composite (or layer 0)
        ws.connect(srcs[k], 0, m, 1)   # B = next layer
        mixers.append(m); prev = m
    rgba = ws.video_to_rgba(matrix)
    ws.connect(prev, 0, rgba, 0)
    return ws, srcs, mixers, rgba


def upload(hf):
    y, u, v = hf.visible()
    return video.DFrame(hf.w, hf.h).upload(y, u, v)


MIXED = [(320, 180), (160, 120), (320, 100), (100, 180), (212, 120), (318, 178), (2, 2), (64, 64)]   # pillar / letter boxes, 7 scaled layers (> MX_CHAIN_MAX_SCALED)
GROWING = [(64, 64), (128, 72), (130, 40), (322, 182), (322, 182), (200, 182), (322, 100), (640, 360)]   # the running composite itself is rescaled; the last layer is DOWNscaled


@pytest.mark.parametrize("sizes", [[(320, 180)] * 6 + [(212, 120)] * 2, [(1920, 1080)] * 6 + [(1280, 720)] * 2, MIXED, GROWING,
                                   [(1920, 1080), (1440, 1080), (1920, 800), (1280, 720), (1918, 1078), (960, 1080), (1920, 1080), (1000, 1000)]],
                         ids=["180p", "1080p", "mixed-boxes", "growing", "1080p-boxes"])
@pytest.mark.parametrize("inline", ["0", "1"], ids=["scaler-kernel", "resampled-in-chain"])
def test_config4_eight_layer_cascade_bit_exact(sizes, inline, monkeypatch):
    monkeypatch.setenv("MX_SCALE_INLINE", inline)   # layers resampled by the stand-alone scaler (default) / inside the RGBA chain kernel
    ws, srcs, mixers, rgba = cascade(sizes, MATRIX)
    g = ws.build(max_ticks_per_run=4)
    layers = [ov.HostFrame(w, h).fill(k, seed=3) for k, (w, h) in enumerate(sizes)]
    dlayers = [upload(l) for l in layers]
    for s, d in zip(srcs, dlayers):
        video.graph_set_video_source(g, s, d, dur=(1, 60), off=(0, 1), repeat=True)
    g.run_ticks(0, 3)   # every layer delivers a new frame on every tick
    # oracle: the same cascade of reference VideoMixers, three ticks
    oms = [ov.OracleVideoMixer(a=0, b=1, fader=FADERS[k]) for k in range(7)]
    want = None
    for tick in range(3):
        prev = (layers[0], (1, 60), (0, 1))
        for k in range(7):
            out = oms[k].run_tick(tick * 735, [prev, (layers[k + 1], (1, 60), (0, 1)), None, None])
            prev = (out, (1, 60), (0, 1))
        want = prev[0]
    got = video.graph_video_output(g, mixers[-1], 0)
    assert (got.width, got.height) == (want.w, want.h)
    for p, (a, b) in enumerate(zip(got.download(), want.visible())):
        assert np.array_equal(a, b), f"plane {p} of the final composite differs"
    assert np.array_equal(video.graph_rgba_output(g, rgba), ov.to_rgba(want, MATRIX))


@pytest.mark.parametrize("inline", ["0", "1"], ids=["scaler-kernel", "resampled-in-chain"])
def test_scaled_layers_held_over_several_ticks_and_replaced(inline, monkeypatch):
    """A layer that the chain kernel may resample itself stays an unevaluated scaler output while it is the channel's stored frame:
    ticks without a new input frame re-use it (video_mixer.rs:94-101,122-148), a new frame of another size re-targets the scaler
    (encode.rs:347-384).  Both ways of computing it must be the oracle's picture."""
    monkeypatch.setenv("MX_SCALE_INLINE", inline)
    sizes = [(320, 180), (160, 120), (212, 120), (320, 180)]
    ws, srcs, mixers, rgba = cascade(sizes, MATRIX)
    g = ws.build()
    oms = [ov.OracleVideoMixer(a=0, b=1, fader=FADERS[k]) for k in range(len(sizes) - 1)]
    # tick -> {layer: (size, seed)}: layers live 3 ticks; layer 1 is replaced by another size on tick 2, everything expires by tick 6
    plan = {0: {0: ((320, 180), 1), 1: ((160, 120), 2), 2: ((212, 120), 3), 3: ((320, 180), 4)},
            2: {1: ((100, 180), 5)}, 3: {0: ((320, 180), 6), 2: ((212, 120), 7)}, 4: {3: ((300, 100), 8)}}
    keep = []
    for tick in range(7):
        new = {}
        for k, (size, seed) in plan.get(tick, {}).items():
            hf = ov.HostFrame(*size).fill(k, seed=seed)
            d = upload(hf); keep.append(d)
            video.graph_set_video_source(g, srcs[k], d, dur=(3, 60), off=(0, 1), repeat=False)
            new[k] = hf
        g.run_ticks(tick, 1)
        prev = (new[0], (3, 60), (0, 1)) if 0 in new else None
        for k in range(len(sizes) - 1):
            b = (new[k + 1], (3, 60), (0, 1)) if (k + 1) in new else None
            out = oms[k].run_tick(tick * 735, [prev, b, None, None])
            prev = (out, (1, 60), (0, 1)) if out is not None else None
        want = prev[0] if prev else None
        got = video.graph_rgba_output(g, rgba)
        if want is None:
            assert got is None or got.size == 0, f"tick {tick}"
            continue
        assert np.array_equal(got, ov.to_rgba(want, MATRIX)), f"tick {tick}: RGBA differs"
        prog = video.graph_video_output(g, mixers[-1], 0)
        for p, (a, b) in enumerate(zip(prog.download(), want.visible())):
            assert np.array_equal(a, b), f"tick {tick} plane {p}"


@pytest.mark.parametrize("inline", ["0", "1"], ids=["scaler-kernel", "resampled-in-chain"])
def test_mixer_inputs_of_other_pixel_formats(inline, monkeypatch):
    """Layers arrive as yuv422p / yuv444p (decoders produce them): each VideoMixer channel's scaler converts while it fits the layer into
    the yuv420p output picture -- also when the layer already has the output's size (encode.rs:342-352 compares the whole settings)."""
    monkeypatch.setenv("MX_SCALE_INLINE", inline)
    spec = [((320, 180), 0), ((320, 180), 2), ((160, 90), 1), ((320, 180), 3), ((212, 120), 3), ((640, 360), 2), ((640, 360), 3)]
    ws, srcs, mixers, rgba = cascade([s for s, _ in spec], MATRIX)
    g = ws.build(max_ticks_per_run=4)
    layers = [ov.HostFrame(w, h, fmt).fill(k, seed=6) for k, ((w, h), fmt) in enumerate(spec)]
    keep = []
    for s, hf in zip(srcs, layers):
        d = video.DFrame(hf.w, hf.h, fmt=hf.fmt).upload(*hf.visible()); keep.append(d)
        video.graph_set_video_source(g, s, d, dur=(1, 60), off=(0, 1), repeat=True)
    g.run_ticks(0, 2)
    oms = [ov.OracleVideoMixer(a=0, b=1, fader=FADERS[k]) for k in range(len(spec) - 1)]
    want = None
    for tick in range(2):
        prev = (layers[0], (1, 60), (0, 1))
        for k in range(len(spec) - 1):
            out = oms[k].run_tick(tick * 735, [prev, (layers[k + 1], (1, 60), (0, 1)), None, None])
            prev = (out, (1, 60), (0, 1))
        want = prev[0]
    got = video.graph_video_output(g, mixers[-1], 0)
    assert (got.width, got.height, got.fmt) == (want.w, want.h, 0)
    for p, (a, b) in enumerate(zip(got.download(), want.visible())):
        assert np.array_equal(a, b), f"plane {p} differs"
    assert np.array_equal(video.graph_rgba_output(g, rgba), ov.to_rgba(want, MATRIX))


def test_video_source_single_shot_then_none_and_passthrough():
    ws = Workspace(44100, 60)
    s = ws.source_video(); m = ws.video_mixer(a=0, b=None, fader=1.0)
    ws.connect(s, 0, m, 0)
    g = ws.build()
    hf = ov.HostFrame(64, 64).fill(2)
    d = upload(hf)
    assert video.graph_video_output(g, m, 0) is None
    video.graph_set_video_source(g, s, d, dur=(1, 30), off=(0, 1), repeat=False)
    seen = []
    for tick in range(4):
        g.run_ticks(tick, 1)
        prog = video.graph_video_output(g, m, 0)
        a = video.graph_video_output(g, m, 1)
        seen.append((prog is not None, a is not None))
    # the frame arrives on tick 0 only (A pass-through that tick), is shown for its 1/30 s = 2 ticks, then expires
    assert seen == [(True, True), (True, False), (False, False), (False, False)]


def test_video_edge_type_is_checked():
    ws = Workspace()
    s = ws.source_video(); e = ws.eq_three(0, 0, 0)
    ws.connect(s, 0, e, 0)
    with pytest.raises(abi.MxError) as ei:
        ws.build()
    assert ei.value.code == abi.MX_ERR_TYPE


def test_video_source_ring_delivers_a_new_frame_every_tick_cycling():
    ws = Workspace(44100, 60)
    s = ws.source_video(); m = ws.video_mixer(a=0, b=None, fader=1.0)
    ws.connect(s, 0, m, 0)
    g = ws.build(max_ticks_per_run=4)
    hosts = [ov.HostFrame(64, 48).fill(k, seed=9) for k in range(3)]
    ring = [upload(h) for h in hosts]
    video.graph_set_video_source_ring(g, s, ring, dur=(1, 60), off=(0, 1))
    for tick in range(7):          # one tick per run: the program output of tick k is frame k mod 3 (fader 1.0 = all A)
        g.run_ticks(tick, 1)
        got = video.graph_video_output(g, m, 0)
        for a, b in zip(got.download(), hosts[tick % 3].visible()):
            assert np.array_equal(a, b), f"tick {tick}"
    g.run_ticks(7, 4)              # a batched run advances the ring four times: ticks 7..10 -> frames 1, 2, 0, 1
    for a, b in zip(video.graph_video_output(g, m, 0).download(), hosts[10 % 3].visible()):
        assert np.array_equal(a, b)
    video.graph_set_video_source_ring(g, s, [], dur=(1, 60), off=(0, 1))   # cleared: the stored frame expires, then None
    g.run_ticks(11, 2)
    assert video.graph_video_output(g, m, 0) is None


@pytest.mark.parametrize("full,small,world", [((320, 180), (212, 120), 4), ((1920, 1080), (1280, 720), 8)])
def test_row_band_compositing_on_the_device_equals_the_unsharded_picture(full, small, world):
    """What the ranks of a row-band sharded job run (mixlab_amd/shard.py), all on one GPU here: every band gets its own rows of the
    same-size layers and only the halo slice of the scaled layers (mx_video_scale_band), runs the same VideoMixer cascade + RGBA
    sink on band-sized frames; stitched, the bands must be the unsharded oracle picture bit for bit."""
    from mixlab_amd import shard
    from test_cpu_video_bands import rows_of, cascade as oracle_cascade
    W, H = full
    layers = [ov.HostFrame(W, H).fill(k, seed=4) for k in range(6)] + [ov.HostFrame(*small).fill(k, seed=4) for k in (6, 7)]
    whole = []
    for f in layers:
        o = f
        if (f.w, f.h) != (W, H):
            o = ov.HostFrame(W, H); ov.dynamic_scale(f, o)
        whole.append(o)
    want = oracle_cascade(whole)
    want_rgba = ov.to_rgba(want, MATRIX)
    got_rgba = np.zeros_like(want_rgba)
    got_planes = [np.zeros_like(p) for p in want.visible()]
    for (row0, rows) in shard.row_bands(H, world):
        ws, srcs, mixers, rgba = cascade([(W, rows)] * 8, MATRIX)
        g = ws.build()
        keep = []
        for k, f in enumerate(layers):
            if (f.w, f.h) == (W, H):
                d = upload(rows_of(f, row0, rows))
            else:
                d = video.DFrame(W, rows)
                need = shard.band_source_rows((row0, rows), f.w, f.h, W, H)
                if need is not None:
                    sl = upload(rows_of(f, need[0], need[1]))
                    video.scale_band(sl, f.h, need[0], d, W, H, row0)
                    keep.append(sl)
            keep.append(d)
            video.graph_set_video_source(g, srcs[k], d, dur=(1, 60), off=(0, 1), repeat=True)
        g.run_ticks(0, 1)
        band = video.graph_video_output(g, mixers[-1], 0)
        for p, a in enumerate(band.download()):
            c = 1 if p else 0
            got_planes[p][row0 >> c:(row0 + rows) >> c, :] = a
        got_rgba[row0:row0 + rows] = video.graph_rgba_output(g, rgba)
    for p, (a, b) in enumerate(zip(got_planes, want.visible())):
        assert np.array_equal(a, b), f"plane {p}: stitched device bands differ from the unsharded picture"
    assert np.array_equal(got_rgba, want_rgba)
    # a slice without its halo is refused, not read past
    f = layers[6]
    band = shard.row_bands(H, world)[1]
    need = shard.band_source_rows(band, f.w, f.h, W, H)
    with pytest.raises(abi.MxError):
        video.scale_band(upload(rows_of(f, need[0] + 2, need[1] - 2)), f.h, need[0] + 2, video.DFrame(W, band[1]), W, H, band[0])


def test_row_band_of_a_strongly_upscaled_layer_reads_only_its_slice():
    """200x212 into 1920x1080 over 8 bands: a 128 x 32 output tile needs ~11 source rows, the tiled scaler's staging slots load 48 -- rows
    that do not exist behind a band's halo slice must be clamped away, not read (found by tools/stress_bands.py as a GPU memory fault)."""
    from mixlab_amd import shard
    from test_cpu_video_bands import rows_of
    W, H, world = 1920, 1080, 8
    layer = ov.HostFrame(200, 212).fill(22, seed=22)
    want = ov.HostFrame(W, H); ov.dynamic_scale(layer, want)
    got = [np.zeros_like(p) for p in want.visible()]
    for (row0, rows) in shard.row_bands(H, world):
        need = shard.band_source_rows((row0, rows), layer.w, layer.h, W, H)
        d = video.DFrame(W, rows)
        video.scale_band(upload(rows_of(layer, need[0], need[1])), layer.h, need[0], d, W, H, row0)
        for p, a in enumerate(d.download()):
            c = 1 if p else 0
            got[p][row0 >> c:(row0 + rows) >> c, :] = a
    for p, (a, b) in enumerate(zip(got, want.visible())):
        assert np.array_equal(a, b), f"plane {p}"


@pytest.mark.parametrize("full,small,world", [((320, 180), (212, 120), 4), ((1920, 1080), (1280, 720), 8)])
def test_row_band_job_with_band_scaling_sources_in_one_submission(full, small, world):
    """The sharded job as a rank runs it: the smaller layers enter the graph as halo slices, their SOURCE nodes scale them to the band
    every tick (mx_graph_set_video_source_band) inside a multi-tick submission; layers change every tick (rings of 2).  Stitched, the
    last tick's bands are the unsharded oracle picture of that tick."""
    from mixlab_amd import shard
    from test_cpu_video_bands import rows_of, cascade as oracle_cascade
    W, H = full
    n_ticks = 3
    sets = [[ov.HostFrame(W, H).fill(k, seed=4 + r) for k in range(6)] + [ov.HostFrame(*small).fill(k, seed=4 + r) for k in (6, 7)] for r in range(2)]
    last = sets[(n_ticks - 1) % 2]
    whole = []
    for f in last:
        o = f
        if (f.w, f.h) != (W, H):
            o = ov.HostFrame(W, H); ov.dynamic_scale(f, o)
        whole.append(o)
    want_rgba = ov.to_rgba(oracle_cascade(whole), MATRIX)
    got_rgba = np.zeros_like(want_rgba)
    for (row0, rows) in shard.row_bands(H, world):
        ws, srcs, mixers, rgba = cascade([(W, rows)] * 8, MATRIX)
        g = ws.build(max_ticks_per_run=4)
        keep = []
        for k in range(8):
            ring = []
            for r in range(2):
                f = sets[r][k]
                if (f.w, f.h) == (W, H):
                    ring.append(upload(rows_of(f, row0, rows)))
                else:
                    need = shard.band_source_rows((row0, rows), f.w, f.h, W, H)
                    ring.append(upload(rows_of(f, need[0], need[1])))
            if (sets[0][k].w, sets[0][k].h) != (W, H):
                f = sets[0][k]
                need = shard.band_source_rows((row0, rows), f.w, f.h, W, H)
                video.graph_set_video_source_band(g, srcs[k], f.w, f.h, need[0], need[1], W, H, row0, rows)
            keep.append(ring)
            video.graph_set_video_source_ring(g, srcs[k], ring, dur=(1, 60), off=(0, 1))
        g.run_ticks(0, n_ticks)
        got_rgba[row0:row0 + rows] = video.graph_rgba_output(g, rgba)
    assert np.array_equal(got_rgba, want_rgba)


def test_two_rgba_sinks_missing_frames_and_a_size_change_inside_one_submission():
    """The RGBA sink of a batched run is launched one tick late, together with the next tick's scaler tiles.  Two sinks on
    two cascades, a tick on which one cascade has no frame at all, and a sink whose picture grows mid-run (its buffer is reallocated
    with a chain still pending): after the run each sink holds what the in-order schedule gives -- its last tick's picture."""
    ws = Workspace(44100, 60)
    sa, sb, sc = ws.source_video(), ws.source_video(), ws.source_video()
    m1 = ws.video_mixer(a=0, b=1, fader=0.6); ws.connect(sa, 0, m1, 0); ws.connect(sb, 0, m1, 1)
    m2 = ws.video_mixer(a=0, b=1, fader=0.3); ws.connect(sc, 0, m2, 0); ws.connect(sb, 0, m2, 1)
    r1, r2 = ws.video_to_rgba(MATRIX), ws.video_to_rgba(None)
    ws.connect(m1, 0, r1, 0); ws.connect(m2, 0, r2, 0)
    g = ws.build(max_ticks_per_run=8)
    o1, o2 = ov.OracleVideoMixer(a=0, b=1, fader=0.6), ov.OracleVideoMixer(a=0, b=1, fader=0.3)
    big = [ov.HostFrame(320, 180).fill(k, seed=21) for k in range(4)]
    small = [ov.HostFrame(212, 120).fill(k, seed=22) for k in range(4)]
    huge = [ov.HostFrame(640, 360).fill(k, seed=23) for k in range(2)]
    keep = [upload(f) for f in big + small + huge]
    dbig, dsmall, dhuge = keep[:4], keep[4:8], keep[8:]
    # sa: a new 320x180 frame every tick; sb: a 212x120 frame every tick (scaled into both cascades); sc: short-lived frames that
    # leave gaps (cascade 2 has no picture on some ticks), and a 640x360 one from tick 5 on (sink 2's buffer grows)
    video.graph_set_video_source_ring(g, sa, dbig, dur=(1, 60), off=(0, 1))
    video.graph_set_video_source_ring(g, sb, dsmall, dur=(1, 60), off=(0, 1))
    T = 8
    want1 = want2 = None
    # sc is driven per run segment: ticks 0-1 small single shots, 2-4 nothing (sb alone keeps cascade 2 alive), 5-7 the huge ring
    def oracle_tick(t, c_frame):
        nonlocal want1, want2
        a, b = big[t % 4], small[t % 4]
        out1 = o1.run_tick(t * 735, [(a, (1, 60), (0, 1)), (b, (1, 60), (0, 1)), None, None])
        out2 = o2.run_tick(t * 735, [(c_frame, (1, 60), (0, 1)) if c_frame is not None else None, (b, (1, 60), (0, 1)), None, None])
        want1, want2 = out1, out2
    video.graph_set_video_source(g, sc, dsmall[0], dur=(1, 60), off=(0, 1), repeat=True)
    g.run_ticks(0, 2)
    for t in (0, 1):
        oracle_tick(t, small[0])
    video.graph_set_video_source(g, sc, None, dur=(1, 60), off=(0, 1), repeat=False)
    g.run_ticks(2, 3)
    for t in (2, 3, 4):
        oracle_tick(t, None)
    assert np.array_equal(video.graph_rgba_output(g, r2), ov.to_rgba(want2, None))
    video.graph_set_video_source_ring(g, sc, dhuge, dur=(1, 60), off=(0, 1))
    g.run_ticks(5, 3)
    for t in (5, 6, 7):
        oracle_tick(t, huge[(t - 5) % 2])
    got1, got2 = video.graph_rgba_output(g, r1), video.graph_rgba_output(g, r2)
    assert got2.shape[:2] == (360, 640)
    assert np.array_equal(got1, ov.to_rgba(want1, MATRIX))
    assert np.array_equal(got2, ov.to_rgba(want2, None))


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("inline", ["0", "1"], ids=["scaler-kernel", "resampled-in-chain"])
def test_random_cascade_scenarios_in_random_batches(seed, inline, monkeypatch):
    """Seeded scenarios over a 4-layer cascade: every layer gets frames of random size / pixel format / life time at random ticks (with
    gaps, re-targets and downscales), the graph is run in random batches of 1-4 ticks, and at the end of every batch the RGBA sink and
    the program frame must be the oracle cascade's of that tick (stored frames, expiry, lazy frames, alternating scaler outputs and the
    deferred sink launch all interact here)."""
    monkeypatch.setenv("MX_SCALE_INLINE", inline)
    rng = np.random.default_rng(1000 + seed)
    n_layers, n_ticks = 4, 18
    sizes = [(320, 180), (212, 120), (160, 120), (320, 100), (100, 180), (640, 360), (322, 182)]
    ws, srcs, mixers, rgba = cascade([(320, 180)] * n_layers, None)
    g = ws.build(max_ticks_per_run=4)
    oms = [ov.OracleVideoMixer(a=0, b=1, fader=FADERS[k]) for k in range(n_layers - 1)]
    # plan[tick][layer] = HostFrame or absent; the first tick always has layer 0
    plan = []
    for t in range(n_ticks):
        row = {}
        for k in range(n_layers):
            if (t == 0 and k == 0) or rng.random() < 0.45:
                w, h = sizes[int(rng.integers(len(sizes)))]
                fmt = int(rng.integers(4)) if (w % 2 == 0 and h % 2 == 0) else 0
                row[k] = (ov.HostFrame(w, h, fmt).fill(k, seed=int(rng.integers(1 << 12))), int(rng.integers(1, 4)))
        plan.append(row)
    keep = []
    t = 0
    while t < n_ticks:
        batch = int(min(rng.integers(1, 5), n_ticks - t))
        # a batch delivers at most one new frame per layer (on its first tick): later plan rows inside the batch are moved to its start
        new = {}
        for tt in range(t, t + batch):
            for k, v in plan[tt].items():
                new.setdefault(k, v)
        for k in range(n_layers):
            if k in new:
                hf, life = new[k]
                d = video.DFrame(hf.w, hf.h, fmt=hf.fmt).upload(*hf.visible()); keep.append(d)
                video.graph_set_video_source(g, srcs[k], d, dur=(life, 60), off=(0, 1), repeat=False)
            else:
                video.graph_set_video_source(g, srcs[k], None, dur=(1, 60), off=(0, 1), repeat=False)
        g.run_ticks(t, batch)
        want = None
        for tt in range(t, t + batch):
            first = tt == t
            prev = (new[0][0], (new[0][1], 60), (0, 1)) if (first and 0 in new) else None
            for k in range(n_layers - 1):
                b = (new[k + 1][0], (new[k + 1][1], 60), (0, 1)) if (first and (k + 1) in new) else None
                out = oms[k].run_tick(tt * 735, [prev, b, None, None])
                prev = (out, (1, 60), (0, 1)) if out is not None else None
            want = prev[0] if prev else None
        got = video.graph_rgba_output(g, rgba)
        if want is None:
            assert got is None, f"tick {t + batch - 1}: a picture where the oracle has none"
        else:
            assert got is not None and np.array_equal(got, ov.to_rgba(want, None)), f"batch ending at tick {t + batch - 1}: RGBA differs"
            prog = video.graph_video_output(g, mixers[-1], 0)
            for p, (a, b) in enumerate(zip(prog.download(), want.visible())):
                assert np.array_equal(a, b), f"batch ending at tick {t + batch - 1}: plane {p}"
        t += batch


@pytest.mark.parametrize("size", [(1920, 1080), (322, 182), (66, 38)], ids=["1080p", "322x182", "66x38"])
@pytest.mark.parametrize("matrix", [MATRIX, [4096, 0, 0, 0, 0, 4096, 0, 0, 0, 0, 4096, 0], [-3000, 7000, 300, -90000, 32639, -32639, 127, 4096, -128, 129, -129, 250000]],
                         ids=["bench", "identity", "extremes"])
def test_colour_matrix_on_the_matrix_cores_is_bit_exact(size, matrix, monkeypatch):
    """MX_VIDEO_MFMA_MATRIX=1 (an experiment kept as an opt-in, DESIGN.md "Colour matrix on the matrix cores"): the Q12 3 x 4 matrix of the RGBA sink evaluated with
    v_mfma_i32_4x4x4_16b_i8 -- each lane's own pixel as the B column of its four-lane block, the coefficient rows (split into high and low bytes) as A, the constant in
    the accumulator.  Integer arithmetic: the picture must be the oracle's, bit for bit, also with negative coefficients, coefficients at the i8 x 256 limit, large
    constants, and widths that leave lanes outside the picture."""
    monkeypatch.setenv("MX_VIDEO_MFMA_MATRIX", "1")
    sizes = [size] * 4
    ws, srcs, mixers, rgba = cascade(sizes, matrix)
    g = ws.build(max_ticks_per_run=3)
    layers = [ov.HostFrame(*size).fill(k, seed=11) for k in range(4)]
    keep = [upload(l) for l in layers]
    for s, d in zip(srcs, keep):
        video.graph_set_video_source(g, s, d, dur=(1, 60), off=(0, 1), repeat=True)
    g.run_ticks(0, 3)
    oms = [ov.OracleVideoMixer(a=0, b=1, fader=FADERS[k]) for k in range(3)]
    for tick in range(3):
        prev = (layers[0], (1, 60), (0, 1))
        for k in range(3):
            prev = (oms[k].run_tick(tick * 735, [prev, (layers[k + 1], (1, 60), (0, 1)), None, None]), (1, 60), (0, 1))
    assert np.array_equal(video.graph_rgba_output(g, rgba), ov.to_rgba(prev[0], matrix))


def test_a_callers_stream_can_be_retired_and_its_handle_reused():
    """mx_stream_retired: what the batched video launches keep per (device, stream) -- descriptor slots whose content is compared with the next launch's --
    is dropped when the host says its stream is gone; a graph that later runs on a stream with the same handle starts from an empty ring and composes the right picture."""
    import ctypes as C
    with pytest.raises(abi.MxError):
        video.stream_retired(None)
    hip = C.CDLL("libamdhip64.so")                 # the runtime the library itself is linked against (importing torch here would bring a second one)
    handle = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(handle)) == 0 and handle.value

    class _St:                                     # a caller-owned stream
        cuda_stream = handle.value

        @staticmethod
        def synchronize():
            assert hip.hipStreamSynchronize(handle) == 0
    st = _St()
    sizes = [(320, 180)] * 3 + [(212, 120)]
    for rnd in range(2):
        ws, srcs, mixers, rgba = cascade(sizes, MATRIX)
        g = ws.build(max_ticks_per_run=4, stream=st.cuda_stream)
        layers = [ov.HostFrame(w, h).fill(k, seed=20 + rnd) for k, (w, h) in enumerate(sizes)]
        keep = [upload(l) for l in layers]
        for s, d in zip(srcs, keep):
            video.graph_set_video_source(g, s, d, dur=(1, 60), off=(0, 1), repeat=True)
        g.run_ticks(0, 4)
        oms = [ov.OracleVideoMixer(a=0, b=1, fader=FADERS[k]) for k in range(3)]
        for tick in range(4):
            prev = (layers[0], (1, 60), (0, 1))
            for k in range(3):
                prev = (oms[k].run_tick(tick * 735, [prev, (layers[k + 1], (1, 60), (0, 1)), None, None]), (1, 60), (0, 1))
        assert np.array_equal(video.graph_rgba_output(g, rgba), ov.to_rgba(prev[0], MATRIX)), f"round {rnd}"
        g.close(); del keep
        st.synchronize()
        video.stream_retired(st.cuda_stream)       # twice in the second round is harmless too
    video.stream_retired(st.cuda_stream)
    assert hip.hipStreamDestroy(handle) == 0
