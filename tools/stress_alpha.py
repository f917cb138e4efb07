"""Stress of the per-pixel alpha composite (row x5, build-specified): cascades of 2 - 8 layers of random sizes (full-size, pillar- and letter-boxed, up- and downscaled),
a random subset of them carrying coverage planes of random patterns (opaque / every byte value / soft disc), random faders with ends of travel mixed in, new layers entering
on input A or B, 1 - 5 ticks per run and one or two runs, the fused RGBA sink (k_video_batch's coverage instantiation) and the last mixer's YUV program, both scaler forms
(MX_SCALE_INLINE); every picture against the oracle cascade, bit for bit.   python tools/stress_alpha.py [first_seed] [count]"""
import os, sys, pathlib, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import alpha_patterns as ap
import oracle_video as ov
from mixlab_amd import video
from mixlab_amd.workspace import Workspace

MATRIX = [4096, 0, 0, 0, 0, 4096, 0, 0, 0, 0, 4096, 0]


def upload(hf):
    y, u, v = hf.visible()
    has = hasattr(hf, "alpha")
    d = video.DFrame(hf.w, hf.h, fmt=video.PIXFMT_YUVA420P if has else video.PIXFMT_YUV420P).upload(y, u, v)
    if has:
        d.upload_alpha(hf.visible_alpha())
    return d


def scenario(seed):
    rng = np.random.default_rng(seed)
    os.environ["MX_SCALE_INLINE"] = str(int(rng.integers(0, 2)))
    n = int(rng.integers(2, 9))
    W, H = [(320, 180), (322, 182), (640, 360), (200, 120), (1280, 720)][int(rng.integers(0, 5))]
    sizes = []
    for k in range(n):
        c = rng.random()
        if k == 0 or c < 0.45:
            sizes.append((W, H))
        elif c < 0.75:
            sizes.append((2 * int(rng.integers(8, W // 2 + 1)), 2 * int(rng.integers(8, H // 2 + 1))))        # smaller: upscaled with bars
        else:
            sizes.append((2 * int(rng.integers(W // 2, W + 40)), 2 * int(rng.integers(H // 2, H + 40))))  # larger: downscaled
    faders = [float(rng.choice([0.0, 1.0, rng.random(), rng.random(), rng.random()])) for _ in range(n - 1)]
    on_a = [bool(rng.integers(0, 2)) for _ in range(n - 1)]           # the new layer enters on input A (running composite on B) or on B
    matrix = MATRIX if rng.random() < 0.3 else [int(x) for x in rng.integers(-3000, 5000, size=12)]
    layers = []
    for k, (w, h) in enumerate(sizes):
        hf = ov.HostFrame(w, h).fill(int(rng.integers(0, 8)), seed=int(rng.integers(0, 1000)))
        if rng.random() < 0.55:
            hf.set_alpha(ap.alpha_plane(w, h, ap.PATTERNS[int(rng.integers(0, 3))], int(rng.integers(0, 50))))
        layers.append(hf)
    ws = Workspace(44100, 60)
    srcs = [ws.source_video() for _ in sizes]
    prev, mixers = srcs[0], []
    for k in range(1, n):
        m = ws.video_mixer(a=0, b=1, fader=faders[k - 1])
        if on_a[k - 1]:
            ws.connect(srcs[k], 0, m, 0); ws.connect(prev, 0, m, 1)
        else:
            ws.connect(prev, 0, m, 0); ws.connect(srcs[k], 0, m, 1)
        mixers.append(m); prev = m
    rgba = ws.video_to_rgba(matrix)
    ws.connect(prev, 0, rgba, 0)
    ticks = int(rng.integers(1, 6)); runs = int(rng.integers(1, 3))
    g = ws.build(max_ticks_per_run=ticks)
    keep = [upload(l) for l in layers]
    for s, d in zip(srcs, keep):
        video.graph_set_video_source(g, s, d, dur=(1, 60), off=(0, 1), repeat=True)
    oms = [ov.OracleVideoMixer(a=0, b=1, fader=faders[k]) for k in range(n - 1)]
    tick = 0
    for r in range(runs):
        g.run_ticks(tick, ticks)
        for _ in range(ticks):
            prev_o = (layers[0], (1, 60), (0, 1))
            for k in range(n - 1):
                other = (layers[k + 1], (1, 60), (0, 1))
                ins = [other, prev_o, None, None] if on_a[k] else [prev_o, other, None, None]
                prev_o = (oms[k].run_tick(tick * 735, ins), (1, 60), (0, 1))
            tick += 1
        want = prev_o[0]
        got = video.graph_video_output(g, mixers[-1], 0)
        for p, (x, y) in enumerate(zip(got.download(), want.visible())):
            if not np.array_equal(x, y):
                raise AssertionError(f"program plane {p}: {int((x != y).sum())} samples differ (run {r}; sizes {sizes}, faders {faders}, on_a {on_a})")
        if not np.array_equal(video.graph_rgba_output(g, rgba), ov.to_rgba(want, matrix)):
            raise AssertionError(f"RGBA differs (run {r}; sizes {sizes}, faders {faders}, on_a {on_a})")
    g.close()
    return sum(hasattr(l, "alpha") for l in layers)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    fails = with_cov = 0
    for seed in range(first, first + count):
        try:
            with_cov += scenario(seed)
        except Exception as e:   # noqa: BLE001
            fails += 1
            print(f"seed {seed}: FAIL {e}"); traceback.print_exc()
    print(f"{count} alpha cascades from seed {first}, {fails} failures; layers with coverage planes in all: {with_cov}")
    sys.exit(1 if fails else 0)


main()
