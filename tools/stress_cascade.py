"""Video sub-graph stress: a cascade of 2 - 5 VideoMixers fed by paced sources (frames of random size, pixel format, life time and offset
on random ticks INSIDE a submission, through mx_graph_queue_video_source), random batch lengths, the RGBA sink at the end and -- in half
of the runs -- a Monitor node beside it that keeps every tick's program picture.  RGBA + program frame at the end of every batch, and
every tick's picture through the Monitor, against the oracle cascade.  Both scaler forms (MX_SCALE_INLINE).
Usage: python tools/stress_cascade.py [first] [count]"""
import os, sys, pathlib, traceback
from fractions import Fraction as F
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle_video as ov
from mixlab_amd import ingest, video
from mixlab_amd.workspace import Workspace

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
SR, SPT = 44100, 735


def rsize(rng):
    if rng.random() < 0.3:
        return [(320, 180), (212, 120), (640, 360), (2, 2), (322, 182), (64, 600)][int(rng.integers(0, 6))]
    return int(rng.integers(1, 220)) * 2, int(rng.integers(1, 160)) * 2


def run(seed):
    rng = np.random.default_rng(seed)
    os.environ["MX_SCALE_INLINE"] = str(int(rng.integers(0, 2)))
    n_layers = int(rng.integers(2, 6)) if rng.random() < 0.8 else int(rng.integers(6, 15)); n_ticks = int(rng.integers(6, 26)); with_monitor = rng.random() < 0.5
    faders = [float(rng.choice([0.0, 1.0, rng.uniform(0, 1)])) for _ in range(n_layers - 1)]
    matrix = None if rng.random() < 0.5 else [int(v) for v in rng.integers(-600, 4600, 12)]
    ws = Workspace(SR, 60)
    srcs = [ws.source_video() for _ in range(n_layers)]
    prev, mixers = srcs[0], []
    for k in range(1, n_layers):
        m = ws.video_mixer(a=0, b=1, fader=faders[k - 1]); ws.connect(prev, 0, m, 0); ws.connect(srcs[k], 0, m, 1); mixers.append(m); prev = m
    rgba = ws.video_to_rgba(matrix); ws.connect(prev, 0, rgba, 0)
    mon_size = rsize(rng)
    mon = None
    if with_monitor:
        mon = ws.monitor(*mon_size); ws.connect(prev, 0, mon, 0)
    max_batch = int(rng.integers(1, 7))
    g = ws.build(max_ticks_per_run=max_batch)
    oms = [ov.OracleVideoMixer(a=0, b=1, fader=faders[k]) for k in range(n_layers - 1)]
    what = f"seed {seed}: {n_layers} layers, {n_ticks} ticks, batches <= {max_batch}, monitor {mon_size if with_monitor else None}, inline {os.environ['MX_SCALE_INLINE']}"
    keep = []
    t = 0
    while t < n_ticks:
        batch = int(min(rng.integers(1, max_batch + 1), n_ticks - t))
        plan = [dict() for _ in range(batch)]
        for kk in range(batch):
            for k in range(n_layers):
                if (t + kk == 0 and k == 0) or rng.random() < 0.4:
                    w, h = rsize(rng); fmt = int(rng.choice([0, 0, 1, 2, 3]))
                    hf = ov.HostFrame(w, h, fmt).fill(int(rng.integers(0, 50)), seed=int(rng.integers(1 << 12)))
                    dur, off = F(int(rng.integers(1, 5)), 60), F(int(rng.integers(-100, 600)), SR)
                    plan[kk][k] = (hf, dur, off)
                    d = video.DFrame(w, h, fmt=fmt).upload(*hf.visible()); keep.append(d)
                    ingest.graph_queue_video_source(g, srcs[k], t + kk, d, dur=dur, off=off)
        g.run_ticks(t, batch)
        want = None; per_tick = []
        for kk in range(batch):
            vin = lambda k: (plan[kk][k][0], (plan[kk][k][1].numerator, plan[kk][k][1].denominator), (plan[kk][k][2].numerator, plan[kk][k][2].denominator)) if k in plan[kk] else None
            prevf = vin(0)
            for k in range(n_layers - 1):
                out = oms[k].run_tick((t + kk) * SPT, [prevf, vin(k + 1), None, None])
                prevf = (out, (1, 60), (0, 1)) if out is not None else None
            want = prevf[0] if prevf else None
            per_tick.append(want)
        got = video.graph_rgba_output(g, rgba)
        end = f"{what}: batch [{t}, {t + batch})"
        if want is None:
            assert got is None, end + ": a picture where the oracle has none"
        else:
            assert got is not None and np.array_equal(got, ov.to_rgba(want, matrix)), end + f": RGBA differs ({want.w}x{want.h})"
            prog = video.graph_video_output(g, mixers[-1], 0)
            for p, (a, b) in enumerate(zip(prog.download(), want.visible())):
                assert np.array_equal(a, b), end + f": plane {p}"
        if mon is not None:
            packed = ingest.graph_read_monitor_video(g, mon, 0, batch)
            for kk, pic in enumerate(per_tick):
                assert (packed[kk] is None) == (pic is None), end + f": monitor presence tick {kk}"
                if pic is None:
                    continue
                if (pic.w, pic.h) == mon_size:
                    w2 = pic
                else:
                    w2 = ov.HostFrame(*mon_size); ov.blank(w2); ov.dynamic_scale(pic, w2)
                for p, (a, b) in enumerate(zip(packed[kk], w2.visible())):
                    assert np.array_equal(a, b), end + f": monitor tick {kk} plane {p}"
        keep = keep[-60:]
        t += batch


bad = 0
for seed in range(first, first + count):
    try:
        run(seed)
    except Exception:
        bad += 1; traceback.print_exc(limit=3)
        if bad >= 3:
            break
print(f"{count} cascade scenarios, {bad} failures")
sys.exit(1 if bad else 0)
