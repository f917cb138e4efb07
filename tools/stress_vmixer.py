"""VideoMixer scenarios with RANDOM picture sizes and input pixel formats (thin, tiny, up- and downscaled, 4:2:0 / 4:2:2 / 4:4:4 / nv12),
random arrivals, durations, offsets, A / B / fader changes: program frame presence and pixels, tick by tick, against the oracle state
machine.  Usage: python tools/stress_vmixer.py [first_seed] [count]"""
import sys, pathlib, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle_video as ov
from mixlab_amd import video

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
SPT = 735


def rsize(rng):
    w, h = int(rng.integers(1, 200)) * 2, int(rng.integers(1, 150)) * 2
    if rng.random() < 0.1:
        w = int(rng.choice([2, 4, 640, 1280]))
    if rng.random() < 0.1:
        h = int(rng.choice([2, 4, 360, 720]))
    return w, h


def run(seed):
    rng = np.random.default_rng(seed)
    pool = []
    for _ in range(8):
        w, h = rsize(rng); fmt = int(rng.choice([0, 0, 0, 1, 2, 3]))
        pool.append(ov.HostFrame(w, h, fmt).fill(int(rng.integers(0, 50)), seed=int(rng.integers(0, 99))))
    a, b, fader = 0, 1, float(rng.uniform(0, 1))
    gm, om = video.VideoMixer(a=a, b=b, fader=fader), ov.OracleVideoMixer(a=a, b=b, fader=fader)
    keep = []
    for tick in range(24):
        ins_h = [None] * 4
        for ch in range(4):
            if rng.random() < (0.5, 0.3, 0.15, 0.05)[ch]:
                dur = [(1, 60), (1, 30), (1, 20), (1, 10), (2, 25)][int(rng.integers(0, 5))]
                off = [(0, 1), (1, 240), (1, 120), (-1, 61)][int(rng.integers(0, 4))]
                ins_h[ch] = (pool[int(rng.integers(0, len(pool)))], dur, off)
        if rng.random() < 0.15:
            a = [None, 0, 1, 2, 3][int(rng.integers(0, 5))]; b = [None, 0, 1, 2, 3][int(rng.integers(0, 5))]
            fader = float([0.0, 1.0, rng.uniform(0, 1)][int(rng.integers(0, 3))])
            gm.update(a=a, b=b, fader=fader); om.update(a=a, b=b, fader=fader)
        ins_d = []
        for e in ins_h:
            if e is None:
                ins_d.append(None)
            else:
                d = video.DFrame(e[0].w, e[0].h, fmt=e[0].fmt).upload(*e[0].visible()); keep.append(d)
                ins_d.append((d, e[1], e[2]))
        prog, fa, fb = gm.run_tick(tick * SPT, ins_d)
        want = om.run_tick(tick * SPT, ins_h)
        what = f"seed {seed} tick {tick} sizes {[(f.w, f.h, f.fmt) for f in pool]}"
        assert (prog is None) == (want is None), what + ": program presence"
        if want is not None:
            assert (prog.width, prog.height) == (want.w, want.h), what + ": target size"
            for p, (x, y) in enumerate(zip(prog.download(), want.visible())):
                assert np.array_equal(x, y), what + f": plane {p} differs"
        keep = keep[-16:]


bad = 0
for seed in range(first, first + count):
    try:
        run(seed)
    except Exception:
        bad += 1; traceback.print_exc(limit=3)
        if bad >= 3:
            break
print(f"{count} scenarios, {bad} failures")
sys.exit(1 if bad else 0)
