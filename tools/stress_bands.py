"""Row-band stress: for random picture sizes, layer sizes (up- and downscaled, pillar / letter boxed) and rank counts, the bands
of a scaled layer computed from halo slices only (mx_video_scale_band, and the graph's band-scaling source node) stitch to the
unsharded DynamicScaler picture bit for bit.  Usage: python tools/stress_bands.py [first_seed] [count]"""
import sys, pathlib, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle_video as ov
from mixlab_amd import shard, video
from mixlab_amd.workspace import Workspace
from test_cpu_video_bands import rows_of

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100


def upload(hf):
    return video.DFrame(hf.w, hf.h).upload(*hf.visible())


bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    world = int(rng.choice([2, 3, 4, 8]))
    W = int(rng.integers(8, 400)) * 2
    H = int(rng.integers(world, 300)) * 2
    if rng.random() < 0.2:
        W, H = 1920, 1080
    lw, lh = int(rng.integers(1, 500)) * 2, int(rng.integers(1, 400)) * 2
    what = f"seed {seed}: layer {lw}x{lh} into {W}x{H} over {world} bands"
    if "-v" in sys.argv:
        print(what, flush=True)
    try:
        layer = ov.HostFrame(lw, lh).fill(seed % 40, seed=seed)
        want = ov.HostFrame(W, H); ov.dynamic_scale(layer, want)
        got = [np.zeros_like(p) for p in want.visible()]
        got_g = [np.zeros_like(p) for p in want.visible()]
        for (row0, rows) in shard.row_bands(H, world):
            d = video.DFrame(W, rows)
            need = shard.band_source_rows((row0, rows), lw, lh, W, H)
            if (lw, lh) == (W, H):
                d = upload(rows_of(layer, row0, rows)); dg = d
            else:
                dg = None
                if need is not None:
                    sl = upload(rows_of(layer, need[0], need[1]))
                    video.scale_band(sl, lh, need[0], d, W, H, row0)
                    # the same band through a graph whose source node scales its halo slice every tick
                    ws = Workspace(44100, 60)
                    sv = ws.source_video(); mx = ws.video_mixer(a=0, b=None, fader=1.0); ws.connect(sv, 0, mx, 0)
                    g = ws.build()
                    video.graph_set_video_source_band(g, sv, lw, lh, need[0], need[1], W, H, row0, rows)
                    video.graph_set_video_source(g, sv, sl, dur=(1, 60), off=(0, 1), repeat=True)
                    g.run_ticks(0, 1)
                    dg = video.graph_video_output(g, mx, 0)
            for p, a in enumerate(d.download()):
                c = 1 if p else 0
                got[p][row0 >> c:(row0 + rows) >> c, :] = a
            if dg is not None:
                for p, a in enumerate(dg.download()):
                    c = 1 if p else 0
                    got_g[p][row0 >> c:(row0 + rows) >> c, :] = a
            else:
                blank = video.DFrame(W, rows)
                for p, a in enumerate(blank.download()):
                    c = 1 if p else 0
                    got_g[p][row0 >> c:(row0 + rows) >> c, :] = a
        for p, (a, b) in enumerate(zip(got, want.visible())):
            assert np.array_equal(a, b), f"{what}: plane {p}: stitched bands differ"
        for p, (a, b) in enumerate(zip(got_g, want.visible())):
            assert np.array_equal(a, b), f"{what}: plane {p}: stitched graph-source bands differ"
    except Exception:
        bad += 1; print(what); traceback.print_exc(limit=3)
        if bad >= 3:
            break
print(f"{count} band jobs, {bad} failures")
sys.exit(1 if bad else 0)
