"""Monitor-node stress: a source of random pictures (sizes, pixel formats, sparse ticks, signed offsets) straight into a Monitor of a random
encoder size, and through a VideoMixer first; random batch lengths.  Per tick: presence, timestamps, the picture (per-tick handle and the
packed read-back) against DynamicScaler of the oracle; PCM against the oracle's f32 -> i16.  Usage: python tools/stress_monitor.py [first] [count]"""
import ctypes as C, sys, pathlib, traceback
from fractions import Fraction as F
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle, oracle_video as ov, synth
from mixlab_amd import abi, ingest, video
from mixlab_amd.workspace import Workspace

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
oracle.lib.orc_f32_to_i16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]


def to_i16(x):
    x = np.ascontiguousarray(x, np.float32); out = np.empty(x.size, np.int16)
    oracle.lib.orc_f32_to_i16(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), x.size)
    return out


def rsize(rng):
    if rng.random() < 0.2:
        return [(560, 350), (1120, 700), (1920, 1080), (1280, 720), (2, 2)][int(rng.integers(0, 5))]
    return int(rng.integers(1, 300)) * 2, int(rng.integers(1, 200)) * 2


def run(seed):
    rng = np.random.default_rng(seed)
    SR, SPT = [(44100, 735), (48000, 800)][int(rng.integers(0, 2))]
    T, runs, first_tick = int(rng.choice([1, 3, 8, 17])), int(rng.integers(1, 4)), int(rng.integers(0, 1000))
    mon_size = rsize(rng)
    through_mixer = rng.random() < 0.5
    ws = Workspace(SR, 60)
    sv = ws.source_video()
    last = sv
    fader, gain = float(rng.uniform(0, 1)), float(rng.uniform(0.5, 2.5))
    omx = ov.OracleVideoMixer(a=0, b=None, fader=fader, sample_rate=SR)
    if through_mixer:
        mx = ws.video_mixer(a=0, b=None, fader=fader); ws.connect(sv, 0, mx, 0); last = mx
    src = ws.source_stereo(); amp = ws.amplifier(gain, 0.0); ws.connect(src, 0, amp, 0)
    m = ws.monitor(*mon_size)
    ws.connect(last, 0, m, 0); ws.connect(amp, 0, m, 1)
    g = ws.build(max_ticks_per_run=T)
    what = f"seed {seed}: monitor {mon_size}, T {T} x {runs}, mixer {through_mixer}, {SR} Hz"
    keep = []
    for r in range(runs):
        t0 = first_tick + r * T
        audio = synth.noise(seed * 10 + r, T * 2 * SPT)
        g.write_source(src, audio, T)
        plan = {}
        for k in range(T):
            if rng.random() < 0.6:
                w, h = rsize(rng); fmt = 0 if through_mixer and rng.random() < 0.5 else int(rng.choice([0, 0, 1, 2, 3]))
                hf = ov.HostFrame(w, h, fmt).fill(int(rng.integers(0, 50)), seed=int(rng.integers(0, 999)))
                dur, off = F(int(rng.integers(1, 5)), 60), F(int(rng.integers(-300, 700)), SR)
                plan[k] = (hf, dur, off)
                d = video.DFrame(w, h, fmt=fmt).upload(*hf.visible()); keep.append(d)
                ingest.graph_queue_video_source(g, sv, t0 + k, d, dur=dur, off=off)
        g.run_ticks(t0, T)
        got_pcm = ingest.graph_read_monitor_audio_i16(g, m, T, SPT)
        want_pcm = to_i16(oracle.amplifier_run(gain, 0.0, audio, None))
        assert np.array_equal(got_pcm, want_pcm), what + ": PCM"
        packed = ingest.graph_read_monitor_video(g, m, 0, T)
        for k in range(T):
            tick = t0 + k
            ts, vid = ingest.graph_read_monitor_tick(g, m, k)
            assert ts == F(tick * SPT, SR) - F(first_tick * SPT, SR), what + f": ts tick {k}"
            pic, want_ts, want_dur = None, None, None
            if through_mixer:
                e = plan.get(k)
                vin = (e[0], (e[1].numerator, e[1].denominator), (e[2].numerator, e[2].denominator)) if e else None
                pic = omx.run_tick(tick * SPT, [vin, None, None, None])
                want_ts, want_dur = ts, F(1, 60)
            elif k in plan:
                pic, want_dur, off = plan[k]; want_ts = ts + off
            assert (vid is None) == (pic is None), what + f": presence tick {k}: device {vid is not None} oracle {pic is not None}; plan {[(kk, (v[0].w, v[0].h, v[0].fmt), str(v[1]), str(v[2])) for kk, v in plan.items()]}"
            assert (packed[k] is None) == (pic is None), what + f": packed presence tick {k}"
            if pic is None:
                continue
            frame, frame_ts, dur = vid
            assert frame_ts == want_ts and dur == want_dur, what + f": frame timestamps tick {k}"
            if (pic.w, pic.h, pic.fmt) == (mon_size[0], mon_size[1], 0):
                want = pic
            else:
                want = ov.HostFrame(*mon_size); ov.blank(want); ov.dynamic_scale(pic, want)
            for p, (x, y, z) in enumerate(zip(frame.download(), want.visible(), packed[k])):
                assert np.array_equal(x, y), what + f": tick {k} plane {p} ({pic.w}x{pic.h} fmt {pic.fmt})"
                assert np.array_equal(z, y), what + f": packed tick {k} plane {p}"
        keep = keep[-40:]


bad = 0
for seed in range(first, first + count):
    try:
        run(seed)
    except Exception:
        bad += 1; traceback.print_exc(limit=3)
        if bad >= 3:
            break
print(f"{count} monitor scenarios, {bad} failures")
sys.exit(1 if bad else 0)
