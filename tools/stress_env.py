"""Envelope / Amplifier stress in the fused strip (Trigger -> Envelope -> Amplifier control of EqThree -> panner) and unfused: random and
EXTREME Envelope parameters (zero, tiny, huge, negative sustain, > 1 sustain), amplifier depths outside [0, 1], random gate patterns
per tick, random batch lengths; amplifier outputs and the mix against the oracle.  Usage: python tools/stress_env.py [first] [count]"""
import ctypes as C, os, sys, pathlib, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle, synth
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace

def canon(a):
    """bit patterns with every NaN mapped to one pattern: the sign / payload of a NaN an invalid operation GENERATES (0 x inf with a zero
    release time, inf - inf) is implementation-defined in IEEE 754 -- x86 sets the sign bit, gfx950's f64 multiply does not"""
    a = np.ascontiguousarray(a, np.float32)
    b = a.view(np.uint32).copy()
    b[np.isnan(a)] = 0x7fc00000
    return b


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100


def pick(rng, normal, extremes):
    return float(rng.choice(extremes)) if rng.random() < 0.35 else float(rng.uniform(*normal))


def run(seed):
    rng = np.random.default_rng(seed)
    SR, SPT = [(44100, 735), (48000, 800)][int(rng.integers(0, 2))]
    n = int(rng.integers(1, 5))
    T = int(rng.choice([1, 4, 9, 40, 130]))
    os.environ["MX_EQ_SPEC_CHUNKS"] = str(int(rng.choice([0, 0, 2, 5])))
    ws = Workspace(SR, 60)
    srcs, trigs, amps = [], [], []
    global PARAMS
    PARAMS = []
    for k in range(n):
        trig = ws.trigger(bool(rng.integers(0, 2)))
        ep = (pick(rng, (0.5, 60), [0.0, 1e-6, 1e-3, 5000.0]), pick(rng, (1, 600), [0.0, 1e-6, 1e-3, 20000.0]),
              pick(rng, (0, 1), [0.0, 1.0, -0.25, 1.5, 1e-9]), pick(rng, (1, 600), [0.0, 1e-6, 1e-3, 20000.0]))
        PARAMS.append(ep)
        env = ws.envelope(*ep)
        src = ws.source_mono(); eq = ws.eq_three(float(rng.uniform(-12, 6)), float(rng.uniform(-12, 6)), float(rng.uniform(-12, 6)))
        pan = ws.stereo_panner(); amp = ws.amplifier(pick(rng, (0.2, 1.2), [0.0, 1.0, -1.0, 3.0]), pick(rng, (0, 1), [0.0, 1.0, -0.5, 2.0]))
        ws.connect(trig, 0, env, 0); ws.connect(src, 0, eq, 0); ws.connect(eq, 0, pan, 0); ws.connect(eq, 0, pan, 1); ws.connect(pan, 0, amp, 0); ws.connect(env, 0, amp, 1)
        srcs.append(src); trigs.append(trig); amps.append(amp)
    mix = ws.mixer([(float(rng.uniform(-6, 3)), float(rng.uniform(0.2, 1)), bool(rng.integers(0, 2))) for _ in range(n)])
    for k, a in enumerate(amps):
        ws.connect(a, 0, mix, k)
    flags = int(rng.choice([0, 0, abi.FLAG_NO_FUSE]))
    g = ws.build(max_ticks_per_run=T, flags=flags)
    og = oracle.OracleGraph(ws)
    p_toggle = float(rng.choice([0.02, 0.2, 0.5]))
    gates = [[bool(rng.integers(0, 2))] for _ in range(n)]
    runs = 2 if T <= 40 else 1
    for k in range(n):
        for _ in range(runs * T):
            gates[k].append(gates[k][-1] ^ (rng.random() < p_toggle))
    noise = [synth.noise(seed * 50 + k, runs * T * SPT) for k in range(n)]
    for r in range(runs):
        keep, events = [], []
        for k, tr in enumerate(trigs):
            g.update_params(tr, abi.TriggerParams(1 if gates[k][r * T] else 0))
            for c in range(1, T):
                if gates[k][r * T + c] != gates[k][r * T + c - 1]:
                    p = abi.TriggerParams(1 if gates[k][r * T + c] else 0); keep.append(p)
                    events.append(abi.ParamEvent(tr, c, C.cast(C.pointer(p), C.c_void_p), C.sizeof(p)))
        if events:
            g.schedule_params_batch((abi.ParamEvent * len(events))(*events))
        for k, s in enumerate(srcs):
            g.write_source(s, noise[k][r * T * SPT:(r + 1) * T * SPT], T)
        g.run_ticks(r * T, T)
        got_m = g.read_output(mix, 0, T, True)
        got_a = [g.read_output(a, 0, T, True) for a in amps] if flags else None
        for kk in range(T):
            tick = r * T + kk
            for k, tr in enumerate(trigs):
                og.update_params(tr, abi.TriggerParams(1 if gates[k][tick] else 0))
            for k, s in enumerate(srcs):
                og.set_source(s, noise[k][tick * SPT:(tick + 1) * SPT])
            og.run_tick(tick)
            sl = slice(kk * 2 * SPT, (kk + 1) * 2 * SPT)
            what = f"seed {seed}: {SR} Hz, {n} strips, T {T}, flags {flags}, tick {tick}"
            gm_, wm_ = np.asarray(got_m[sl], np.float32), np.asarray(og.output(mix, 0), np.float32)
            d = np.flatnonzero(canon(gm_) != canon(wm_))
            if d.size:
                raise AssertionError(what + f": master differs at {d[:4].tolist()} ({d.size}): got {gm_[d[:4]].tolist()} want {wm_[d[:4]].tolist()}; params {PARAMS}")
            if got_a:
                for k, a in enumerate(amps):
                    if not np.array_equal(canon(got_a[k][sl]), canon(og.output(a, 0))):
                        raise AssertionError(what + f": amplifier {k} differs")


bad = 0
for seed in range(first, first + count):
    try:
        run(seed)
    except Exception:
        bad += 1; traceback.print_exc(limit=3)
        if bad >= 3:
            break
print(f"{count} envelope scenarios, {bad} failures")
sys.exit(1 if bad else 0)
