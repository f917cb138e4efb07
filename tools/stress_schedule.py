"""Random parameter updates at random tick boundaries inside batched submissions (Engine::client_update between ticks,
src/engine.rs:192-214,277-398): several config-2 style strips plus an oscillator bus, every module kind updated at random ticks, random
batch lengths, fused and unfused; master / cue against the oracle ticked with the same updates.
Usage: python tools/stress_schedule.py [first_seed] [count]"""
import sys, pathlib, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle, synth
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100


def run(seed):
    rng = np.random.default_rng(seed)
    SR, SPT = [(44100, 735), (48000, 800)][int(rng.integers(0, 2))]
    n = int(rng.integers(1, 6))
    T = int(rng.choice([3, 8, 20, 33]))
    runs = 2
    ws = Workspace(SR, 60)
    srcs, eqs, trigs, envs, amps = [], [], [], [], []
    chans = []
    for k in range(n):
        s = ws.source_mono(); e = ws.eq_three(float(rng.uniform(-9, 9)), float(rng.uniform(-9, 9)), float(rng.uniform(-9, 9))); pan = ws.stereo_panner()
        t = ws.trigger(bool(rng.integers(0, 2))); ev = ws.envelope(float(rng.uniform(1, 40)), float(rng.uniform(10, 300)), float(rng.uniform(0, 1)), float(rng.uniform(10, 300)))
        a = ws.amplifier(float(rng.uniform(0.2, 1.2)), float(rng.uniform(0, 1)))
        ws.connect(s, 0, e, 0); ws.connect(e, 0, pan, 0); ws.connect(e, 0, pan, 1); ws.connect(pan, 0, a, 0); ws.connect(t, 0, ev, 0); ws.connect(ev, 0, a, 1)
        srcs.append(s); eqs.append(e); trigs.append(t); envs.append(ev); amps.append(a)
        chans.append((float(rng.uniform(-12, 6)), float(rng.uniform(0, 1)), bool(rng.integers(0, 2))))
    osc = ws.oscillator(float(rng.uniform(50, 2000)), int(rng.choice([abi.WAVE_SAW, abi.WAVE_TRIANGLE, abi.WAVE_SQUARE, abi.WAVE_ON])))
    chans.append((0.0, 0.5, False))
    mix = ws.mixer(chans)
    for k, a in enumerate(amps):
        ws.connect(a, 0, mix, k)
    ws.connect(osc, 1, mix, n)
    flags = int(rng.choice([0, abi.FLAG_NO_FUSE]))
    g = ws.build(max_ticks_per_run=T, flags=flags)
    og = oracle.OracleGraph(ws)
    noise = [synth.noise(100 * seed + k, runs * T * SPT) for k in range(n)]

    def random_update():
        kind = int(rng.integers(0, 6)); k = int(rng.integers(0, n))
        if kind == 0:
            return eqs[k], abi.EqThreeParams(float(rng.uniform(-12, 12)), float(rng.uniform(-12, 12)), float(rng.uniform(-12, 12)))
        if kind == 1:
            return trigs[k], abi.TriggerParams(int(rng.integers(0, 2)))
        if kind == 2:
            return envs[k], abi.EnvelopeParams(float(rng.uniform(0.5, 40)), float(rng.uniform(5, 300)), float(rng.uniform(0, 1)), float(rng.uniform(5, 300)))
        if kind == 3:
            return amps[k], abi.AmplifierParams(float(rng.uniform(0, 1.5)), float(rng.uniform(0, 1)))
        if kind == 4:
            return osc, abi.OscillatorParams(float(rng.uniform(50, 3000)), int(rng.choice([abi.WAVE_SAW, abi.WAVE_TRIANGLE, abi.WAVE_SQUARE, abi.WAVE_ON, abi.WAVE_OFF])), 0)
        return mix, [abi.MixerChannelParams(float(rng.uniform(-12, 6)), float(rng.uniform(0, 1)), int(rng.integers(0, 2))) for _ in range(n + 1)]

    for r in range(runs):
        ups = {}
        for _ in range(int(rng.integers(0, 9))):
            ups.setdefault(int(rng.integers(0, T)), []).append(random_update())
        for tick, lst in ups.items():
            for node, p in lst:
                g.schedule_params(node, tick, p)
        for k, s in enumerate(srcs):
            g.write_source(s, noise[k][r * T * SPT:(r + 1) * T * SPT], T)
        g.run_ticks(r * T, T)
        got_m, got_c = g.read_output(mix, 0, T, True), g.read_output(mix, 1, T, True)
        for kk in range(T):
            tick = r * T + kk
            for node, p in ups.get(kk, []):
                og.update_params(node, p)
            for k, s in enumerate(srcs):
                og.set_source(s, noise[k][tick * SPT:(tick + 1) * SPT])
            og.run_tick(tick)
            sl = slice(kk * 2 * SPT, (kk + 1) * 2 * SPT)
            for name, got, want in (("master", got_m[sl], og.output(mix, 0)), ("cue", got_c[sl], og.output(mix, 1))):
                if not np.array_equal(np.asarray(got).view(np.uint32), np.asarray(want, np.float32).view(np.uint32)):
                    raise AssertionError(f"seed {seed} ({SR} Hz, {n} strips, T {T}, flags {flags}): {name} differs on tick {tick}; updates {sorted(ups)}")


bad = 0
for seed in range(first, first + count):
    try:
        run(seed)
    except Exception:
        bad += 1; traceback.print_exc(limit=3)
        if bad >= 3:
            break
print(f"{count} schedules, {bad} failures")
sys.exit(1 if bad else 0)
