"""Stress of the speculative EqThree's PROOF / REPAIR pass (k_eq_three_repair: four-lane groups, prefix scan, coalescence, standing
states, fills, islands side by side, the in-order fallback) on the input class it exists for: programme that falls exactly silent (or to
an exact DC level) and comes back.  Random layouts per strip -- silences from a few samples to seconds, bursts shorter than the poles'
decay, DC plateaus at random levels, single-sample clicks inside silences, strips muted from the first sample or from the middle of a run,
-0.0 runs, denormal-level noise -- random chunk counts (islands beyond one round of sixteen), forced short warm-ups, both rates, plain EQs and
the fused strip (inline Envelope + Amplifier), two or three runs with the state carried; exact order or, with --contract, the contracted
one (MX_FLAG_FP_CONTRACT against the oracle's contract mode).  Every output compared bit for bit.

    python tools/stress_eq_silences.py [first_seed] [count] [--contract]
"""
import os
import pathlib
import sys
import traceback

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
import synth  # noqa: E402
from mixlab_amd import abi  # noqa: E402
from mixlab_amd.workspace import Workspace  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
first = int(args[0]) if len(args) > 0 else 0
count = int(args[1]) if len(args) > 1 else 50
CONTRACT = "--contract" in sys.argv


def material(rng, seed, k, length):
    x = synth.noise(100000 * seed + k, length).copy()
    kind = int(rng.integers(0, 9))
    if kind == 0:                                   # silences of random length at random places
        pos = int(rng.integers(0, 20000))
        while pos < length:
            n = int(rng.choice([3, 40, 700, 1500, 5000, 20000, 90000]))
            x[pos:pos + n] = 0.0
            pos += n + int(rng.choice([5, 300, 1200, 2500, 9000, 60000]))
    elif kind == 1:                                 # DC plateaus
        pos = int(rng.integers(0, 30000))
        while pos < length:
            n = int(rng.choice([2000, 8000, 40000]))
            x[pos:pos + n] = np.float32(rng.choice([0.25, -0.5, 1.0, 1e-3, -1e-20]))
            pos += n + int(rng.choice([100, 2000, 30000]))
    elif kind == 2:                                 # muted from the start
        x[:] = 0.0
    elif kind == 3:                                 # muted from somewhere on, for ever
        x[int(rng.integers(0, length)):] = 0.0
    elif kind == 4:                                 # silence with clicks
        x[:] = 0.0
        x[rng.integers(0, length, size=int(rng.integers(1, 12)))] = np.float32(0.7)
    elif kind == 5:                                 # minus-zero runs inside plus-zero silence (bit patterns differ, values do not)
        pos = int(rng.integers(0, 5000))
        while pos < length:
            x[pos:pos + 30000] = 0.0
            x[pos + 7000:pos + 7100] = np.float32(-0.0)
            pos += 30000 + int(rng.integers(500, 20000))
    elif kind == 6:                                 # very quiet programme with gaps
        x = (x * np.float32(1e-30)).astype(np.float32)
        x[length // 3: length // 2] = 0.0
    elif kind == 7:                                 # a tone that stops dead and restarts
        t = np.arange(length)
        x = (0.6 * np.sin(2 * np.pi * 330.0 * t / 48000.0)).astype(np.float32)
        for a in range(int(rng.integers(1000, 40000)), length, int(rng.integers(30000, 120000))):
            x[a:a + int(rng.integers(100, 50000))] = 0.0
    # kind 8: plain noise
    return np.ascontiguousarray(x, dtype=np.float32)


def run(seed):
    rng = np.random.default_rng(seed)
    SR, SPT = [(44100, 735), (48000, 800)][int(rng.integers(0, 2))]
    n = int(rng.choice([1, 3, 8, 17]))
    T = int(rng.choice([60, 128, 300, 700]))
    runs = int(rng.choice([2, 3]))
    fused = bool(rng.integers(0, 2))
    ctl_kind = int(rng.integers(0, 4)) if fused else 0     # 0 inline Envelope; 1 a shared triangle LFO, 2 a source per strip, 3 ONE Envelope for all strips (not folded): control BUFFERS
    os.environ["MX_EQ_SPEC_CHUNKS"] = str(int(rng.choice([0, 0, 7, 24, 64, 130, 300, 700])))
    os.environ["MX_EQ_SPEC_WARM"] = str(int(rng.choice([0, 0, 0, 64, 512])))
    os.environ["MX_EQ_SPEC_SB"] = str(int(rng.choice([321, 321, 16])))
    desc = (f"seed {seed}: {SR} Hz, {n} strips, {runs} runs of {T} ticks, fused {fused}, chunks {os.environ['MX_EQ_SPEC_CHUNKS']}, warm {os.environ['MX_EQ_SPEC_WARM']}, "
            f"sb {os.environ['MX_EQ_SPEC_SB']}, ctl {ctl_kind}, contract {CONTRACT}")
    flags = abi.FLAG_FP_CONTRACT if CONTRACT else 0
    L = runs * T * SPT
    sig = [material(rng, seed, k, L) for k in range(n)]
    csig = [np.abs(material(rng, seed, 500 + k, L)) for k in range(n)] if ctl_kind == 2 else []
    ws = Workspace(SR, 60)
    srcs, outs, trigs, ctl_srcs = [], [], [], []
    lfo = ws.oscillator(float(rng.uniform(0.2, 9.0)), abi.WAVE_TRIANGLE) if ctl_kind == 1 else None
    shared_env = None
    if ctl_kind == 3:
        st = ws.trigger(bool(rng.integers(0, 2))); shared_env = ws.envelope(5.0, 80.0, 0.6, 40.0); ws.connect(st, 0, shared_env, 0); trigs.append(st)
        sink = ws.amplifier(1.0, 1.0); ws.connect(shared_env, 0, sink, 1)      # a second consumer keeps it a module of its own even with one strip
    for k in range(n):
        s = ws.source_mono(); e = ws.eq_three(*(float(v) for v in rng.uniform(-24.0, 6.0, 3)))
        ws.connect(s, 0, e, 0); srcs.append(s)
        if fused:
            pan = ws.stereo_panner(); amp = ws.amplifier(float(rng.uniform(0.5, 1.2)), float(rng.uniform(0.0, 1.0)))
            ws.connect(e, 0, pan, 0); ws.connect(e, 0, pan, 1); ws.connect(pan, 0, amp, 0)
            if ctl_kind == 0:
                trig = ws.trigger(bool(rng.integers(0, 2))); env = ws.envelope(5.0, 80.0, 0.6, 40.0)
                ws.connect(trig, 0, env, 0); ws.connect(env, 0, amp, 1); trigs.append(trig)
            elif ctl_kind == 1:
                ws.connect(lfo, 0, amp, 1)
            elif ctl_kind == 2:
                cs = ws.source_mono(); ws.connect(cs, 0, amp, 1); ctl_srcs.append(cs)
            else:
                ws.connect(shared_env, 0, amp, 1)
            outs.append((amp, True))
        else:
            outs.append((e, False))
    g = ws.build(max_ticks_per_run=T, flags=flags)
    with oracle.fp_contract(CONTRACT):
        og = oracle.OracleGraph(ws)
        for r in range(runs):
            sl = slice(r * T * SPT, (r + 1) * T * SPT)
            toggles = {}
            for tr in trigs:
                for t in sorted(set(int(v) for v in rng.integers(1, T, size=int(rng.integers(0, 4))))):
                    toggles.setdefault(t, []).append((tr, int(rng.integers(0, 2))))
            for t, lst in toggles.items():
                for tr, v in lst:
                    g.schedule_params(tr, t, abi.TriggerParams(v))
            for k, s in enumerate(srcs):
                g.write_source(s, sig[k][sl], T)
            for k, s in enumerate(ctl_srcs):
                g.write_source(s, csig[k][sl], T)
            g.run_ticks(r * T, T)
            got = [g.read_output(nd, 0, T, st) for nd, st in outs]
            for t in range(T):
                for tr, v in toggles.get(t, []):
                    og.update_params(tr, abi.TriggerParams(v))
                for k, s in enumerate(srcs):
                    og.set_source(s, sig[k][r * T * SPT + t * SPT: r * T * SPT + (t + 1) * SPT])
                for k, s in enumerate(ctl_srcs):
                    og.set_source(s, csig[k][r * T * SPT + t * SPT: r * T * SPT + (t + 1) * SPT])
                og.run_tick(r * T + t)
                for k, (nd, st) in enumerate(outs):
                    w = og.output(nd, 0)
                    gg = got[k][t * w.size:(t + 1) * w.size]
                    if not np.array_equal(gg.view(np.uint32), w.view(np.uint32)):
                        i = int(np.flatnonzero(gg.view(np.uint32) != w.view(np.uint32))[0])
                        raise AssertionError(f"{desc}: strip {k} run {r} tick {t} sample {i}: got {gg[i]!r} want {w[i]!r} ({int((gg.view(np.uint32) != w.view(np.uint32)).sum())} differ in this tick)")
    st = g.eq_repair_stats()
    g.close()
    return desc, st


tot = {}
bad = 0
for seed in range(first, first + count):
    try:
        desc, st = run(seed)
        for k, v in st.items():
            tot[k] = tot.get(k, 0) + v
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("FAIL", e if isinstance(e, AssertionError) else traceback.format_exc(), flush=True)
print(f"stress_eq_silences: {count} scenarios from seed {first}, {bad} failures; repair pass totals {tot}", flush=True)
sys.exit(1 if bad else 0)
