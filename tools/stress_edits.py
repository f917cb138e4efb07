"""Topology-edit stress (Engine::client_update, src/engine.rs:277-398: modules persist while the graph around them changes): a pool of
units -- EqThree, Envelope, FIR, each behind its own source -- of which every epoch's graph holds a random subset in a random node order,
sometimes with fused consumers behind the EqThree; survivors carry their state through mx_graph_adopt_state, a unit that was away comes
back fresh (the reference destroys a removed module).  Every unit's output, every epoch, against its own continuous oracle.
Usage: python tools/stress_edits.py [first] [count]"""
import sys, pathlib, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle, synth
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def run(seed):
    rng = np.random.default_rng(seed)
    SR, SPT = [(44100, 735), (48000, 800)][int(rng.integers(0, 2))]
    T, epochs, K = int(rng.choice([1, 3, 6])), int(rng.integers(3, 7)), int(rng.integers(2, 9))
    units = []
    for u in range(K):
        kind = ["eq", "env", "fir"][int(rng.integers(0, 3))]
        d = {"kind": kind, "state": None, "present": False}
        if kind == "eq":
            d["p"] = tuple(float(v) for v in rng.uniform(-9, 6, 3))
        elif kind == "env":
            d["p"] = (float(rng.uniform(1, 40)), float(rng.uniform(20, 400)), float(rng.uniform(0, 1)), float(rng.uniform(20, 400)))
        else:
            d["p"] = np.asarray(rng.uniform(-0.3, 0.3, int(rng.integers(1, 40))), np.float64)
        units.append(d)
    g_old, ids_old = None, None
    for ep in range(epochs):
        present = [u for u in range(K) if rng.random() < 0.7] or [0]
        order = list(rng.permutation(present))
        ws = Workspace(SR, 60)
        ids = {}
        for u in order:
            d = units[u]
            for _ in range(int(rng.integers(0, 3))):      # unrelated modules in between: node indices shift from epoch to epoch
                ws.oscillator(float(rng.uniform(50, 500)), abi.WAVE_SAW)
            if d["kind"] == "eq":
                s = ws.source_mono(); m = ws.eq_three(*d["p"]); ws.connect(s, 0, m, 0)
                if rng.random() < 0.5:                     # a fused consumer chain behind it in this epoch
                    pan = ws.stereo_panner(); amp = ws.amplifier(1.0, 0.0); ws.connect(m, 0, pan, 0); ws.connect(m, 0, pan, 1); ws.connect(pan, 0, amp, 0)
                    ids[u] = (s, m, amp)
                else:
                    ids[u] = (s, m, None)
            elif d["kind"] == "env":
                s = ws.source_mono(); m = ws.envelope(*d["p"]); ws.connect(s, 0, m, 0); ids[u] = (s, m, None)
            else:
                s = ws.source_stereo(); m = ws.fir(d["p"]); ws.connect(s, 0, m, 0); ids[u] = (s, m, None)
        g = ws.build(max_ticks_per_run=T)
        if g_old is not None:
            mapping = [-1] * len(ws.nodes)
            for u in order:
                if units[u]["present"]:
                    mapping[ids[u][1]] = ids_old[u][1]       # the module itself survives; its source is stateless
            g.adopt_state(g_old, mapping)
            g_old.close() if hasattr(g_old, "close") else None
        for u in range(K):
            if u not in present:
                units[u]["present"] = False; units[u]["state"] = None   # destroyed: comes back fresh
        # inputs + oracle
        for u in order:
            d = units[u]
            s, m, amp = ids[u]
            t0 = ep * T
            if d["kind"] == "eq":
                x = synth.noise(int((seed * 131 + ep * 16 + int(u)) % 100000), T * SPT)
                if d["state"] is None:
                    d["state"] = oracle.eq_three_new(SR)
                d["want"] = oracle.eq_three_run(d["state"], d["p"], x)
                g.write_source(s, x, T)
            elif d["kind"] == "env":
                x = (synth.noise(int((seed * 131 + ep * 16 + int(u)) % 100000), T * SPT) > 0.3).astype(np.float32)
                x = np.repeat(x[:: 97], 97)[: T * SPT] if x.size >= 97 else x
                if d["state"] is None:
                    d["state"] = oracle.EnvState()
                d["want"] = np.concatenate([oracle.envelope_run(d["state"], d["p"], SR, (t0 + k) * SPT, x[k * SPT:(k + 1) * SPT], SPT) for k in range(T)])
                g.write_source(s, x, T)
            else:
                x = synth.noise(int((seed * 131 + ep * 16 + int(u)) % 100000), T * 2 * SPT)
                if d["state"] is None:
                    d["state"] = np.zeros((len(d["p"]) - 1) * 2, np.float32)
                d["want"] = oracle.fir_run(d["p"], d["state"], x)
                g.write_source(s, x, T)
            d["present"] = True
        g.run_ticks(ep * T, T)
        for u in order:
            d = units[u]; s, m, amp = ids[u]
            what = f"seed {seed}: epoch {ep}, unit {u} ({d['kind']}), {SR} Hz, T {T}"
            if d["kind"] == "eq" and amp is not None:
                got = g.read_output(amp, 0, T, True)
                want = oracle.amplifier_run(1.0, 0.0, np.repeat(d["want"], 2), None)
            else:
                got = g.read_output(m, 0, T, d["kind"] == "fir")
                want = d["want"]
            assert np.array_equal(bits(got), bits(want)), what
        g_old, ids_old = g, ids


bad = 0
for seed in range(first, first + count):
    try:
        run(seed)
    except Exception:
        bad += 1; traceback.print_exc(limit=3)
        if bad >= 3:
            break
print(f"{count} edit sequences, {bad} failures")
sys.exit(1 if bad else 0)
