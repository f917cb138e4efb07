"""Module-path stress (the ModuleT::run_tick surface, mx_module_run_tick): EqThree, Envelope and Amplifier instances driven by sequences of
calls with RANDOM buffer lengths (1 - 40 000 samples: every exact-EqThree kernel form -- one lane, split cascades, speculative chunks -- is
crossed), state carried from call to call, gate signals with arbitrary per-sample markers; against the oracle run over the concatenated
stream.  Usage: python tools/stress_modules.py [first] [count]"""
import os, sys, pathlib, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle, synth
from mixlab_amd import abi

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def lengths(rng):
    n = int(rng.integers(1, 7))
    return [int(rng.choice([1, 2, 3, 63, 64, 65, 735, 800, 2559, 2560, 2561, 5000, 20000, 40000])) if rng.random() < 0.6 else int(rng.integers(1, 3000)) for _ in range(n)]


def run(seed):
    rng = np.random.default_rng(seed)
    SR = int(rng.choice([44100, 48000]))
    os.environ["MX_EQ_SPEC_CHUNKS"] = str(int(rng.choice([0, 0, 0, 2, 3, 7])))
    os.environ["MX_EQ_SPEC_WARM"] = str(int(rng.choice([0, 0, 0, 128])))
    ls = lengths(rng); total = sum(ls)
    x = synth.noise(seed % 90000, total)
    if rng.random() < 0.2:
        x = x * np.float32(rng.choice([0.0, 1e-30, 1e3]))
    what = f"seed {seed}: {SR} Hz, calls {ls}, chunks {os.environ['MX_EQ_SPEC_CHUNKS']} warm {os.environ['MX_EQ_SPEC_WARM']}"
    # EqThree
    gains = tuple(float(v) for v in rng.uniform(-24, 6, 3))
    st = oracle.eq_three_new(SR); want = oracle.eq_three_run(st, gains, x)
    m = abi.Module(abi.KIND_EQ_THREE, abi.EqThreeParams(*gains), sample_rate=SR)
    got = np.empty_like(x); pos = 0
    for n in ls:
        m.run_tick(pos, [(abi.MX_MONO, x[pos:pos + n])], [(abi.MX_MONO, got[pos:pos + n])]); pos += n
    d = np.flatnonzero(bits(got) != bits(want))
    assert d.size == 0, what + f": EqThree differs at {d[:3].tolist()} of {total}"
    # Envelope: markers anywhere in the stream
    gate = synth.noise((seed + 7) % 90000, total).copy()
    gate[rng.random(total) < 0.01] = 1.0; gate[rng.random(total) < 0.01] = 0.0
    ep = (float(rng.uniform(0.5, 50)), float(rng.uniform(5, 500)), float(rng.uniform(0, 1)), float(rng.uniform(5, 500)))
    vst = oracle.EnvState(); want_e = np.empty(total, np.float32); pos = 0
    me = abi.Module(abi.KIND_ENVELOPE, abi.EnvelopeParams(*ep), sample_rate=SR)
    got_e = np.empty(total, np.float32)
    for n in ls:
        want_e[pos:pos + n] = oracle.envelope_run(vst, ep, SR, pos, gate[pos:pos + n], n)
        me.run_tick(pos, [(abi.MX_MONO, gate[pos:pos + n])], [(abi.MX_MONO, got_e[pos:pos + n])]); pos += n
    d = np.flatnonzero(bits(got_e) != bits(want_e))
    assert d.size == 0, what + f": Envelope {ep} differs at {d[:3].tolist()}"
    # Amplifier with that envelope as control
    xs = np.repeat(x, 2)
    amp, depth = float(rng.uniform(0, 1.5)), float(rng.uniform(0, 1))
    want_a = oracle.amplifier_run(amp, depth, xs, want_e)
    ma = abi.Module(abi.KIND_AMPLIFIER, abi.AmplifierParams(amp, depth), sample_rate=SR)
    got_a = np.empty_like(xs); pos = 0
    for n in ls:
        ma.run_tick(pos, [(abi.MX_STEREO, xs[2 * pos:2 * (pos + n)]), (abi.MX_MONO, want_e[pos:pos + n])], [(abi.MX_STEREO, got_a[2 * pos:2 * (pos + n)])]); pos += n
    d = np.flatnonzero(bits(got_a) != bits(want_a))
    assert d.size == 0, what + f": Amplifier differs at {d[:3].tolist()}"


bad = 0
for seed in range(first, first + count):
    try:
        run(seed)
    except Exception:
        bad += 1; traceback.print_exc(limit=3)
        if bad >= 3:
            break
print(f"{count} module sequences, {bad} failures")
sys.exit(1 if bad else 0)
