"""Mixer stress: (1) the module path with random channel counts (1 - 1100), buffer lengths, cue flags, disconnected inputs and forced
kernel choices (streaming / cooperative); (2) graphs whose Mixer sees mono-dup strips (fused EqThree -> panner -> amplifier), plain stereo
sources and an oscillator at once (the mixed descriptor path), batched.  Bit-exact against the oracle.
Usage: python tools/stress_mixer.py [first_seed] [count]"""
import os, sys, pathlib, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle, synth
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100


def bits(a):
    """bit patterns, with every NaN mapped to one pattern: which NaN (sign, payload) comes out when SEVERAL NaN-producing events meet in
    one sum is left to the implementation by IEEE 754 (x86 picks by operand order, which the compiler chooses) -- NaN-ness is compared,
    everything else (infinities, signed zeros, subnormals) bit for bit"""
    a = np.ascontiguousarray(a, np.float32)
    b = a.view(np.uint32).copy()
    b[np.isnan(a)] = 0x7fc00000
    return b


def module_case(seed):
    rng = np.random.default_rng(seed)
    n_ch = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 127, 128, 129, 200, 511, 1100])) if rng.random() < 0.5 else int(rng.integers(1, 400))
    length = 2 * int(rng.choice([1, 2, 63, 64, 65, 733, 735, 800, 1600, 2999]))
    os.environ["MX_MIXER_COOP_BLOCKS"] = str(int(rng.choice([0, 512, 100000])))
    chans = [(float(rng.uniform(-24, 6)), float(rng.uniform(0, 1)), bool(rng.integers(0, 2))) for _ in range(n_ch)]
    ins = [None if rng.random() < 0.1 else synth.noise(seed * 2000 + i, length) for i in range(n_ch)]
    if rng.random() < 0.2:
        for a in ins:
            if a is not None and rng.random() < 0.3:
                a[int(rng.integers(0, length))] = [np.inf, -np.inf, np.nan, 1e-42, -0.0][int(rng.integers(0, 5))]
    want_m, want_c = oracle.mixer_run(chans, ins, length)
    m = abi.Module(abi.KIND_MIXER, [abi.MixerChannelParams(g, f, 1 if c else 0) for g, f, c in chans])
    got_m, got_c = np.empty(length, np.float32), np.empty(length, np.float32)
    m.run_tick(0, [(abi.MX_STEREO if a is not None else abi.MX_DISCONNECTED, a) for a in ins], [(abi.MX_STEREO, got_m), (abi.MX_STEREO, got_c)])
    what = f"seed {seed}: module mixer {n_ch} ch, {length} floats, coop {os.environ['MX_MIXER_COOP_BLOCKS']}"
    for name, got, want in (("master", got_m, want_m), ("cue", got_c, want_c)):
        d = np.flatnonzero(bits(got) != bits(want))
        assert d.size == 0, what + f": {name}: {d.size} differ, first at {d[:4].tolist()} got {got[d[:4]].tolist()} ({[hex(v) for v in bits(got)[d[:4]]]}) want {want[d[:4]].tolist()} ({[hex(v) for v in bits(want)[d[:4]]]})"


def graph_case(seed):
    rng = np.random.default_rng(seed + 10**6)
    SR, SPT = [(44100, 735), (48000, 800)][int(rng.integers(0, 2))]
    os.environ["MX_MIXER_COOP_BLOCKS"] = str(int(rng.choice([0, 512, 100000])))
    n_strip, n_plain = int(rng.integers(0, 140)), int(rng.integers(0, 70))
    T = int(rng.choice([1, 2, 5, 16]))
    ws = Workspace(SR, 60)
    srcs, plain, chans, ins = [], [], [], []
    for k in range(n_strip):
        s = ws.source_mono(); e = ws.eq_three(float(rng.uniform(-6, 6)), 0.0, float(rng.uniform(-6, 6))); pan = ws.stereo_panner(); a = ws.amplifier(float(rng.uniform(0.3, 1)), 0.0)
        ws.connect(s, 0, e, 0); ws.connect(e, 0, pan, 0); ws.connect(e, 0, pan, 1); ws.connect(pan, 0, a, 0)
        srcs.append(s); ins.append(a)
    for k in range(n_plain):
        s = ws.source_stereo(); plain.append(s); ins.append(s)
    osc = ws.oscillator(float(rng.uniform(100, 900)), abi.WAVE_SAW); ins.append(None)
    order = rng.permutation(len(ins))
    chans = [(float(rng.uniform(-12, 3)), float(rng.uniform(0, 1)), bool(rng.integers(0, 2))) for _ in ins]
    mix = ws.mixer(chans)
    for slot, idx in enumerate(order):
        node = ins[idx]
        if node is None:
            ws.connect(osc, 1, mix, slot)
        elif rng.random() < 0.95:
            ws.connect(node, 0, mix, slot)
    g = ws.build(max_ticks_per_run=T, flags=int(rng.choice([0, 0, abi.FLAG_NO_FUSE])))
    og = oracle.OracleGraph(ws)
    xm = [synth.noise(seed * 3000 + k, 2 * T * SPT) for k in range(n_strip)]
    xs = [synth.noise(seed * 3000 + 1500 + k, 2 * T * 2 * SPT) for k in range(n_plain)]
    for r in range(2):
        for k, s in enumerate(srcs):
            g.write_source(s, xm[k][r * T * SPT:(r + 1) * T * SPT], T)
        for k, s in enumerate(plain):
            g.write_source(s, xs[k][r * T * 2 * SPT:(r + 1) * T * 2 * SPT], T)
        g.run_ticks(r * T, T)
        got_m, got_c = g.read_output(mix, 0, T, True), g.read_output(mix, 1, T, True)
        for kk in range(T):
            tick = r * T + kk
            for k, s in enumerate(srcs):
                og.set_source(s, xm[k][tick * SPT:(tick + 1) * SPT])
            for k, s in enumerate(plain):
                og.set_source(s, xs[k][tick * 2 * SPT:(tick + 1) * 2 * SPT])
            og.run_tick(tick)
            sl = slice(kk * 2 * SPT, (kk + 1) * 2 * SPT)
            what = f"seed {seed}: graph mixer {n_strip} strips + {n_plain} stereo + osc, T {T}, {SR} Hz, coop {os.environ['MX_MIXER_COOP_BLOCKS']}, tick {tick}"
            assert np.array_equal(bits(got_m[sl]), bits(og.output(mix, 0))), what + ": master"
            assert np.array_equal(bits(got_c[sl]), bits(og.output(mix, 1))), what + ": cue"


bad = 0
for seed in range(first, first + count):
    for fn in (module_case, graph_case):
        try:
            fn(seed)
        except Exception:
            bad += 1; traceback.print_exc(limit=3)
    if bad >= 3:
        break
print(f"{count} module + {count} graph mixer cases, {bad} failures")
sys.exit(1 if bad else 0)
