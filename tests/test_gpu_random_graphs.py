"""Randomised differential test of the graph executor: seeded random module graphs -- fan-out, unconnected inputs,
cycles (back-edges read Disconnected, engine.rs:479-482), modules nobody listens to, mixers of odd widths -- run on the
device (fused and unfused, several ticks per submission and one) and on the CPU oracle's graph runner, every port
compared bit for bit.

Only modules whose device arithmetic is bit-exact are drawn (EqThree in exact-order mode; oscillators Saw / Triangle /
On / Off -- Sine / Square / FmSine differ from the host libm by <= 1 ULP and would make everything downstream inexact;
they have their own tests).  What is under test is the scheduler: run order, levels, launch groups, the fusion planner's
conditions, slab layout, state carry.
"""
import numpy as np
import pytest

import oracle
import synth
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace

pytestmark = pytest.mark.gpu

SR, SPT = 44100, 735
MONO, STEREO = 1, 2


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def random_graph(seed):
    rng = np.random.default_rng(seed)
    ws = Workspace(SR, 60)
    outs = {MONO: [], STEREO: []}      # (node, port) by line type
    ins = []                           # (node, port, type)
    sources = []
    n_nodes = int(rng.integers(8, 40))
    for _ in range(n_nodes):
        k = rng.choice(["src_m", "src_s", "osc", "trig", "eq", "env", "amp", "pan", "split", "mix", "plot", "eq", "amp", "pan"])
        if k == "src_m":
            n = ws.source_mono(); sources.append((n, MONO)); outs[MONO].append((n, 0))
        elif k == "src_s":
            n = ws.source_stereo(); sources.append((n, STEREO)); outs[STEREO].append((n, 0))
        elif k == "osc":
            n = ws.oscillator(float(rng.uniform(50, 2000)), int(rng.choice([abi.WAVE_SAW, abi.WAVE_TRIANGLE, abi.WAVE_ON, abi.WAVE_OFF])))
            outs[MONO].append((n, 0)); outs[STEREO].append((n, 1))
        elif k == "trig":
            n = ws.trigger(bool(rng.integers(0, 2))); outs[MONO].append((n, 0))
        elif k == "eq":
            n = ws.eq_three(*[float(v) for v in rng.uniform(-24, 6, 3)]); ins.append((n, 0, MONO)); outs[MONO].append((n, 0))
        elif k == "env":
            n = ws.envelope(float(rng.uniform(1, 50)), float(rng.uniform(5, 600)), float(rng.uniform(0.1, 1.0)), float(rng.uniform(5, 300)))
            ins.append((n, 0, MONO)); outs[MONO].append((n, 0))
        elif k == "amp":
            n = ws.amplifier(float(rng.uniform(0.1, 2.0)), float(rng.uniform(0.0, 1.0)))
            ins.append((n, 0, STEREO)); ins.append((n, 1, MONO)); outs[STEREO].append((n, 0))
        elif k == "pan":
            n = ws.stereo_panner(); ins.append((n, 0, MONO)); ins.append((n, 1, MONO)); outs[STEREO].append((n, 0))
        elif k == "split":
            n = ws.stereo_splitter(); ins.append((n, 0, STEREO)); outs[MONO].append((n, 0)); outs[MONO].append((n, 1))
        elif k == "mix":
            w = int(rng.integers(0, 7))
            n = ws.mixer([(float(rng.uniform(-24, 6)), float(rng.uniform(0, 1)), bool(rng.integers(0, 2))) for _ in range(w)])
            for c in range(w):
                ins.append((n, c, STEREO))
            outs[STEREO].append((n, 0)); outs[STEREO].append((n, 1))
        else:
            n = ws.plotter(); ins.append((n, 0, STEREO))
    # strips like the benchmark's, so the fusion planner has something to chew on -- with random deviations from the pattern
    for _ in range(int(rng.integers(1, 4))):
        s = ws.source_mono(); sources.append((s, MONO)); e = ws.eq_three(*[float(v) for v in rng.uniform(-24, 6, 3)])
        p = ws.stereo_panner(); a = ws.amplifier(1.0, 0.5); t = ws.trigger(bool(rng.integers(0, 2))); v = ws.envelope()
        ws.connect(s, 0, e, 0); ws.connect(e, 0, p, 0); ws.connect(e, 0, p, 1); ws.connect(p, 0, a, 0); ws.connect(t, 0, v, 0); ws.connect(v, 0, a, 1)
        outs[MONO] += [(e, 0), (v, 0), (t, 0)]; outs[STEREO] += [(p, 0), (a, 0)]
        ins += [(e, 0, MONO), (p, 0, MONO), (p, 1, MONO), (a, 0, STEREO), (a, 1, MONO), (v, 0, MONO)]
    for (n, port, ty) in ins:
        if outs[ty] and rng.random() < 0.8:          # anything of the right type, earlier or later: cycles happen
            sn, sp = outs[ty][int(rng.integers(0, len(outs[ty])))]
            ws.connect(sn, sp, n, port)
    return ws, sources


def port_types(ws):
    res = []
    for kind, params in ws.nodes:
        res.append({abi.KIND_AMPLIFIER: [STEREO], abi.KIND_ENVELOPE: [MONO], abi.KIND_EQ_THREE: [MONO], abi.KIND_MIXER: [STEREO, STEREO],
                    abi.KIND_OSCILLATOR: [MONO, STEREO], abi.KIND_PLOTTER: [], abi.KIND_STEREO_PANNER: [STEREO],
                    abi.KIND_STEREO_SPLITTER: [MONO, MONO], abi.KIND_TRIGGER: [MONO], abi.KIND_SOURCE_MONO: [MONO],
                    abi.KIND_SOURCE_STEREO: [STEREO]}[kind])
    return res


@pytest.mark.parametrize("seed", list(range(64)))
def test_random_graph_matches_the_oracle_on_every_port(seed):
    ws, sources = random_graph(seed)
    T, runs = 3, 2
    og = oracle.OracleGraph(ws)
    order = og.run_order()
    graphs = {"fused": ws.build(max_ticks_per_run=T, flags=abi.FLAG_EQ_EXACT),
              "unfused": ws.build(max_ticks_per_run=T, flags=abi.FLAG_EQ_EXACT | abi.FLAG_NO_FUSE),
              "ticked": ws.build(max_ticks_per_run=1, flags=abi.FLAG_EQ_EXACT),
              # where the graph ends in a Mixer bank fed by computed ports, that bank runs on a second stream (else the flag changes nothing)
              "overlap": ws.build(max_ticks_per_run=T, flags=abi.FLAG_EQ_EXACT | abi.FLAG_OVERLAP_TAIL)}
    for g in graphs.values():
        assert g.run_order() == order
    data = {n: synth.noise(4000 + 31 * seed + n, runs * T * SPT * (1 if ty == MONO else 2)) for (n, ty) in sources}
    types = port_types(ws)
    in_order = set(order)
    for run in range(runs):
        want = {}
        for t in range(T):
            tick = run * T + t
            for (n, ty) in sources:
                w = SPT * (1 if ty == MONO else 2)
                og.set_source(n, data[n][tick * w:(tick + 1) * w])
            og.run_tick(tick)
            for n in in_order:
                for p in range(len(types[n])):
                    want.setdefault((n, p), []).append(og.output(n, p))
        got = {}
        for name, g in graphs.items():
            step = 1 if name == "ticked" else T
            for t0 in range(0, T, step):
                for (n, ty) in sources:
                    w = SPT * (1 if ty == MONO else 2)
                    g.write_source(n, data[n][(run * T + t0) * w:(run * T + t0 + step) * w], step)
                g.run_ticks(run * T + t0, step)
                for n in in_order:
                    for p, ty in enumerate(types[n]):
                        try:
                            chunk = g.read_output(n, p, step, ty == STEREO)
                        except abi.MxError as e:
                            assert name != "unfused" and "MX_FLAG_NO_FUSE" in str(e)     # folded away by the graph compiler: not observable
                            continue
                        got.setdefault((name, n, p), []).append(chunk)
        for (name, n, p), chunks in got.items():
            a, b = np.concatenate(chunks), np.concatenate(want[(n, p)])
            assert np.array_equal(bits(a), bits(b)), f"seed {seed} run {run}: {name} graph, node {n} (kind {ws.nodes[n][0]}) port {p} differs from the oracle"


@pytest.mark.parametrize("seed", list(range(100, 124)))
def test_random_graph_fusion_and_batching_are_invisible_with_the_time_parallel_eq(seed):
    # default EqThree (chunked scan): fused == unfused on every surviving port, bit for bit; one tick per submission differs from
    # batched only where chunk boundaries differ, i.e. by <= 1 ULP per EqThree in the path -- checked on EqThree ports directly
    ws, sources = random_graph(seed)
    T = 4
    gf = ws.build(max_ticks_per_run=T)
    gu = ws.build(max_ticks_per_run=T, flags=abi.FLAG_NO_FUSE)
    data = {n: synth.noise(9000 + 17 * seed + n, T * SPT * (1 if ty == MONO else 2)) for (n, ty) in sources}
    for g in (gf, gu):
        for (n, ty) in sources:
            g.write_source(n, data[n], T)
        g.run_ticks(0, T)
    types = port_types(ws)
    for n in gf.run_order():
        for p, ty in enumerate(types[n]):
            try:
                a = gf.read_output(n, p, T, ty == STEREO)
            except abi.MxError as e:
                assert "MX_FLAG_NO_FUSE" in str(e)
                continue
            assert np.array_equal(bits(a), bits(gu.read_output(n, p, T, ty == STEREO))), f"seed {seed}: node {n} port {p}: fused != unfused"
