"""Envelope on long streams: the state machine over the samples (envelope.rs:91-120) cut into SEGMENTS that run side by side (mx_k_envelope.hip: k_env_flags notes
which tiles hold markers, k_env_resolve steps through the candidate tiles and leaves the state at every segment's start, k_envelope runs one wave per segment).
Bit-exact against the oracle graph: gates that are module outputs (buffers with markers sprinkled in noise, long constant stretches, edges on tile and segment
boundaries, no markers at all), Trigger gates that are not folded into an EqThree (the Envelope has two consumers), state carried across runs, both rates, the
planner's segmenting and forced ones, and the contracted order."""
import numpy as np
import pytest

import oracle
import synth
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace
from test_gpu_audio_parity import RATES, assert_bit_exact

pytestmark = pytest.mark.gpu


def gate_pattern(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "blocks":                                   # a gate signal: exact 0.0 / 1.0, held for random lengths (a few samples to seconds)
        g = np.zeros(n, np.float32); pos = 0; v = 0.0
        while pos < n:
            ln = int(rng.choice([1, 7, 64, 500, 4096, 40000, 200000])); g[pos:pos + ln] = v; pos += ln; v = 1.0 - v
        return g
    if kind == "sprinkled":                                # markers sprinkled in noise
        g = synth.noise(seed, n).copy()
        g[rng.integers(0, n, n // 5000)] = 1.0; g[rng.integers(0, n, n // 7000)] = 0.0
        return g
    if kind == "boundaries":                               # edges exactly on tile (64), step (512) and segment boundaries, and one sample either side
        g = np.full(n, 0.5, np.float32)
        for b in range(512, n, 8192):
            g[b - 1] = 1.0; g[b] = 0.0; g[b + 63] = 1.0; g[b + 64] = 1.0; g[b + 65] = 0.0
        return g
    if kind == "never":
        return np.full(n, 0.25, np.float32)
    if kind == "rare":                                     # one rising and one falling edge in the whole stream
        g = np.full(n, 0.5, np.float32); g[n // 3] = 1.0; g[2 * n // 3 + 11] = 0.0
        return g
    raise ValueError(kind)


@pytest.mark.parametrize("rate", RATES)
@pytest.mark.parametrize("segments", ["-1", "7", "64", "1"])
def test_envelopes_gated_by_buffers_in_segments(rate, segments, monkeypatch):
    SR, SPT = rate
    T = 96                                                  # 76 800 / 70 560 frames: the planner segments on its own (segments = "-1")
    monkeypatch.setenv("MX_ENV_SEGMENTS", segments)
    kinds = ["blocks", "sprinkled", "boundaries", "never", "rare", "blocks"]
    ws = Workspace(SR, 60)
    srcs, envs = [], []
    for k, _kind in enumerate(kinds):
        s = ws.source_mono(); e = ws.envelope(*[(25.0, 500.0, 0.8, 200.0), (1.0, 10.0, 0.3, 5.0), (5.0, 80.0, 0.6, 40.0)][k % 3])
        ws.connect(s, 0, e, 0); srcs.append(s); envs.append(e)
    g = ws.build(max_ticks_per_run=T)
    og = oracle.OracleGraph(ws)
    for run in range(3):
        gates = [gate_pattern(kind, T * SPT, 100 * run + k) for k, kind in enumerate(kinds)]
        for s, v in zip(srcs, gates):
            g.write_source(s, v, T)
        g.run_ticks(run * T, T)
        got = [g.read_output(e, 0, T, False) for e in envs]
        for t in range(T):
            for s, v in zip(srcs, gates):
                og.set_source(s, v[t * SPT:(t + 1) * SPT])
            og.run_tick(run * T + t)
            for k, e in enumerate(envs):
                assert_bit_exact(got[k][t * SPT:(t + 1) * SPT], og.output(e, 0), f"segments {segments} run {run} envelope {k} ({kinds[k]}) tick {t}")
    g.close()


@pytest.mark.parametrize("flags", [0, abi.FLAG_FP_CONTRACT], ids=["exact", "contracted"])
def test_trigger_gated_envelope_with_two_consumers_in_segments(flags):
    """An Envelope whose gate is a Trigger but which feeds TWO Amplifiers is not folded into an EqThree: k_envelope runs it from the Trigger's per-tick bits, in segments
    on a long run, with toggles scheduled inside the run."""
    SR, SPT, T = 48000, 800, 128
    ws = Workspace(SR, 60)
    trig = ws.trigger(False); env = ws.envelope(5.0, 300.0, 0.7, 120.0); ws.connect(trig, 0, env, 0)
    s1, s2 = ws.source_stereo(), ws.source_stereo()
    a1, a2 = ws.amplifier(1.0, 0.5), ws.amplifier(0.8, 1.0)
    ws.connect(s1, 0, a1, 0); ws.connect(env, 0, a1, 1); ws.connect(s2, 0, a2, 0); ws.connect(env, 0, a2, 1)
    g = ws.build(max_ticks_per_run=T, flags=flags)
    with oracle.fp_contract(bool(flags)):
        og = oracle.OracleGraph(ws)
        toggles = {3: 1, 40: 0, 41: 1, 90: 0, 127: 1}
        for run in range(2):
            x1, x2 = synth.noise(300 + run, 2 * T * SPT), synth.noise(310 + run, 2 * T * SPT)
            g.write_source(s1, x1, T); g.write_source(s2, x2, T)
            for t, v in toggles.items():
                g.schedule_params(trig, t, abi.TriggerParams(v ^ run))
            g.run_ticks(run * T, T)
            got = [g.read_output(n, 0, T, st) for n, st in ((env, False), (a1, True), (a2, True))]
            for t in range(T):
                if t in toggles:
                    og.update_params(trig, abi.TriggerParams(toggles[t] ^ run))
                og.set_source(s1, x1[t * 2 * SPT:(t + 1) * 2 * SPT]); og.set_source(s2, x2[t * 2 * SPT:(t + 1) * 2 * SPT])
                og.run_tick(run * T + t)
                for k, n in enumerate((env, a1, a2)):
                    w = og.output(n, 0)
                    assert_bit_exact(got[k][t * w.size:(t + 1) * w.size], w, f"run {run} node {k} tick {t}")
    g.close()
