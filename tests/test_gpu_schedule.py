"""Parameter updates BETWEEN the ticks of one submission (mx_graph_schedule_params): the reference drains its command queue
after every tick (src/engine.rs:192-214) and applies ModuleT::update (src/engine.rs:277-398), so a batched run must be able to
change a module's params at any tick boundary inside the batch.

Oracle: the graph runner ticked one tick at a time with update_params between the ticks.  Bit-exact in the default (exact
EqThree) mode; the opt-in fast mode keeps its <= 1 ULP contract per strip.

SURVEY 8d config 2 as written: gates toggle every 30 ticks with per-strip phase k mod 60.
"""
import ctypes as C

import numpy as np
import pytest

import oracle
import synth
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace
from test_gpu_audio_parity import RATES, assert_bit_exact, assert_ulp, strips

pytestmark = pytest.mark.gpu


def gate_open(tick, k):
    """Trigger of strip k: toggles every 30 ticks, phase k mod 60."""
    return ((tick + k) // 30) % 2 == 1


def schedule_gates(g, trigs, t0, batch, og=None):
    """Set every gate for tick t0 and queue its toggles inside [t0, t0 + batch)."""
    keep = []
    events = []
    for k, tr in enumerate(trigs):
        g.update_params(tr, abi.TriggerParams(1 if gate_open(t0, k) else 0))
        for c in range(1, batch):
            if gate_open(t0 + c, k) != gate_open(t0 + c - 1, k):
                p = abi.TriggerParams(1 if gate_open(t0 + c, k) else 0)
                keep.append(p)
                events.append(abi.ParamEvent(tr, c, C.cast(C.pointer(p), C.c_void_p), C.sizeof(p)))
    if events:
        arr = (abi.ParamEvent * len(events))(*events)
        g.schedule_params_batch(arr)
    return len(events)


@pytest.mark.parametrize("rate", RATES)
@pytest.mark.parametrize("batch,mode", [(64, "exact"), (256, "exact"), (64, "unfused"), (64, "fast")])
def test_config2_gates_toggle_every_30_ticks_inside_batches(rate, batch, mode):
    SR, SPT = rate
    n_strips = 24
    n_ticks = 2 * batch if batch <= 64 else batch
    flags = {"exact": 0, "unfused": abi.FLAG_NO_FUSE, "fast": abi.FLAG_EQ_FAST}[mode]
    ws, mix, srcs, trigs = strips(n_strips, SR)
    og = oracle.OracleGraph(ws)
    g = ws.build(max_ticks_per_run=batch, flags=flags)
    noise = [synth.noise(k, n_ticks * SPT) for k in range(n_strips)]
    amp_ids = [mix + 6 * k + 6 for k in range(n_strips)]
    n_events = 0
    for t0 in range(0, n_ticks, batch):
        n_events += schedule_gates(g, trigs, t0, batch)
        for k, s in enumerate(srcs):
            g.write_source(s, noise[k][t0 * SPT:(t0 + batch) * SPT], batch)
        g.run_ticks(t0, batch)
        got_m = g.read_output(mix, 0, batch, True)
        got_c = g.read_output(mix, 1, batch, True)
        got_amp = [g.read_output(a, 0, batch, True) for a in amp_ids] if mode != "exact" else None
        for kk in range(batch):
            tick = t0 + kk
            for k, tr in enumerate(trigs):
                og.update_params(tr, abi.TriggerParams(1 if gate_open(tick, k) else 0))    # client_update before the tick
            for k, s in enumerate(srcs):
                og.set_source(s, noise[k][tick * SPT:(tick + 1) * SPT])
            og.run_tick(tick)
            sl = slice(kk * 2 * SPT, (kk + 1) * 2 * SPT)
            if mode == "fast":
                for k, a in enumerate(amp_ids):
                    assert_ulp(got_amp[k][sl], og.output(a, 0), 1, f"strip {k} tick {tick}")
            else:
                assert_bit_exact(got_m[sl], og.output(mix, 0), f"master tick {tick}")
                assert_bit_exact(got_c[sl], og.output(mix, 1), f"cue tick {tick}")
                if got_amp:
                    for k, a in enumerate(amp_ids):
                        assert_bit_exact(got_amp[k][sl], og.output(a, 0), f"strip {k} tick {tick}")
    assert n_events >= n_strips * (n_ticks // 30 - 2)    # the toggles really were inside the batches


def test_trigger_with_two_consumers_materialises_its_scheduled_gate():
    # a Trigger read by an Envelope AND a Mixer-bound panner is not folded: its port carries the per-tick gate
    SR, SPT, T = 44100, 735, 12
    ws = Workspace(SR, 60)
    trig = ws.trigger(False); env = ws.envelope(2.0, 20.0, 0.5, 10.0); pan = ws.stereo_panner()
    ws.connect(trig, 0, env, 0); ws.connect(trig, 0, pan, 0); ws.connect(env, 0, pan, 1)
    g = ws.build(max_ticks_per_run=T)
    og = oracle.OracleGraph(ws)
    pattern = [0, 1, 1, 0, 0, 0, 1, 0, 1, 1, 1, 0]
    g.update_params(trig, abi.TriggerParams(pattern[0]))
    for c in range(1, T):
        if pattern[c] != pattern[c - 1]:
            g.schedule_params(trig, c, abi.TriggerParams(pattern[c]))
    g.run_ticks(0, T)
    got_t, got_e, got_p = g.read_output(trig, 0, T, False), g.read_output(env, 0, T, False), g.read_output(pan, 0, T, True)
    for t in range(T):
        og.update_params(trig, abi.TriggerParams(pattern[t]))
        og.run_tick(t)
        assert_bit_exact(got_t[t * SPT:(t + 1) * SPT], og.output(trig, 0), f"trigger tick {t}")
        assert_bit_exact(got_e[t * SPT:(t + 1) * SPT], og.output(env, 0), f"envelope tick {t}")
        assert_bit_exact(got_p[t * 2 * SPT:(t + 1) * 2 * SPT], og.output(pan, 0), f"panner tick {t}")
    # the Trigger keeps the last scheduled params for the next, unscheduled run
    g.run_ticks(T, 2)
    for t in range(T, T + 2):
        og.run_tick(t)
    assert_bit_exact(g.read_output(env, 0, 2, False)[SPT:], og.output(env, 0), "tick after the scheduled run")


@pytest.mark.parametrize("rate", RATES)
def test_updates_of_other_modules_cut_the_run_into_spans(rate):
    """EqThree gains, Mixer faders, Amplifier and Envelope params, an Oscillator's frequency and waveform change at tick
    boundaries inside one submission (the run is launched span by span); state carries across the cuts."""
    SR, SPT = rate
    T = 20
    ws = Workspace(SR, 60)
    src = ws.source_mono(); eq = ws.eq_three(3.0, 0.0, -3.0); pan = ws.stereo_panner()
    trig = ws.trigger(True); env = ws.envelope(10.0, 100.0, 0.7, 50.0); amp = ws.amplifier(1.0, 0.5)
    osc = ws.oscillator(220.0, abi.WAVE_SAW)
    mix = ws.mixer([(0.0, 1.0, False), (-6.0, 0.5, True)])
    ws.connect(src, 0, eq, 0); ws.connect(eq, 0, pan, 0); ws.connect(eq, 0, pan, 1); ws.connect(pan, 0, amp, 0)
    ws.connect(trig, 0, env, 0); ws.connect(env, 0, amp, 1); ws.connect(amp, 0, mix, 0); ws.connect(osc, 1, mix, 1)
    updates = {
        3: [(eq, abi.EqThreeParams(-6.0, 2.0, 4.0))],
        7: [(mix, [abi.MixerChannelParams(-3.0, 0.8, 1), abi.MixerChannelParams(0.0, 1.0, 0)]), (trig, abi.TriggerParams(0))],
        8: [(amp, abi.AmplifierParams(0.7, 0.9))],
        12: [(osc, abi.OscillatorParams(330.0, abi.WAVE_TRIANGLE, 0)), (env, abi.EnvelopeParams(1.0, 10.0, 0.2, 300.0)), (trig, abi.TriggerParams(1))],
        19: [(eq, abi.EqThreeParams(0.0, 0.0, 0.0))],
    }
    x = synth.noise(950, 2 * T * SPT)
    for flags in (0, abi.FLAG_NO_FUSE):
        g = ws.build(max_ticks_per_run=T, flags=flags)
        og = oracle.OracleGraph(ws)
        for run in range(2):
            if run == 0:
                for tick, ups in updates.items():
                    for node, p in ups:
                        g.schedule_params(node, tick, p)
            g.write_source(src, x[run * T * SPT:(run + 1) * T * SPT], T)
            g.run_ticks(run * T, T)
            got_m, got_c = g.read_output(mix, 0, T, True), g.read_output(mix, 1, T, True)
            for kk in range(T):
                tick = run * T + kk
                if run == 0:
                    for node, p in updates.get(kk, []):
                        og.update_params(node, p)
                og.set_source(src, x[tick * SPT:(tick + 1) * SPT])
                og.run_tick(tick)
                sl = slice(kk * 2 * SPT, (kk + 1) * 2 * SPT)
                assert_bit_exact(got_m[sl], og.output(mix, 0), f"master tick {tick} (flags {flags})")
                assert_bit_exact(got_c[sl], og.output(mix, 1), f"cue tick {tick} (flags {flags})")


def test_schedule_errors():
    ws = Workspace(44100, 60)
    t = ws.trigger(False); e = ws.envelope(); ws.connect(t, 0, e, 0)
    g = ws.build(max_ticks_per_run=4)
    with pytest.raises(abi.MxError):
        g.schedule_params(99, 0, abi.TriggerParams(1))                     # no such node
    with pytest.raises(abi.MxError):
        g.schedule_params(t, 0, abi.EnvelopeParams(1.0, 1.0, 0.5, 1.0))    # wrong params for the node
    g.schedule_params(t, 4, abi.TriggerParams(1))                          # tick 4 of a 4-tick run does not exist
    with pytest.raises(abi.MxError) as err:
        g.run_ticks(0, 4)
    assert err.value.code == abi.MX_ERR_INVALID
    g.run_ticks(0, 4)                                                      # the failed run dropped its schedule
    assert not g.read_output(e, 0, 4, False).any()
