"""The default EqThree path on long streams: the SPECULATIVE time-parallel form (k_eq_three_spec) and the pass that proves it
bit-exact chunk by chunk (k_eq_three_repair) -- mixlab_amd/csrc/mx_k_eq_exact.hip.

Every case is compared bit for bit with the oracle's sequential order (src/module/eq_three.rs:58-89), which is pinned on the
reference's golden pair.  Besides plain noise the cases aim at the verification pass itself:

* a warm-up forced far too short (MX_EQ_SPEC_WARM) makes EVERY chunk boundary fail -- the repair pass then re-derives the
  whole stream and the result must still be the sequential order's;
* digital silence / DC after a signal (the poles stall a few ulps from the fixed point, on the side they came from) is the
  input class on which speculation really fails: repairs are counted and the output is still exact;
* NaN / infinity bursts poison the poles of the sequential filter for ever: the speculative lanes after the burst never see
  it, the repair pass must.
"""
import numpy as np
import pytest

import oracle
import synth
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace
from test_gpu_audio_parity import RATES, assert_bit_exact, bits

pytestmark = pytest.mark.gpu


def _eq_graph(sr, gains_list, T, flags=0):
    ws = Workspace(sr, 60)
    srcs, eqs = [], []
    for g3 in gains_list:
        s = ws.source_mono(); e = ws.eq_three(*g3)
        ws.connect(s, 0, e, 0); srcs.append(s); eqs.append(e)
    return ws, srcs, eqs, ws.build(max_ticks_per_run=T, flags=flags)


def _gains(n, seed=90):
    g = synth.uniform(seed, 3 * n, -24.0, 6.0)
    return [tuple(float(v) for v in g[3 * k:3 * k + 3]) for k in range(n)]


def _signals(n, length):
    """Inputs that exercise both outcomes of the speculation."""
    out = []
    for k in range(n):
        x = synth.noise(700 + k, length).copy()
        kind = k % 6
        if kind == 1:                                  # bursts of signal and digital silence
            for a in range(0, length, 30000):
                x[a + 9000:a + 30000] = 0.0
        elif kind == 2:                                # a tone that stops dead, then DC, then noise again
            t = np.arange(length)
            x = (0.8 * np.sin(2 * np.pi * 220.0 * t / 48000.0)).astype(np.float32)
            x[length // 3: length // 2] = 0.0
            x[length // 2: length // 2 + 40000] = np.float32(0.25)
        elif kind == 3:                                # sparse impulses on silence
            x[:] = 0.0; x[::50000] = 1.0
        elif kind == 4:                                # very quiet noise
            x = (x * np.float32(1e-6)).astype(np.float32)
        elif kind == 5:                                # silence from the first sample on (also what a Disconnected input reads)
            x[:] = 0.0
        out.append(np.ascontiguousarray(x, dtype=np.float32))
    return out


@pytest.mark.parametrize("rate", RATES)
@pytest.mark.parametrize("chunks", [0, 2, 7, 64, 130])   # 0 = the planner's own choice
def test_spec_eq_bit_exact_on_live_and_stalling_inputs_state_carried(rate, chunks, monkeypatch):
    SR, SPT = rate
    if chunks:
        monkeypatch.setenv("MX_EQ_SPEC_CHUNKS", str(chunks))
    n, T, runs = 12, 300, 2
    gl = _gains(n)
    ws, srcs, eqs, g = _eq_graph(SR, gl, T)
    sig = _signals(n, runs * T * SPT)
    states = [oracle.eq_three_new(SR) for _ in range(n)]
    for run in range(runs):
        sl = slice(run * T * SPT, (run + 1) * T * SPT)
        for k, s in enumerate(srcs):
            g.write_source(s, sig[k][sl], T)
        g.run_ticks(run * T, T)
        for k, e in enumerate(eqs):
            want = oracle.eq_three_run(states[k], gl[k], sig[k][sl])
            assert_bit_exact(g.read_output(e, 0, T, False), want, f"speculative EQ, instance {k} (signal kind {k % 6}), run {run}")
    ran, repaired = g.eq_spec_stats()
    assert ran > 0, "the speculative kernel did not run"
    # live inputs (kinds 0 and 4) never need a repair; stalling inputs may
    assert repaired < ran


def test_spec_eq_noise_needs_no_repairs():
    SR, SPT = 48000, 800
    n, T = 16, 256
    gl = _gains(n, 91)
    ws, srcs, eqs, g = _eq_graph(SR, gl, T)
    x = [synth.noise(800 + k, T * SPT) for k in range(n)]
    for k, s in enumerate(srcs):
        g.write_source(s, x[k], T)
    g.run_ticks(0, T)
    for k, e in enumerate(eqs):
        assert_bit_exact(g.read_output(e, 0, T, False), oracle.eq_three_run(oracle.eq_three_new(SR), gl[k], x[k]), f"instance {k}")
    ran, repaired = g.eq_spec_stats()
    assert ran >= 2 * n and repaired == 0, f"{repaired} of {ran} chunks needed a repair on plain noise"


@pytest.mark.parametrize("warm", [16, 64, 256])
def test_spec_eq_with_a_useless_warm_up_is_repaired_to_the_sequential_order(warm, monkeypatch):
    monkeypatch.setenv("MX_EQ_SPEC_WARM", str(warm))
    monkeypatch.setenv("MX_EQ_SPEC_CHUNKS", "24")
    SR, SPT = 44100, 735
    n, T = 6, 120
    gl = _gains(n, 92)
    ws, srcs, eqs, g = _eq_graph(SR, gl, T)
    x = [synth.noise(810 + k, 2 * T * SPT) for k in range(n)]
    states = [oracle.eq_three_new(SR) for _ in range(n)]
    for run in range(2):
        sl = slice(run * T * SPT, (run + 1) * T * SPT)
        for k, s in enumerate(srcs):
            g.write_source(s, x[k][sl], T)
        g.run_ticks(run * T, T)
        for k, e in enumerate(eqs):
            assert_bit_exact(g.read_output(e, 0, T, False), oracle.eq_three_run(states[k], gl[k], x[k][sl]), f"instance {k} run {run}")
    ran, repaired = g.eq_spec_stats()
    assert repaired >= ran // 2, f"only {repaired} of {ran} chunks were repaired: the short warm-up was expected to fail nearly everywhere"


def test_spec_eq_nan_and_infinity_poison_the_poles_exactly_like_the_sequential_filter():
    SR, SPT = 48000, 800
    T = 200
    gl = [(3.0, -2.0, 1.0), (0.0, 0.0, 0.0)]
    ws, srcs, eqs, g = _eq_graph(SR, gl, T)
    x0 = synth.noise(820, T * SPT).copy(); x0[40000] = np.float32("nan")
    x1 = synth.noise(821, T * SPT).copy(); x1[70000] = np.float32("inf"); x1[70001] = np.float32("-inf")
    for s, x in zip(srcs, (x0, x1)):
        g.write_source(s, x, T)
    g.run_ticks(0, T)
    for k, (e, x) in enumerate(zip(eqs, (x0, x1))):
        want = oracle.eq_three_run(oracle.eq_three_new(SR), gl[k], x)
        got = g.read_output(e, 0, T, False)
        # NaN sign / payload bits are the ISA's business (x86 makes its default NaN negative, gfx950 positive): NaNs must sit
        # at the same samples, everything else is compared bit for bit
        assert np.array_equal(np.isnan(got), np.isnan(want)), f"poisoned instance {k}: NaNs at different samples"
        ok = ~np.isnan(want)
        assert_bit_exact(got[ok], want[ok], f"poisoned instance {k}")
        assert np.isnan(got[-1])


@pytest.mark.parametrize("chunks", ["0", "96", "400"])
def test_spec_eq_programme_with_many_silences_islands_repaired_side_by_side(chunks, monkeypatch):
    """Programme that falls silent and comes back, again and again (what a desk's strips carry): every onset of digital silence fails the
    boundaries of the chunks behind it -- an island; the repair wave walks the islands of a strip side by side, one lane each, and checks
    in stream order that each started from what it assumed.  Four shapes: gaps seconds apart (islands far from each other), gaps so
    close that an island is still apart from the speculative run when the next begins (the in-order fallback that rewrites everything),
    one long silence, and silences broken by single-sample clicks.  State carried over two runs; also with a folded Envelope + Amplifier."""
    if chunks != "0":
        monkeypatch.setenv("MX_EQ_SPEC_CHUNKS", chunks)
    SR, SPT = 48000, 800
    n, T, runs = 8, 400, 2
    gl = _gains(n, 93)
    L = runs * T * SPT
    sig = []
    for k in range(n):
        x = synth.noise(900 + k, L).copy()
        kind = k % 4
        if kind == 0:                                   # 0.9 s of programme, 0.6 s of silence
            for a in range(20000, L, 72000):
                x[a:a + 28800] = 0.0
        elif kind == 1:                                 # short bursts between silences: 2 400 samples of signal, 9 000 of silence
            for a in range(5000, L, 11400):
                x[a:a + 9000] = 0.0
        elif kind == 2:                                 # one long silence in the middle
            x[L // 5: 4 * L // 5] = 0.0
        else:                                           # silences with a click inside
            for a in range(10000, L, 50000):
                x[a:a + 30000] = 0.0
                x[a + 15000] = np.float32(0.5)
        sig.append(np.ascontiguousarray(x, dtype=np.float32))
    ws, srcs, eqs, g = _eq_graph(SR, gl, T)
    states = [oracle.eq_three_new(SR) for _ in range(n)]
    for run in range(runs):
        sl = slice(run * T * SPT, (run + 1) * T * SPT)
        for k, s in enumerate(srcs):
            g.write_source(s, sig[k][sl], T)
        g.run_ticks(run * T, T)
        for k, e in enumerate(eqs):
            want = oracle.eq_three_run(states[k], gl[k], sig[k][sl])
            assert_bit_exact(g.read_output(e, 0, T, False), want, f"instance {k} (shape {k % 4}), run {run}, chunks {chunks}")
    ran, repaired = g.eq_spec_stats()
    assert repaired > 0, "no chunk needed a repair: the islands were not exercised"


@pytest.mark.parametrize("rate", RATES)
@pytest.mark.parametrize("chunks", ["0", "24"])
def test_spec_eq_chunks_shorter_than_the_warm_up(rate, chunks, monkeypatch):
    """A short submission (24 ticks) cut into one-tick chunks: 735 / 800 samples against a warm-up of 1 280.  The chunks whose warm-up
    window would reach the stream's start warm up from there, from the carried state (exact); the others speculate as always.  Every
    epilogue form (the tiled kernel and -- with a control BUFFER -- the direct one), two submissions so that the state carries."""
    SR, SPT = rate
    T = 24
    monkeypatch.setenv("MX_EQ_SPEC_CHUNKS", chunks)
    ws = Workspace(SR, 60)
    src = [ws.source_mono() for _ in range(3)]; ctl_src = ws.source_mono()
    eq = [ws.eq_three(3.0 - k, -2.0 + k, 1.5 * k) for k in range(3)]
    pan = [ws.stereo_panner() for _ in range(2)]
    for s_, e in zip(src, eq):
        ws.connect(s_, 0, e, 0)
    for k in range(2):
        ws.connect(eq[k + 1], 0, pan[k], 0); ws.connect(eq[k + 1], 0, pan[k], 1)
    amp_buf = ws.amplifier(0.9, 0.6); ws.connect(pan[0], 0, amp_buf, 0); ws.connect(ctl_src, 0, amp_buf, 1)
    trig = ws.trigger(True); env = ws.envelope(5.0, 80.0, 0.6, 40.0)
    amp_env = ws.amplifier(1.0, 0.5); ws.connect(pan[1], 0, amp_env, 0); ws.connect(trig, 0, env, 0); ws.connect(env, 0, amp_env, 1)
    mix = ws.mixer([(0.0, 1.0, True)]); ws.connect(amp_env, 0, mix, 0)
    g = ws.build(max_ticks_per_run=T)
    og = oracle.OracleGraph(ws)
    outs = {"eq": (eq[0], 0, False), "amp_buf": (amp_buf, 0, True), "master": (mix, 0, True)}
    for run in range(2):
        x = [synth.noise(860 + 3 * run + k, T * SPT) for k in range(3)]
        ctl = np.abs(synth.noise(870 + run, T * SPT))
        for s_, v in zip(src, x):
            g.write_source(s_, v, T)
        g.write_source(ctl_src, ctl, T)
        g.schedule_params(trig, 7, abi.TriggerParams(run)); g.schedule_params(trig, 15, abi.TriggerParams(1 - run))
        g.run_ticks(run * T, T)
        got = {k: g.read_output(nd, port, T, st) for k, (nd, port, st) in outs.items()}
        for t in range(T):
            if t == 7: og.update_params(trig, abi.TriggerParams(run))
            if t == 15: og.update_params(trig, abi.TriggerParams(1 - run))
            for s_, v in zip(src, x):
                og.set_source(s_, v[t * SPT:(t + 1) * SPT])
            og.set_source(ctl_src, ctl[t * SPT:(t + 1) * SPT])
            og.run_tick(run * T + t)
            for k, (nd, port, st) in outs.items():
                w = og.output(nd, port)
                assert_bit_exact(got[k][t * w.size:(t + 1) * w.size], w, f"run {run} {k} tick {t}")
    ran, _ = g.eq_spec_stats()
    # one-tick chunks: the plan really is shorter than the warm-up (at 44.1 kHz the EQ with the inline Envelope takes chunks of FOUR ticks -- whole ticks in
    # multiples of four samples: 2 940 -- the other two keep theirs)
    assert ran >= (2 * 3 * 20 if SPT % 4 == 0 else 2 * (2 * 20 + T // 4))


@pytest.mark.parametrize("rate", RATES)
def test_spec_eq_every_fused_epilogue_matches_the_oracle_graph(rate):
    """The chunk loop's four epilogue modes on long streams: plain EQ; EQ -> Panner (stereo store); -> Amplifier with a control
    BUFFER; -> Amplifier with a Disconnected control; -> Amplifier whose control is an inline Envelope (per-tick states), the last
    one stored as one float per frame for the Mixer."""
    SR, SPT = rate
    T = 240
    ws = Workspace(SR, 60)
    src = [ws.source_mono() for _ in range(5)]
    ctl_src = ws.source_mono()
    eq = [ws.eq_three(4.0 - k, -1.0 + 0.5 * k, 2.0 - k) for k in range(5)]
    for s, e in zip(src, eq):
        ws.connect(s, 0, e, 0)
    pan = [ws.stereo_panner() for _ in range(4)]
    for k in range(4):
        ws.connect(eq[k + 1], 0, pan[k], 0); ws.connect(eq[k + 1], 0, pan[k], 1)
    amp_buf = ws.amplifier(0.9, 0.6); ws.connect(pan[1], 0, amp_buf, 0); ws.connect(ctl_src, 0, amp_buf, 1)
    amp_dis = ws.amplifier(1.2, 0.3); ws.connect(pan[2], 0, amp_dis, 0)
    trig = ws.trigger(True); env = ws.envelope(5.0, 80.0, 0.6, 40.0)
    amp_env = ws.amplifier(1.0, 0.5); ws.connect(pan[3], 0, amp_env, 0); ws.connect(trig, 0, env, 0); ws.connect(env, 0, amp_env, 1)
    mix = ws.mixer([(0.0, 1.0, True)]); ws.connect(amp_env, 0, mix, 0)
    g = ws.build(max_ticks_per_run=T)
    og = oracle.OracleGraph(ws)
    x = [synth.noise(830 + k, T * SPT) for k in range(5)]
    ctl = np.abs(synth.noise(840, T * SPT))
    for s, v in zip(src, x):
        g.write_source(s, v, T)
    g.write_source(ctl_src, ctl, T)
    # the Trigger closes at tick 100 and re-opens at tick 180 inside the run
    g.schedule_params(trig, 100, abi.TriggerParams(0)); g.schedule_params(trig, 180, abi.TriggerParams(1))
    g.run_ticks(0, T)
    outs = {"eq": (eq[0], 0, False), "pan": (pan[0], 0, True), "amp_buf": (amp_buf, 0, True), "amp_dis": (amp_dis, 0, True), "master": (mix, 0, True)}
    got = {k: g.read_output(nd, port, T, st) for k, (nd, port, st) in outs.items()}
    for t in range(T):
        if t == 100: og.update_params(trig, abi.TriggerParams(0))
        if t == 180: og.update_params(trig, abi.TriggerParams(1))
        for s, v in zip(src, x):
            og.set_source(s, v[t * SPT:(t + 1) * SPT])
        og.set_source(ctl_src, ctl[t * SPT:(t + 1) * SPT])
        og.run_tick(t)
        for k, (nd, port, st) in outs.items():
            w = og.output(nd, port)
            sl = slice(t * w.size, (t + 1) * w.size)
            assert_bit_exact(got[k][sl], w, f"{k} tick {t}")
    ran, _ = g.eq_spec_stats()
    assert ran > 0


@pytest.mark.parametrize("T,chunks", [(30, "0"), (50, "0"), (101, "0"), (33, "0"), (257, "8"), (257, "2"), (148, "3")])
def test_spec_eq_streams_that_are_not_whole_pieces_of_four_samples(T, chunks, monkeypatch):
    """735 T frames at 44.1 kHz are a multiple of four only when T is: the tiled kernel takes the frames up to the last multiple of four and the proof / repair
    kernel walks the one to three samples left from the exact state it ends up with, through the same epilogue (until round 4 such a submission fell back to the
    direct-load kernel).  Plain EQ, EQ -> Panner -> Amplifier with an inline Envelope (ragged ticks) and with a control buffer; three runs, state carried.
    Forced chunk counts give chunks of 36 and 132 ticks, whose rows sit 28 samples into their lines: the largest shift of the aligned-row grid (a first version
    assumed 24, the largest one the planner's own chunks have -- tools/stress_eq_shapes.py found it)."""
    monkeypatch.setenv("MX_EQ_SPEC_CHUNKS", chunks)
    SR, SPT = 44100, 735
    ws = Workspace(SR, 60)
    src = [ws.source_mono() for _ in range(3)]; ctl_src = ws.source_mono()
    eq = [ws.eq_three(1.0 + k, -2.0 * k, 0.5 * k) for k in range(3)]
    for s_, e in zip(src, eq):
        ws.connect(s_, 0, e, 0)
    pan = [ws.stereo_panner() for _ in range(2)]
    for k in range(2):
        ws.connect(eq[k + 1], 0, pan[k], 0); ws.connect(eq[k + 1], 0, pan[k], 1)
    trig = ws.trigger(True); env = ws.envelope(5.0, 80.0, 0.6, 40.0); amp_env = ws.amplifier(1.1, 0.7)
    ws.connect(pan[0], 0, amp_env, 0); ws.connect(trig, 0, env, 0); ws.connect(env, 0, amp_env, 1)
    amp_buf = ws.amplifier(0.9, 0.4); ws.connect(pan[1], 0, amp_buf, 0); ws.connect(ctl_src, 0, amp_buf, 1)
    g = ws.build(max_ticks_per_run=T)
    og = oracle.OracleGraph(ws)
    outs = [(eq[0], False), (amp_env, True), (amp_buf, True)]
    for run in range(3):
        x = [synth.noise(1200 + 5 * run + k, T * SPT) for k in range(3)]
        cx = np.abs(synth.noise(1250 + run, T * SPT))
        for s_, v in zip(src, x):
            g.write_source(s_, v, T)
        g.write_source(ctl_src, cx, T)
        g.schedule_params(trig, T // 3, abi.TriggerParams(run % 2)); g.schedule_params(trig, 2 * T // 3, abi.TriggerParams(1 - run % 2))
        g.run_ticks(run * T, T)
        got = [g.read_output(nd, 0, T, st) for nd, st in outs]
        for t in range(T):
            if t == T // 3: og.update_params(trig, abi.TriggerParams(run % 2))
            if t == 2 * T // 3: og.update_params(trig, abi.TriggerParams(1 - run % 2))
            for s_, v in zip(src, x):
                og.set_source(s_, v[t * SPT:(t + 1) * SPT])
            og.set_source(ctl_src, cx[t * SPT:(t + 1) * SPT])
            og.run_tick(run * T + t)
            for k, (nd, _st) in enumerate(outs):
                w = og.output(nd, 0)
                assert_bit_exact(got[k][t * w.size:(t + 1) * w.size], w, f"T {T} run {run} output {k} tick {t}")
    assert g.eq_spec_stats()[0] > 0


@pytest.mark.parametrize("rate", RATES)
@pytest.mark.parametrize("chunks", ["0", "130"])
def test_spec_eq_amplifier_modulated_by_a_buffer_takes_the_control_through_its_own_tile(rate, chunks, monkeypatch):
    """EqThree -> StereoPanner -> Amplifier whose control is another module's OUTPUT (an oscillator as LFO shared by several strips, a source per strip, an
    Envelope that cannot be folded in because two Amplifiers read it): the tiled speculative kernel stages the control of every super-block in a second
    tile beside the input's (until round 4 this mode kept the direct-load kernel: 7x slower).  Long runs, state carried, stereo and Mixer-only (one float per
    frame) stores, the planner's chunking and a forced one whose last chunk is ragged; MX_EQ_CTL_DIRECT=1 (the old form) gives the same bits."""
    SR, SPT = rate
    T = 160
    monkeypatch.setenv("MX_EQ_SPEC_CHUNKS", chunks)
    ws = Workspace(SR, 60)
    lfo = ws.oscillator(3.0, abi.WAVE_TRIANGLE)
    trig = ws.trigger(True); env = ws.envelope(5.0, 60.0, 0.5, 30.0); ws.connect(trig, 0, env, 0)
    src, ctl_src, outs = [], ws.source_mono(), []
    mix = ws.mixer([(0.0, 1.0, k % 2 == 0) for k in range(3)])
    for k in range(6):
        s = ws.source_mono(); e = ws.eq_three(2.0 - k, 0.5 * k, -1.0 + k); p = ws.stereo_panner(); a = ws.amplifier(0.8 + 0.1 * k, 0.2 + 0.15 * k)
        ws.connect(s, 0, e, 0); ws.connect(e, 0, p, 0); ws.connect(e, 0, p, 1); ws.connect(p, 0, a, 0)
        ws.connect([lfo, lfo, ctl_src, env, env, lfo][k], 0, a, 1)         # env feeds two Amplifiers: not folded, a buffer like the others
        if k < 3:
            ws.connect(a, 0, mix, k)                                        # read by a Mixer only: stored as one float per frame
        else:
            outs.append(a)                                                  # a terminal: stored as stereo
        src.append(s)
    g = ws.build(max_ticks_per_run=T)
    og = oracle.OracleGraph(ws)
    for run in range(2):
        x = [synth.noise(900 + 7 * run + k, T * SPT) for k in range(6)]
        cx = synth.noise(950 + run, T * SPT)
        for s_, v in zip(src, x):
            g.write_source(s_, v, T)
        g.write_source(ctl_src, cx, T)
        g.schedule_params(trig, 40, abi.TriggerParams(run)); g.schedule_params(trig, 90, abi.TriggerParams(1 - run))
        g.run_ticks(run * T, T)
        got = [g.read_output(mix, 0, T, True), g.read_output(mix, 1, T, True)] + [g.read_output(a, 0, T, True) for a in outs]
        for t in range(T):
            if t == 40: og.update_params(trig, abi.TriggerParams(run))
            if t == 90: og.update_params(trig, abi.TriggerParams(1 - run))
            for s_, v in zip(src, x):
                og.set_source(s_, v[t * SPT:(t + 1) * SPT])
            og.set_source(ctl_src, cx[t * SPT:(t + 1) * SPT])
            og.run_tick(run * T + t)
            want = [og.output(mix, 0), og.output(mix, 1)] + [og.output(a, 0) for a in outs]
            for k, (gv, w) in enumerate(zip(got, want)):
                assert_bit_exact(gv[t * w.size:(t + 1) * w.size], w, f"run {run} output {k} tick {t}")
    assert g.eq_spec_stats()[0] > 0


@pytest.mark.parametrize("env_p", [(25.0, 500.0, 0.8, 200.0), (0.0, 100.0, 0.5, 50.0), (5.0, 0.0, 0.7, 0.0), (10.0, 40.0, 1.5, 30.0),
                                   (3.0, 20.0, -0.25, 15.0), (1e-3, 1e-3, 0.0, 1e-3), (400.0, 3000.0, 0.3, 2500.0)])
@pytest.mark.parametrize("rate", RATES)
def test_spec_eq_inline_envelope_with_unusual_parameters_matches_the_oracle(env_p, rate):
    """The inline Envelope's branch-free form assumes finite parameters with non-negative slopes and a non-negative
    off_amplitude; everything else (zero attack / decay / release times -> infinite slopes, sustain outside [0, 1]) must fall
    back to the general form and still be the reference's arithmetic.  Gates toggle inside the batch."""
    (SR, SPT), T = rate, 240
    ws = Workspace(SR, 60)
    src = ws.source_mono(); eq = ws.eq_three(2.0, -1.0, 3.0); pan = ws.stereo_panner()
    trig = ws.trigger(False); env = ws.envelope(*env_p); amp = ws.amplifier(0.9, 0.8)
    ws.connect(src, 0, eq, 0); ws.connect(eq, 0, pan, 0); ws.connect(eq, 0, pan, 1); ws.connect(pan, 0, amp, 0)
    ws.connect(trig, 0, env, 0); ws.connect(env, 0, amp, 1)
    g = ws.build(max_ticks_per_run=T)
    og = oracle.OracleGraph(ws)
    x = synth.noise(870, T * SPT)
    g.write_source(src, x, T)
    toggles = {7: 1, 45: 0, 46: 1, 47: 0, 120: 1, 200: 0}
    for t, v in toggles.items():
        g.schedule_params(trig, t, abi.TriggerParams(v))
    g.run_ticks(0, T)
    got = g.read_output(amp, 0, T, True)
    for t in range(T):
        if t in toggles:
            og.update_params(trig, abi.TriggerParams(toggles[t]))
        og.set_source(src, x[t * SPT:(t + 1) * SPT])
        og.run_tick(t)
        w = og.output(amp, 0)
        ok = ~np.isnan(w)
        assert np.array_equal(np.isnan(got[t * 2 * SPT:(t + 1) * 2 * SPT]), np.isnan(w)), f"tick {t}: NaNs at different samples"
        assert_bit_exact(got[t * 2 * SPT:(t + 1) * 2 * SPT][ok], w[ok], f"tick {t} (envelope {env_p})")
    assert g.eq_spec_stats()[0] > 0


@pytest.mark.parametrize("mode", ["1", "2", "3"])
def test_spec_eq_repair_paths_that_live_signals_never_take_still_give_the_sequential_order(mode, monkeypatch):
    """Two paths of the repair pass are reached only when the f32 outputs of two standing states a few f64 ulps apart differ -- the FILL of a
    constant chunk from the true state, and the in-order fallback (rewrite everything) behind an island that ended apart from the speculative
    run.  MX_EQ_REPAIR_TEST forces them (1: fill, 2: every island ends apart, 3: both): they may only cost time, never a bit.  Programme with
    silences, DC plateaus and a strip muted to the end, state carried over two runs, plain EQs and the folded Envelope + Amplifier."""
    monkeypatch.setenv("MX_EQ_REPAIR_TEST", mode)
    monkeypatch.setenv("MX_EQ_SPEC_CHUNKS", "96")
    test_spec_eq_programme_with_many_silences_islands_repaired_side_by_side("96", monkeypatch)
    SR, SPT, T = 48000, 800, 240
    ws = Workspace(SR, 60)
    src = ws.source_mono(); eq = ws.eq_three(2.0, -1.0, 3.0); pan = ws.stereo_panner()
    trig = ws.trigger(True); env = ws.envelope(5.0, 80.0, 0.6, 40.0); amp = ws.amplifier(0.9, 0.8)
    ws.connect(src, 0, eq, 0); ws.connect(eq, 0, pan, 0); ws.connect(eq, 0, pan, 1); ws.connect(pan, 0, amp, 0); ws.connect(trig, 0, env, 0); ws.connect(env, 0, amp, 1)
    g = ws.build(max_ticks_per_run=T)
    og = oracle.OracleGraph(ws)
    # programme with a positive offset: both cascades reach every silence from ABOVE, where a lane that warmed up from zeros stands BELOW (two
    # trajectories from the same side stall on the same value and prove themselves -- half of all silences of zero-mean programme do)
    x = (np.float32(0.5) + np.float32(0.3) * synth.noise(950, 2 * T * SPT)).astype(np.float32)
    x[20000:60000] = 0.0; x[120000:150000] = 0.0; x[150000:170000] = np.float32(0.25); x[180000:180003] = 0.0; x[250000:] = 0.0   # two islands far apart in run 0, one to the end in run 1
    for run in range(2):
        sl = slice(run * T * SPT, (run + 1) * T * SPT)
        g.write_source(src, x[sl], T)
        g.schedule_params(trig, 50, abi.TriggerParams(0)); g.schedule_params(trig, 130, abi.TriggerParams(1))
        g.run_ticks(run * T, T)
        got = g.read_output(amp, 0, T, True)
        for t in range(T):
            if t == 50: og.update_params(trig, abi.TriggerParams(0))
            if t == 130: og.update_params(trig, abi.TriggerParams(1))
            og.set_source(src, x[run * T * SPT + t * SPT: run * T * SPT + (t + 1) * SPT])
            og.run_tick(run * T + t)
            assert_bit_exact(got[t * 2 * SPT:(t + 1) * 2 * SPT], og.output(amp, 0), f"run {run} tick {t} (MX_EQ_REPAIR_TEST={mode})")
    st = g.eq_repair_stats()
    if mode in ("1", "3"):
        assert st["fill_steps_16"] > 0, st
    if mode in ("2", "3"):
        assert st["in_order_walks"] > 0, st


@pytest.mark.parametrize("sb", ["16", "32", "321"])
def test_spec_eq_every_tile_shape_is_bit_exact(sb, monkeypatch):
    """The tiled kernel moves WHOLE 128-byte lines through ONE tile per wave (MX_EQ_SPEC_SB=321) where three or four waves share a SIMD -- the
    benchmarked shape, test_gpu_full_size.py -- and through TWO tiles (32) where at most two do, which is every small graph of this suite; half
    lines through two tiles (16, round 3's default) is the contracted order's two-tile shape.  Each forced here over the same graphs."""
    monkeypatch.setenv("MX_EQ_SPEC_SB", sb)
    test_spec_eq_every_fused_epilogue_matches_the_oracle_graph((48000, 800))
    test_spec_eq_bit_exact_on_live_and_stalling_inputs_state_carried((48000, 800), 0, monkeypatch)


def test_one_lane_per_instance_kernel_still_matches_the_oracle():
    """Short streams of FEW instances now take the split-cascade path (k_eq_three_poles + k_eq_three_emit); k_eq_three_exact -- one lane
    per instance -- remains the path of short streams with thousands of instances (bench.py's 10 240-strip real-time graph).  The
    switch is read once per process, so the one-lane kernel is exercised by a child pytest with MX_EQ_POLES_BELOW=0 over the tests that
    pin EqThree: the reference's golden pair, state carried across calls, and the config-2 strips at both rates."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MX_EQ_POLES_BELOW="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(here, "test_gpu_audio_parity.py"), "-k", "eq_three or config2_strips_exact or config1"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(here))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
