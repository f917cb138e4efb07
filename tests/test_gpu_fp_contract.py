"""MX_FLAG_FP_CONTRACT: the CONTRACTED order of the f64 kernels (include/mixlab_gpu.h) -- the reference's expressions with every
multiply fused into the add that consumes it (EqThree eq_three.rs:76-88,117-124; Envelope envelope.rs:46-47; Amplifier
amplifier.rs:71-73; the build-specified Fir / Resample accumulation).  Not the reference's bits: within 1 ULP of them.

Two kinds of checks:

* **the contracted order is an order**: the device equals the oracle's contract mode (explicit fma(), oracle/mixlab_oracle.c
  "CONTRACT MODE") BIT FOR BIT -- on every shape tests/test_gpu_eq_exact_spec.py puts the speculative kernel, its proof and its
  repair pass through (those tests are re-run here with the flag set and the oracle switched), on the config-2 graph fused, unfused
  and ticked, and on the FIR / resampler chain;
* **how far it is from the reference's order**: every f32 output within 1 ULP of the exact oracle, with the number of differing
  samples counted and asserted small -- the reference's golden pair (expected: none), config 2 at 1024 strips, config 3.
  (A bus downstream of a Mixer is an f32 sum of such values: it is bounded by one ULP of its largest addend, not of itself.)
"""
import numpy as np
import pytest

import oracle
import synth
import test_gpu_eq_exact_spec as spec
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace
from test_gpu_audio_parity import RATES, _golden, assert_bit_exact, bits, strips
from test_gpu_fir_resample import polyphase_table, reverb_taps

pytestmark = pytest.mark.gpu


@pytest.fixture
def contracted(monkeypatch):
    """Every graph built inside the test carries MX_FLAG_FP_CONTRACT; the oracle evaluates in its contract mode."""
    orig = Workspace.build

    def build(self, max_ticks_per_run=1, flags=0, device=-1, stream=None):
        return orig(self, max_ticks_per_run, flags | abi.FLAG_FP_CONTRACT, device, stream)

    monkeypatch.setattr(Workspace, "build", build)
    with oracle.fp_contract():
        yield


# ---------------------------------------------------------------------------------------------------------------------------
# the contracted order is an order: bit for bit against the oracle's contract mode, on every shape of the exact-order suite
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rate", RATES)
@pytest.mark.parametrize("chunks", [0, 7, 130])
def test_contracted_spec_eq_live_and_stalling_inputs_state_carried(rate, chunks, contracted, monkeypatch):
    spec.test_spec_eq_bit_exact_on_live_and_stalling_inputs_state_carried(rate, chunks, monkeypatch)


def test_contracted_spec_eq_noise_needs_no_repairs(contracted):
    spec.test_spec_eq_noise_needs_no_repairs()


@pytest.mark.parametrize("warm", [16, 256])
def test_contracted_spec_eq_with_a_useless_warm_up_is_repaired(warm, contracted, monkeypatch):
    spec.test_spec_eq_with_a_useless_warm_up_is_repaired_to_the_sequential_order(warm, monkeypatch)


def test_contracted_spec_eq_nan_and_infinity(contracted):
    spec.test_spec_eq_nan_and_infinity_poison_the_poles_exactly_like_the_sequential_filter()


@pytest.mark.parametrize("chunks", ["0", "96"])
def test_contracted_spec_eq_programme_with_many_silences(chunks, contracted, monkeypatch):
    spec.test_spec_eq_programme_with_many_silences_islands_repaired_side_by_side(chunks, monkeypatch)


@pytest.mark.parametrize("rate", RATES)
@pytest.mark.parametrize("chunks", ["0", "24"])
def test_contracted_spec_eq_chunks_shorter_than_the_warm_up(rate, chunks, contracted, monkeypatch):
    spec.test_spec_eq_chunks_shorter_than_the_warm_up(rate, chunks, monkeypatch)


@pytest.mark.parametrize("rate", RATES)
def test_contracted_spec_eq_every_fused_epilogue(rate, contracted):
    spec.test_spec_eq_every_fused_epilogue_matches_the_oracle_graph(rate)


@pytest.mark.parametrize("T,chunks", [(50, "0"), (257, "8")])
def test_contracted_spec_eq_streams_that_are_not_whole_pieces_of_four_samples(T, chunks, contracted, monkeypatch):
    spec.test_spec_eq_streams_that_are_not_whole_pieces_of_four_samples(T, chunks, monkeypatch)


@pytest.mark.parametrize("rate", RATES)
def test_contracted_spec_eq_amplifier_modulated_by_a_buffer(rate, contracted, monkeypatch):
    spec.test_spec_eq_amplifier_modulated_by_a_buffer_takes_the_control_through_its_own_tile(rate, "0", monkeypatch)


@pytest.mark.parametrize("env_p", [(25.0, 500.0, 0.8, 200.0), (0.0, 100.0, 0.5, 50.0), (10.0, 40.0, 1.5, 30.0), (3.0, 20.0, -0.25, 15.0)])
@pytest.mark.parametrize("rate", RATES)
def test_contracted_spec_eq_inline_envelope_with_unusual_parameters(env_p, rate, contracted):
    spec.test_spec_eq_inline_envelope_with_unusual_parameters_matches_the_oracle(env_p, rate)


def _run_config2(n_strips, sr, spt, T, runs, flags, batch, noise):
    """config-2 strips, gates set per strip between the runs; returns every strip's Amplifier output (unfused only) and both buses"""
    ws, mix, srcs, trigs = strips(n_strips, sr)
    g = ws.build(max_ticks_per_run=batch, flags=flags)
    res_m, res_c, res_s = [], [], []
    for run in range(runs):
        for k, tr in enumerate(trigs):
            g.update_params(tr, abi.TriggerParams(1 if ((run * T + k) // 3) % 2 else 0))
        for t0 in range(0, T, batch):
            a = (run * T + t0) * spt
            for k, s in enumerate(srcs):
                g.write_source(s, noise[k][a:a + batch * spt], batch)
            g.run_ticks(run * T + t0, batch)
            res_m.append(g.read_output(mix, 0, batch, True)); res_c.append(g.read_output(mix, 1, batch, True))
            if flags & abi.FLAG_NO_FUSE:
                res_s.append(np.stack([g.read_output(mix + 6 * k + 6, 0, batch, True) for k in range(n_strips)]))
    return ws, mix, srcs, trigs, np.concatenate(res_m), np.concatenate(res_c), (np.concatenate(res_s, axis=1) if res_s else None)


def _oracle_config2(ws, mix, srcs, trigs, n_strips, spt, T, runs, noise, want_strips):
    og = oracle.OracleGraph(ws)
    m, c, st = [], [], []
    for run in range(runs):
        for k, tr in enumerate(trigs):
            og.update_params(tr, abi.TriggerParams(1 if ((run * T + k) // 3) % 2 else 0))
        for kk in range(T):
            tick = run * T + kk
            for k, s in enumerate(srcs):
                og.set_source(s, noise[k][tick * spt:(tick + 1) * spt])
            og.run_tick(tick)
            m.append(og.output(mix, 0).copy()); c.append(og.output(mix, 1).copy())
            if want_strips:
                st.append(np.stack([og.output(mix + 6 * k + 6, 0).copy() for k in range(n_strips)]))
    return np.concatenate(m), np.concatenate(c), (np.concatenate(st, axis=1) if st else None)


@pytest.mark.parametrize("rate", RATES)
@pytest.mark.parametrize("shape", ["fused-batched", "fused-ticked", "unfused-batched"])
def test_contracted_config2_strips_equal_the_oracles_contract_mode_bit_for_bit(rate, shape):
    """The contracted order does not depend on how the work is cut: folded into the EqThree kernel or module by module (the stand-alone
    Envelope / Amplifier kernels contract the same expressions), one submission or a tick at a time -- always the oracle's contract mode."""
    SR, SPT = rate
    n, T, runs = 24, 12, 2
    noise = [synth.noise(300 + k, runs * T * SPT) for k in range(n)]
    flags = abi.FLAG_FP_CONTRACT | (abi.FLAG_NO_FUSE if shape.startswith("unfused") else 0)
    ws, mix, srcs, trigs, got_m, got_c, got_s = _run_config2(n, SR, SPT, T, runs, flags, 1 if shape.endswith("ticked") else T, noise)
    with oracle.fp_contract():
        want_m, want_c, want_s = _oracle_config2(ws, mix, srcs, trigs, n, SPT, T, runs, noise, got_s is not None)
    assert_bit_exact(got_m, want_m, f"Master ({shape})")
    assert_bit_exact(got_c, want_c, f"Cue ({shape})")
    if got_s is not None:
        assert_bit_exact(got_s.ravel(), want_s.ravel(), "strip outputs")


def test_contracted_long_submission_gates_toggling_inside_it_bit_for_bit():
    """bench.py's `fp_contract` leg in small: one submission of 512 ticks at 48 kHz, every gate toggling every 30 ticks inside it -- the
    tiled speculative kernel with the branch-free inline Envelope, compiled for the contracted order, proven by the same repair pass."""
    from test_gpu_schedule import gate_open, schedule_gates
    SR, SPT, T, n_strips = 48000, 800, 512, 8
    ws, mix, srcs, trigs = strips(n_strips, SR)
    g = ws.build(max_ticks_per_run=T, flags=abi.FLAG_FP_CONTRACT)
    noise = [synth.noise(400 + k, T * SPT) for k in range(n_strips)]
    schedule_gates(g, trigs, 0, T)
    for k, s in enumerate(srcs):
        g.write_source(s, noise[k], T)
    g.run_ticks(0, T)
    ran, repaired = g.eq_spec_stats()
    assert ran >= 8 * n_strips and repaired == 0
    got_m, got_c = g.read_output(mix, 0, T, True), g.read_output(mix, 1, T, True)
    with oracle.fp_contract():
        og = oracle.OracleGraph(ws)
        for tick in range(T):
            for k, tr in enumerate(trigs):
                if tick == 0 or gate_open(tick, k) != gate_open(tick - 1, k):
                    og.update_params(tr, abi.TriggerParams(1 if gate_open(tick, k) else 0))
            for k, s in enumerate(srcs):
                og.set_source(s, noise[k][tick * SPT:(tick + 1) * SPT])
            og.run_tick(tick)
            sl = slice(tick * 2 * SPT, (tick + 1) * 2 * SPT)
            assert np.array_equal(bits(got_m[sl]), bits(og.output(mix, 0))), f"master differs in tick {tick}"
            assert np.array_equal(bits(got_c[sl]), bits(og.output(mix, 1))), f"cue differs in tick {tick}"


def test_fp_contract_is_a_mode_of_the_exact_order_kernels_not_of_the_scan():
    ws = Workspace(48000, 60)
    s = ws.source_mono(); e = ws.eq_three(0.0, 0.0, 0.0); ws.connect(s, 0, e, 0)
    with pytest.raises(abi.MxError):
        ws.build(max_ticks_per_run=4, flags=abi.FLAG_FP_CONTRACT | abi.FLAG_EQ_FAST)


# ---------------------------------------------------------------------------------------------------------------------------
# how far the contracted order is from the reference's: <= 1 ULP, differing samples counted
# ---------------------------------------------------------------------------------------------------------------------------
def test_contracted_eq_three_on_the_reference_golden_pair_within_one_ulp():
    """The reference's own test vector (eq_three.rs:150-167), through MX_FLAG_FP_CONTRACT: <= 1 ULP required, 0 differing samples
    measured (the f32 store absorbs the last-bit f64 differences on this signal; SURVEY 8c found the same on the CPU)."""
    x, want = _golden()
    for T in (None, 735):
        ws = Workspace(44100, 60)
        s = ws.source_mono(); e = ws.eq_three(4.0, 0.0, 4.0); ws.connect(s, 0, e, 0)
        n_ticks = x.size // 735
        g = ws.build(max_ticks_per_run=n_ticks if T is None else 1, flags=abi.FLAG_FP_CONTRACT)
        if T is None:
            g.write_source(s, x[:n_ticks * 735], n_ticks); g.run_ticks(0, n_ticks)
            got = g.read_output(e, 0, n_ticks, False)
        else:
            out = []
            for t in range(64):
                g.write_source(s, x[t * 735:(t + 1) * 735], 1); g.run_ticks(t, 1); out.append(g.read_output(e, 0, 1, False))
            got = np.concatenate(out)
        d = synth.ulp_diff(got, want[:got.size])
        assert d.max() <= 1, f"{d.max()} ULP from the reference's golden output"
        assert np.count_nonzero(d) <= got.size // 100000, f"{np.count_nonzero(d)} of {got.size} samples differ from the golden output"


def test_contracted_config2_at_2048_ticks_of_44k1_bit_exact_vs_the_contract_oracle(contracted):
    """The benchmark shape at the reference's own rate in the contracted order: the RT instantiations of the tiled kernel (ticks of 735 samples)."""
    import test_gpu_full_size as fs
    fs.test_config2_at_the_benchmarked_batch_length_2048_ticks_bit_exact((44100, 735))


def test_contracted_config2_full_size_every_strip_within_one_ulp_of_the_exact_order():
    """BASELINE configs[1] at its own size and rate (1024 strips, 48 kHz), module by module so that every strip is observable: each
    strip's Amplifier output within 1 ULP of the EXACT oracle's, differing samples counted; the buses equal the ordered sum of the
    device's own strips (the Mixer has nothing to contract)."""
    SR, SPT, N, T, runs = 48000, 800, 1024, 6, 2
    noise = [synth.noise(k, runs * T * SPT) for k in range(N)]
    ws, mix, srcs, trigs, got_m, got_c, got_s = _run_config2(N, SR, SPT, T, runs, abi.FLAG_FP_CONTRACT | abi.FLAG_NO_FUSE, T, noise)
    _wm, _wc, want_s = _oracle_config2(ws, mix, srcs, trigs, N, SPT, T, runs, noise, True)        # the oracle in its EXACT mode
    d = synth.ulp_diff(got_s.ravel(), want_s.ravel())
    assert d.max() <= 1, f"a strip sample is {d.max()} ULP from the exact order"
    n_diff = int(np.count_nonzero(d))
    assert n_diff <= d.size // 100000, f"{n_diff} of {d.size} strip samples differ from the exact order"
    # and the fused graph gives the very same buses (bit for bit): what the bench's fp_contract leg runs
    _ws, _mix, _s, _t, fm, fc_, _ = _run_config2(N, SR, SPT, T, runs, abi.FLAG_FP_CONTRACT, T, noise)
    assert_bit_exact(fm, got_m, "Master: fused vs module by module, contracted"); assert_bit_exact(fc_, got_c, "Cue")
    chans = [(float(synth.uniform(11, N, -24.0, 6.0)[k]), float(synth.uniform(12, N, 0.0, 1.0)[k]), k % 8 == 0) for k in range(N)]
    for run in range(runs):
        sl = slice(run * 2 * T * SPT, (run + 1) * 2 * T * SPT)
        want_m, want_c = oracle.mixer_run(chans, [got_s[k][sl] for k in range(N)], 2 * T * SPT)
        assert_bit_exact(got_m[sl], want_m, "Master = ordered sum of the device's strips"); assert_bit_exact(got_c[sl], want_c, "Cue")


def _config3(n_ch, T, flags):
    SPT = 735
    table = polyphase_table()
    ws = Workspace(44100, 60)
    srcs, outs = [], []
    for k in range(n_ch):
        s = ws.source_stereo(); f = ws.fir(reverb_taps(128, seed=20 + k)); r = ws.resample(160, 147, table)
        ws.connect(s, 0, f, 0); ws.connect(f, 0, r, 0)
        srcs.append(s); outs.append((f, r))
    g = ws.build(max_ticks_per_run=T, flags=flags)
    return ws, srcs, outs, g


def test_contracted_fir_and_resampler_equal_the_oracles_contract_mode_and_stay_within_one_ulp_of_the_spec():
    """config 3's chain (128-tap FIR -> 160/147 polyphase resampler), two submissions so that the histories carry: the contracted
    accumulation acc = fma(h[k], x, acc) equals the oracle's contract mode bit for bit (the whole chain), and each module is within
    1 ULP of the build-specified separate multiply-and-add order on the SAME input (the resampler is fed the device's own FIR output;
    the f32 store absorbs the last-bit f64 differences of the two orders: the oracle's two modes differ on 0 of 120 000 samples)."""
    n_ch, T, SPT = 12, 5, 735
    table = polyphase_table()
    ws, srcs, outs, g = _config3(n_ch, T, abi.FLAG_FP_CONTRACT)
    og_fc = oracle.OracleGraph(ws)
    P = table.shape[1]
    fir_hist = [np.zeros(127 * 2, np.float32) for _ in range(n_ch)]
    rs_hist = [np.zeros((P - 1) * 2, np.float32) for _ in range(n_ch)]
    n_diff = n_tot = 0
    for run in range(2):
        noise = [synth.noise(500 + 10 * run + k, 2 * SPT * T) for k in range(n_ch)]
        for k, s in enumerate(srcs):
            g.write_source(s, noise[k], T)
        g.run_ticks(run * T, T)
        got_f = [g.read_output(f, 0, T, True) for (f, _r) in outs]
        got_r = [g.read_output(r, 0, T, True, rate=(160, 147)) for (_f, r) in outs]
        with oracle.fp_contract():
            for t in range(T):
                for k, s in enumerate(srcs):
                    og_fc.set_source(s, noise[k][t * 2 * SPT:(t + 1) * 2 * SPT])
                og_fc.run_tick(run * T + t)
                for k, (f, r) in enumerate(outs):
                    assert_bit_exact(got_f[k][t * 2 * SPT:(t + 1) * 2 * SPT], og_fc.output(f, 0), f"FIR ch {k} tick {t}")
                    assert_bit_exact(got_r[k][t * 1600:(t + 1) * 1600], og_fc.output(r, 0), f"resampler ch {k} tick {t}")
        # the exact (separate multiply-and-add) spec, module by module on the same inputs
        for k in range(n_ch):
            want_f = oracle.fir_run(reverb_taps(128, seed=20 + k), fir_hist[k], noise[k])
            want_r = oracle.resample_run(table, 160, 147, rs_hist[k], run * T * SPT, run * T * 800, got_f[k], T * 800)
            for got, want, what in ((got_f[k], want_f, "FIR"), (got_r[k], want_r, "resampler")):
                d = synth.ulp_diff(got, want)
                assert d.max() <= 1, f"{what} ch {k} run {run}: {d.max()} ULP from the separate multiply-and-add order"
                n_diff += int(np.count_nonzero(d)); n_tot += d.size
    assert n_diff <= n_tot // 10000, f"{n_diff} of {n_tot} samples differ"
