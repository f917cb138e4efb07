"""BASELINE.json configs[2]: FIR reverb + polyphase 44.1 -> 48 kHz resampler -- BUILD-SPECIFIED modules (the
reference has no counterpart: `TODO implement resampling`, src/icecast/mod.rs:94-97), parity unpinned;
bit-exact against this repo's own oracle."""
import numpy as np
import pytest

import oracle
import synth
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace
from test_gpu_audio_parity import SPT, assert_bit_exact, bits

pytestmark = pytest.mark.gpu


def reverb_taps(n=128, seed=7):
    """seeded exponentially decaying noise."""
    return (synth.uniform(seed, n, -1.0, 1.0) * np.exp(-np.arange(n) / 24.0) * 0.35).astype(np.float64)


def polyphase_table(up=160, down=147, taps_per_phase=16, beta=8.6):
    """Kaiser-windowed sinc prototype at the up-sampled rate, cut-off at the lower Nyquist, gain `up`;
    table[phase][k] = h[phase + k * up]."""
    n = up * taps_per_phase
    m = np.arange(n) - (n - 1) / 2.0
    fc = 0.5 / max(up, down) * 0.92
    h = 2 * fc * np.sinc(2 * fc * m) * np.kaiser(n, beta) * up
    return np.ascontiguousarray(h.reshape(taps_per_phase, up).T)


@pytest.mark.parametrize("n_taps", [1, 2, 3, 5, 128, 131, 300])
def test_fir_module_path_bit_exact_with_history(n_taps):
    taps = reverb_taps(n_taps)
    import struct
    blob = struct.pack("<II", n_taps, 0) + taps.tobytes()
    m = abi.Module(abi.KIND_FIR, blob)
    x = synth.noise(3, 2 * SPT * 6)
    hist = np.zeros(2 * max(1, n_taps - 1), np.float32)
    got = np.empty_like(x)
    want = np.empty_like(x)
    for k in range(6):                      # history must carry across calls (also when n_taps - 1 > frames is false/true)
        sl = slice(k * 2 * SPT, (k + 1) * 2 * SPT)
        want[sl] = oracle.fir_run(taps, hist, x[sl])
        m.run_tick(k * SPT, [(abi.MX_STEREO, x[sl])], [(abi.MX_STEREO, got[sl])])
    assert_bit_exact(got, want, f"FIR {n_taps} taps")


def test_fir_short_calls_shorter_than_the_filter():
    taps = reverb_taps(128)
    import struct
    m = abi.Module(abi.KIND_FIR, struct.pack("<II", 128, 0) + taps.tobytes())
    x = synth.noise(4, 2 * 50 * 9)
    hist = np.zeros(2 * 127, np.float32)
    for k in range(9):                      # 50-frame calls: history spans several calls
        sl = slice(k * 100, (k + 1) * 100)
        want = oracle.fir_run(taps, hist, x[sl])
        got = np.empty(100, np.float32)
        m.run_tick(k * 50, [(abi.MX_STEREO, x[sl])], [(abi.MX_STEREO, got)])
        assert_bit_exact(got, want, f"call {k}")


@pytest.mark.parametrize("batch", [1, 4])
def test_config3_fir_reverb_then_resampler_graph(batch):
    n_ch, n_ticks = 12, 8
    table = polyphase_table()
    ws = Workspace(44100, 60)
    srcs, outs = [], []
    for k in range(n_ch):
        s = ws.source_stereo(); f = ws.fir(reverb_taps(128, seed=20 + k)); r = ws.resample(160, 147, table)
        ws.connect(s, 0, f, 0); ws.connect(f, 0, r, 0)
        srcs.append(s); outs.append((f, r))
    mix = ws.mixer([(0.0, 1.0, k % 2 == 0) for k in range(n_ch)])       # a 48 kHz-domain mixer over the resampled channels
    for k, (_f, r) in enumerate(outs):
        ws.connect(r, 0, mix, k)
    g = ws.build(max_ticks_per_run=batch)
    og = oracle.OracleGraph(ws)
    noise = [synth.noise(60 + k, 2 * SPT * n_ticks) for k in range(n_ch)]
    for t0 in range(0, n_ticks, batch):
        for k, s in enumerate(srcs):
            g.write_source(s, noise[k][t0 * 2 * SPT:(t0 + batch) * 2 * SPT], batch)
        g.run_ticks(t0, batch)
        got_mix = g.read_output(mix, 0, batch, True, rate=(160, 147))
        got_r0 = g.read_output(outs[0][1], 0, batch, True, rate=(160, 147))
        got_f0 = g.read_output(outs[0][0], 0, batch, True)
        assert got_r0.size == batch * 2 * 800
        for kk in range(batch):
            for k, s in enumerate(srcs):
                og.set_source(s, noise[k][(t0 + kk) * 2 * SPT:(t0 + kk + 1) * 2 * SPT])
            og.run_tick(t0 + kk)
            assert_bit_exact(got_f0[kk * 2 * SPT:(kk + 1) * 2 * SPT], og.output(outs[0][0], 0), "FIR out")
            assert_bit_exact(got_r0[kk * 1600:(kk + 1) * 1600], og.output(outs[0][1], 0), "resampler out")
            assert_bit_exact(got_mix[kk * 1600:(kk + 1) * 1600], og.output(mix, 0), "48 kHz-domain mix")


def test_resampler_passes_a_sine_and_rejects_rate_dependent_modules_downstream():
    table = polyphase_table()
    ws = Workspace(44100, 60)
    s = ws.source_stereo(); r = ws.resample(160, 147, table)
    ws.connect(s, 0, r, 0)
    g = ws.build(max_ticks_per_run=10)
    n = np.arange(10 * SPT)
    x = np.sin(2 * np.pi * 1000.0 * n / 44100.0).astype(np.float32)
    g.write_source(s, np.repeat(x, 2), 10)
    g.run_ticks(0, 10)
    y = g.read_output(r, 0, 10, True, rate=(160, 147))[0::2]
    m = np.arange(y.size)
    delay = (160 * 16 - 1) / 2.0 / 160.0 * (48000.0 / 44100.0)         # prototype group delay in output samples
    ref = np.sin(2 * np.pi * 1000.0 * (m - delay) / 48000.0)
    assert np.max(np.abs(y[400:] - ref[400:])) < 2e-3                    # a 1 kHz tone comes out as a 1 kHz tone at 48 kHz
    # a module whose arithmetic depends on the sample rate is refused behind a resampler
    ws2 = Workspace(44100, 60)
    s2 = ws2.source_stereo(); r2 = ws2.resample(160, 147, table); sp = ws2.stereo_splitter(); e = ws2.eq_three(0, 0, 0)
    ws2.connect(s2, 0, r2, 0); ws2.connect(r2, 0, sp, 0); ws2.connect(sp, 0, e, 0)
    with pytest.raises(abi.MxError):
        ws2.build()


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_resampler_and_fir_chains_match_oracle_graph(seed):
    """Seeded random ratios (up- and down-sampling, two resamplers in a row), tap counts, FIR lengths and batch sizes through the
    graph path, against the oracle's graph runner tick by tick -- bit-exact, history carried across submissions."""
    rng = np.random.default_rng(seed)
    ratios = [(160, 147), (2, 1), (1, 3), (4, 5), (3, 7), (8, 7), (1, 1), (147, 160)]   # 735 * up / down stays whole, also chained
    ws = Workspace(44100, 60)
    chains = []
    for _ in range(int(rng.integers(1, 4))):
        s = ws.source_stereo()
        node, rate = s, (1, 1)
        stages = []
        for _stage in range(int(rng.integers(1, 4))):
            if rng.random() < 0.4:
                k = int(rng.integers(1, 70))
                f = ws.fir(rng.uniform(-0.5, 0.5, k)); ws.connect(node, 0, f, 0); node = f
                stages.append(node)
            else:
                cand = [r for r in ratios if (735 * rate[0] * r[0]) % (rate[1] * r[1]) == 0]
                up, down = cand[int(rng.integers(0, len(cand)))]
                tpp = int(rng.integers(1, 24))
                r = ws.resample(up, down, rng.uniform(-1.0, 1.0, (up, tpp))); ws.connect(node, 0, r, 0); node = r
                rate = (rate[0] * up, rate[1] * down)
                stages.append(node)
        chains.append((s, stages, rate))
    T = int(rng.integers(1, 4))
    g = ws.build(max_ticks_per_run=T)
    og = oracle.OracleGraph(ws)
    data = {s: synth.noise(700 + 13 * seed + s, 2 * 735 * T * 3) for (s, _st, _r) in chains}
    for run in range(3):
        for (s, _st, _r) in chains:
            g.write_source(s, data[s][run * T * 1470:(run + 1) * T * 1470], T)
        g.run_ticks(run * T, T)
        want = {}
        for t in range(T):
            for (s, _st, _r) in chains:
                og.set_source(s, data[s][(run * T + t) * 1470:(run * T + t + 1) * 1470])
            og.run_tick(run * T + t)
            for (_s, st, _r) in chains:
                for n in st:
                    want.setdefault(n, []).append(og.output(n, 0))
        for (_s, st, _r) in chains:
            for n in st:
                w = np.concatenate(want[n])
                got = g.read_output(n, 0, T, True, rate=(w.size, 2 * 735 * T))
                assert_bit_exact(got, w, f"seed {seed} run {run} node {n}")


@pytest.mark.parametrize("flags", [0, abi.FLAG_FP_CONTRACT], ids=["exact", "contracted"])
def test_device_fir_and_resamplers_agree_with_scipy_within_one_ulp(flags):
    """The independent pin of the two build-specified modules (tests/golden/make_fir_resample_scipy.py: scipy.signal.lfilter / upfirdn in f64 on seeded
    input), through the graph path in one eight-tick submission: every f32 the device produces is within 1 ULP of scipy's (2^-40 absolute at a zero
    crossing) -- in the spec's order and in the contracted one."""
    import pathlib
    from test_cpu_oracle_and_abi import ulp_distance_f32
    z = np.load(pathlib.Path(__file__).parent / "golden" / "fir_resample_scipy.npz")
    x = np.ascontiguousarray(z["x"]).reshape(-1)
    T = z["x"].shape[0] // 735
    ws = Workspace(44100, 60)
    src = ws.source_stereo()
    fir = ws.fir(z["fir_taps"]); ws.connect(src, 0, fir, 0)
    rs = {}
    for name in "abc":
        up, down, _tpp = (int(v) for v in z[f"rs_{name}_ratio"])
        rs[name] = ws.resample(up, down, z[f"rs_{name}_table"]); ws.connect(src, 0, rs[name], 0)
    g = ws.build(max_ticks_per_run=T, flags=flags)
    g.write_source(src, x, T)
    g.run_ticks(0, T)
    def close(got, want, what):
        got, want = np.asarray(got, np.float32).reshape(-1), np.asarray(want, np.float32).reshape(-1)
        assert got.size == want.size, what
        d = ulp_distance_f32(got, want)
        bad = (d > 1) & (np.abs(got.astype(np.float64) - want.astype(np.float64)) > 2.0 ** -40)
        assert not bad.any(), f"{what}: {int(bad.sum())} samples beyond 1 ULP, first {int(np.flatnonzero(bad)[0])}"
        assert (d != 0).mean() < 5e-3, f"{what}: {(d != 0).mean():.4f} of the samples differ"
    close(g.read_output(fir, 0, T, True), z["y_fir"], "FIR vs scipy.signal.lfilter")
    for name in "abc":
        want = z[f"rs_{name}_y"]
        close(g.read_output(rs[name], 0, T, True, rate=(want.size, 2 * 735 * T)), want, f"resampler {name} vs scipy.signal.upfirdn")
    g.close()


def test_disconnected_input_of_a_mixer_behind_the_resampler_reads_silence_at_full_capacity():
    """A Mixer in the 48 kHz domain (behind a 160/147 Resample) with one Disconnected channel, run at n_ticks ==
    max_ticks_per_run: the zero buffer (src/engine/io.rs:8-9) must cover the UPSAMPLED length -- the disconnected channel
    has fader 1.0 and cue on, so anything but silence would show on both buses."""
    from mixlab_amd.workspace import Workspace
    T = 8
    table = polyphase_table()
    ws = Workspace(44100, 60)
    s = ws.source_stereo(); r = ws.resample(160, 147, table)
    ws.connect(s, 0, r, 0)
    # many ports after the zero region so that an overrun would read live signal
    fill = [ws.oscillator(100.0 + k, abi.WAVE_SAW) for k in range(4)]
    mix = ws.mixer([(0.0, 1.0, True), (0.0, 1.0, True)])
    ws.connect(r, 0, mix, 0)                      # channel 1 stays Disconnected
    g = ws.build(max_ticks_per_run=T)
    x = synth.noise(77, 2 * 735 * T)
    g.write_source(s, x, T)
    g.run_ticks(0, T)                             # full capacity
    hist = np.zeros((table.shape[1] - 1) * 2, np.float32)
    want_r = oracle.resample_run(table, 160, 147, hist, 0, 0, x, 800 * T)
    want_m, want_c = oracle.mixer_run([(0.0, 1.0, True), (0.0, 1.0, True)], [want_r, None], 2 * 800 * T)
    assert np.array_equal(bits(g.read_output(mix, 0, T, True, rate=(160, 147))), bits(want_m))
    assert np.array_equal(bits(g.read_output(mix, 1, T, True, rate=(160, 147))), bits(want_c))
    assert fill
