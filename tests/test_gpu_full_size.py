"""BASELINE.json configurations at their FULL sizes (1024 strips @48 kHz; 256 stereo FIR + resampler channels), checked
through properties that do not need the CPU oracle to replay the whole job:

* the Mixer is an ordered f32 sum: the oracle mixer fed with the DEVICE's own 1024 strip outputs must give the
  device's Master / Cue bit for bit (checksum of parts);
* batching is invisible: T ticks in one submission == T submissions of one tick, bit for bit (exact mode);
* fusion is invisible: every surviving port has the same bits with and without it;
* the time-parallel EqThree stays within 1 ULP of the exact-order kernel (itself bit-exact against the
  reference's golden pair at small sizes), with rare mismatches, on every one of the 1024 strips;
* time-splitting a stream across workgroups leaves the mix within a few ULP (span-initial states differ by ~1e-16).

The small-size tests next door compare the same kernels against the oracle sample by sample.
"""
import numpy as np
import pytest

import oracle
import synth
from mixlab_amd import abi
from test_gpu_audio_parity import strips, bits
from test_gpu_fir_resample import polyphase_table, reverb_taps

pytestmark = pytest.mark.gpu

SR48, SPT48 = 48000, 800
N = 1024


def _feed(g, srcs, noise, run, T, spt):
    for k, s in enumerate(srcs):
        g.write_source(s, noise[k][run * T * spt:(run + 1) * T * spt], T)


@pytest.fixture(scope="module")
def noise48():
    return [synth.noise(k, 16 * SPT48) for k in range(N)]


@pytest.mark.parametrize("coop_blocks", ["0", "100000"])   # the streaming kernel (what long runs use) and the cooperative one
def test_config2_full_size_mixer_is_the_ordered_sum_of_the_device_strips(noise48, coop_blocks, monkeypatch):
    monkeypatch.setenv("MX_MIXER_COOP_BLOCKS", coop_blocks)
    T = 4
    ws, mix, srcs, trigs = strips(N, SR48)
    g = ws.build(max_ticks_per_run=T)
    for k, tr in enumerate(trigs):
        g.update_params(tr, abi.TriggerParams(((k % 60) // 30) == 1))
    _feed(g, srcs, noise48, 0, T, SPT48)
    g.run_ticks(0, T)
    chans = [(float(synth.uniform(11, N, -24.0, 6.0)[k]), float(synth.uniform(12, N, 0.0, 1.0)[k]), k % 8 == 0) for k in range(N)]
    dev_amp = [g.read_output(mix + 6 * k + 6, 0, T, True) for k in range(N)]
    want_m, want_c = oracle.mixer_run(chans, dev_amp, 2 * T * SPT48)
    assert np.array_equal(bits(g.read_output(mix, 0, T, True)), bits(want_m))
    assert np.array_equal(bits(g.read_output(mix, 1, T, True)), bits(want_c))


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_config2_full_size_48khz_every_strip_and_the_mix_against_the_oracle(noise48, mode):
    """BASELINE configs[1] at its own size and rate -- 1024 strips, 48 kHz -- against the ORACLE (not against another device
    kernel): every strip's Amplifier output and the Master / Cue buses, sample by sample, over two runs of 6 ticks with
    the gates set per strip between the runs.  Exact mode: bit-exact.  Fast (time-parallel) mode: every strip within
    1 ULP, and the Mixer bit-exact given the device's own strips."""
    T, runs = 6, 2
    ws, mix, srcs, trigs = strips(N, SR48)
    og = oracle.OracleGraph(ws)
    g = ws.build(max_ticks_per_run=T, flags=abi.FLAG_EQ_EXACT if mode == "exact" else abi.FLAG_EQ_FAST | abi.FLAG_NO_FUSE)
    fused = mode == "exact"
    n_diff = 0
    for run in range(runs):
        for k, tr in enumerate(trigs):
            p = abi.TriggerParams(1 if ((run * T + k) // 3) % 2 else 0)
            g.update_params(tr, p); og.update_params(tr, p)
        _feed(g, srcs, noise48, run, T, SPT48)
        g.run_ticks(run * T, T)
        got_m, got_c = g.read_output(mix, 0, T, True), g.read_output(mix, 1, T, True)
        dev_amp = None if fused else [g.read_output(mix + 6 * k + 6, 0, T, True) for k in range(N)]
        for kk in range(T):
            tick = run * T + kk
            for k, s in enumerate(srcs):
                og.set_source(s, noise48[k][tick * SPT48:(tick + 1) * SPT48])
            og.run_tick(tick)
            sl = slice(kk * 2 * SPT48, (kk + 1) * 2 * SPT48)
            if fused:
                assert np.array_equal(bits(got_m[sl]), bits(og.output(mix, 0))), f"Master tick {tick}"
                assert np.array_equal(bits(got_c[sl]), bits(og.output(mix, 1))), f"Cue tick {tick}"
            else:
                for k in range(N):
                    d = synth.ulp_diff(dev_amp[k][sl], og.output(mix + 6 * k + 6, 0))
                    assert d.max() <= 1, f"strip {k} tick {tick}: {d.max()} ULP"
                    n_diff += int(np.count_nonzero(d))
        if not fused:
            chans = [(float(synth.uniform(11, N, -24.0, 6.0)[k]), float(synth.uniform(12, N, 0.0, 1.0)[k]), k % 8 == 0) for k in range(N)]
            want_m, want_c = oracle.mixer_run(chans, dev_amp, 2 * T * SPT48)
            assert np.array_equal(bits(got_m), bits(want_m)) and np.array_equal(bits(got_c), bits(want_c))
    assert n_diff <= runs * T * N * 2 * SPT48 // 20000


def test_config2_full_size_batching_and_fusion_are_invisible_in_exact_mode(noise48):
    T = 6
    outs = {}
    for name, flags, batch in (("batched", abi.FLAG_EQ_EXACT, T), ("ticked", abi.FLAG_EQ_EXACT, 1), ("unfused", abi.FLAG_EQ_EXACT | abi.FLAG_NO_FUSE, T)):
        ws, mix, srcs, trigs = strips(N, SR48)
        g = ws.build(max_ticks_per_run=batch, flags=flags)
        res_m, res_c = [], []
        for t0 in range(0, T, batch):
            for k, s in enumerate(srcs):
                g.write_source(s, noise48[k][t0 * SPT48:(t0 + batch) * SPT48], batch)
            g.run_ticks(t0, batch)
            res_m.append(g.read_output(mix, 0, batch, True)); res_c.append(g.read_output(mix, 1, batch, True))
        outs[name] = (np.concatenate(res_m), np.concatenate(res_c))
    for other in ("ticked", "unfused"):
        assert np.array_equal(bits(outs["batched"][0]), bits(outs[other][0])), f"Master differs: batched vs {other}"
        assert np.array_equal(bits(outs["batched"][1]), bits(outs[other][1])), f"Cue differs: batched vs {other}"


def test_config2_full_size_time_parallel_eq_within_one_ulp_of_exact_order_on_every_strip(noise48):
    T = 16
    res = {}
    for name, flags in (("exact", abi.FLAG_EQ_EXACT), ("scan", abi.FLAG_EQ_FAST)):
        ws, mix, srcs, trigs = strips(N, SR48)
        g = ws.build(max_ticks_per_run=T, flags=flags)
        for k, tr in enumerate(trigs):
            g.update_params(tr, abi.TriggerParams(((k % 60) // 30) == 1))
        _feed(g, srcs, noise48, 0, T, SPT48)
        g.run_ticks(0, T)
        res[name] = np.stack([g.read_output(mix + 6 * k + 6, 0, T, True)[0::2] for k in range(N)])
    d = synth.ulp_diff(res["scan"].ravel(), res["exact"].ravel())
    assert d.max() <= 1
    assert np.count_nonzero(d) <= d.size // 20000, f"{np.count_nonzero(d)} of {d.size} samples differ by 1 ULP"


def test_rank_sized_shard_time_split_mix_close_to_unsplit(noise48, monkeypatch):
    # what one rank of an 8-GPU job runs: 128 strips, streams cut into 8 spans
    T = 16
    outs = []
    for force in ("1", "0"):
        monkeypatch.setenv("MX_EQ_SPLIT", force)
        ws, mix, srcs, trigs = strips(128, SR48)
        g = ws.build(max_ticks_per_run=T, flags=abi.FLAG_EQ_FAST)
        res = []
        for run in range(1):
            _feed(g, srcs, noise48, run, T, SPT48)
            g.run_ticks(run * T, T)
            res.append(g.read_output(mix, 0, T, True))
        outs.append(np.concatenate(res))
    assert np.max(np.abs(outs[0] - outs[1])) <= 16 * np.spacing(np.float32(np.max(np.abs(outs[0]))))
    assert np.count_nonzero(outs[0] != outs[1]) <= outs[0].size // 500


def test_config3_full_size_256_channels_fir_then_resampler_spot_checked_against_oracle():
    n_ch, T, SPT = 256, 4, 735
    from mixlab_amd.workspace import Workspace
    table = polyphase_table()
    ws = Workspace(44100, 60)
    srcs, outs = [], []
    for k in range(n_ch):
        s = ws.source_stereo(); f = ws.fir(reverb_taps(128, seed=20 + k)); r = ws.resample(160, 147, table)
        ws.connect(s, 0, f, 0); ws.connect(f, 0, r, 0)
        srcs.append(s); outs.append((f, r))
    mix = ws.mixer([(0.0, 1.0, k % 2 == 0) for k in range(n_ch)])
    for k, (_f, r) in enumerate(outs):
        ws.connect(r, 0, mix, k)
    g = ws.build(max_ticks_per_run=T)
    noise = [synth.noise(60 + k, 2 * SPT * T) for k in range(n_ch)]
    for k, s in enumerate(srcs):
        g.write_source(s, noise[k], T)
    g.run_ticks(0, T)
    # every 17th channel against the per-module oracle; the mix against the oracle mixer over the device's own channels
    for k in range(0, n_ch, 17):
        taps = reverb_taps(128, seed=20 + k)
        hist = np.zeros((len(taps) - 1) * 2, np.float32)
        want_f = oracle.fir_run(taps, hist, noise[k])
        assert np.array_equal(bits(g.read_output(outs[k][0], 0, T, True)), bits(want_f)), f"FIR channel {k}"
    dev_r = [g.read_output(r, 0, T, True, rate=(160, 147)) for (_f, r) in outs]
    want_m, want_c = oracle.mixer_run([(0.0, 1.0, k % 2 == 0) for k in range(n_ch)], dev_r, 2 * T * 800)
    assert np.array_equal(bits(g.read_output(mix, 0, T, True, rate=(160, 147))), bits(want_m))
    assert np.array_equal(bits(g.read_output(mix, 1, T, True, rate=(160, 147))), bits(want_c))


@pytest.mark.parametrize("rate", [(48000, 800), (44100, 735)], ids=["48k", "44k1"])
def test_config2_at_the_benchmarked_batch_length_2048_ticks_bit_exact(rate):
    """bench.py's submission shape at full length: 2048 ticks in ONE run, gates toggling every 30 ticks inside it -- the speculative tiled
    EqThree kernel with the inline branch-free Envelope over tens of chunks per strip -- against the oracle ticked tick by tick with its
    gate updates between ticks.  16 strips keep the oracle at a second of CPU.  At 48 kHz ticks are whole super-blocks of the tile; at the
    reference's own 44.1 kHz they are not (735 samples): chunks of eight ticks whose rows are not line-aligned, a tick boundary inside
    one super-block in 23 (the RT instantiations of the kernel)."""
    from test_gpu_schedule import gate_open, schedule_gates
    (SR, SPT), T, n_strips = rate, 2048, 16
    ws, mix, srcs, trigs = strips(n_strips, SR)
    g = ws.build(max_ticks_per_run=T)
    noise = [synth.noise(100 + k, T * SPT) for k in range(n_strips)]
    n_ev = schedule_gates(g, trigs, 0, T)
    assert n_ev >= n_strips * (T // 30 - 2)
    for k, s in enumerate(srcs):
        g.write_source(s, noise[k], T)
    g.run_ticks(0, T)
    ran, repaired = g.eq_spec_stats()
    assert ran >= 16 * n_strips and repaired == 0          # the time-parallel kernel ran, and proved itself without repairs
    got_m, got_c = g.read_output(mix, 0, T, True), g.read_output(mix, 1, T, True)
    og = oracle.OracleGraph(ws)
    for tick in range(T):
        for k, tr in enumerate(trigs):
            if tick == 0 or gate_open(tick, k) != gate_open(tick - 1, k):
                og.update_params(tr, abi.TriggerParams(1 if gate_open(tick, k) else 0))
        for k, s in enumerate(srcs):
            og.set_source(s, noise[k][tick * SPT:(tick + 1) * SPT])
        og.run_tick(tick)
        sl = slice(tick * 2 * SPT, (tick + 1) * 2 * SPT)
        assert np.array_equal(got_m[sl].view(np.uint32), og.output(mix, 0).view(np.uint32)), f"master differs in tick {tick}"
        assert np.array_equal(got_c[sl].view(np.uint32), og.output(mix, 1).view(np.uint32)), f"cue differs in tick {tick}"


@pytest.mark.parametrize("mode", ["exact", "contract"])
def test_headline_shape_1024_strips_x_2048_ticks_with_the_planners_own_plan_against_the_oracle(mode):
    """bench.py's headline job AT ITS OWN SHAPE -- 1024 strips x 2048 ticks per submission @48 kHz, gates toggling inside the submissions through
    mx_graph_schedule_params_batch, no environment forcing: the kernel instantiation, chunk plan and launch geometry are the ones the timed region
    runs -- over two submissions (state carried from the first into the second).  16 sampled strips are replayed through the oracle from tick 0
    and compared bit for bit with the second submission's fused strip outputs; Master / Cue of sampled ticks against the oracle Mixer over the
    device's own 1024 strips (tests/headline_replay.py -- the same checker bench.py runs after its timed region)."""
    import bench
    import headline_replay as hr
    from mixlab_amd.workspace import Workspace

    T, n_steps = 2048, 2
    flags = abi.FLAG_FP_CONTRACT if mode == "contract" else 0
    ws, mix, srcs, trigs = bench.build_strips(abi, Workspace, synth, N, 0, SR48, want_trigs=True)
    g = ws.build(max_ticks_per_run=T, flags=flags)
    base_ticks = 256

    def src_of(j):
        return np.tile(synth.noise(j, base_ticks * SPT48), T // base_ticks)

    for j, s in enumerate(srcs):
        g.write_source(s, src_of(j), T)
    for i in range(n_steps):
        ev = bench.gate_events(abi, trigs, 0, i * T, T)
        g.schedule_params_batch(ev[0], ev[1])
        g.run_ticks(i * T, T)
    ran, repaired = g.eq_spec_stats()
    assert ran >= n_steps * 16 * N                  # the speculative time-parallel kernel ran (not the short-stream path)
    ids = hr.sample_strips(N, 16)
    mg, mf = synth.uniform(11, N, -24.0, 6.0), synth.uniform(12, N, 0.0, 1.0)

    def one(k):
        ws1, mix1, srcs1, trigs1 = bench.build_strips(abi, Workspace, synth, 1, k, SR48, total=N, want_trigs=True)
        return ws1, (mix1, srcs1[0], trigs1[0], mix1 + 6)

    rec = hr.replay_and_compare(g, one, ids, 0, {j: src_of(j) for j in ids}, T, n_steps, mix, lambda j: mix + 6 * j + 6, toggling=True,
                                contract=mode == "contract", check_ticks=6, all_amp_nodes=[mix + 6 * j + 6 for j in range(N)],
                                mixer_channels=[(float(mg[k]), float(mf[k]), k % 8 == 0) for k in range(N)])
    assert rec["verdict"] == "bit-exact", rec
    assert rec["strips_checked"] == 16 and rec["samples_compared"] == 16 * T * 2 * SPT48
    assert rec["buses"]["verdict"] == "bit-exact", rec


def test_headline_shape_bank_released_behind_the_gate_equals_the_one_stream_bank(monkeypatch):
    """Round 5: at the headline's own shape the Mixer bank of run k goes out behind the gate that run k + 1's EqThree launch opens and shares the chip with it.  Only the
    schedule differs from the one-stream form -- which the test above pins to the oracle -- so the buses must be the same bits: the bank of run 1, read by a copy queued
    on the tail stream once run 2 is queued (the only place a gated bank can be seen: every read-back releases a held bank itself), against the same run's buses of a
    graph built with MX_OVERLAP_AUTO=0."""
    import ctypes as C

    import bench
    from mixlab_amd.workspace import Workspace

    T, n_runs = 2048, 3
    ws, mix, srcs, trigs = bench.build_strips(abi, Workspace, synth, N, 0, SR48, want_trigs=True)
    g = ws.build(max_ticks_per_run=T, flags=abi.FLAG_OVERLAP_TAIL)
    tail = g.tail_stream()
    assert tail is not None
    monkeypatch.setenv("MX_OVERLAP_AUTO", "0")
    one = ws.build(max_ticks_per_run=T)
    monkeypatch.delenv("MX_OVERLAP_AUTO")
    assert one.tail_stream() is None
    for j, s in enumerate(srcs):
        buf = np.tile(synth.noise(j, 256 * SPT48), T // 256)
        g.write_source(s, buf, T)
        one.bind_source_device(s, g.output_device_ptr(s, 0)[0])
    hip = C.CDLL("libamdhip64.so")
    pm, fpt = g.output_device_ptr(mix, 0)
    pc, _ = g.output_device_ptr(mix, 1)
    n = fpt * T
    got_m, got_c = np.empty(n, np.float32), np.empty(n, np.float32)
    want = None
    for r in range(n_runs):
        ev = bench.gate_events(abi, trigs, 0, r * T, T)
        g.schedule_params_batch(ev[0], ev[1]); one.schedule_params_batch(ev[0], ev[1])
        g.run_ticks(r * T, T)
        if r == 2:      # run 2 is queued: the bank of run 1 is behind its gate on the tail stream, the bank of run 2 is held
            for dst, src in ((got_m, pm), (got_c, pc)):
                assert hip.hipMemcpyAsync(dst.ctypes.data_as(C.c_void_p), C.c_void_p(src), C.c_size_t(n * 4), 2, C.c_void_p(tail)) == 0
            assert hip.hipStreamSynchronize(C.c_void_p(tail)) == 0
        one.run_ticks(r * T, T)
        if r == 1:
            want = (one.read_output(mix, 0, T, True).copy(), one.read_output(mix, 1, T, True).copy())
    assert np.array_equal(got_m.view(np.uint32), want[0].view(np.uint32)), "master of run 1"
    assert np.array_equal(got_c.view(np.uint32), want[1].view(np.uint32)), "cue of run 1"
    gated, at_once = g.debug_tail_releases()
    assert gated == 2 and at_once == 0, (gated, at_once)
    assert np.array_equal(g.read_output(mix, 0, T, True).view(np.uint32), one.read_output(mix, 0, T, True).view(np.uint32))   # and the last run's, released by the read-back
