"""BASELINE.json configs[4] on ONE GPU: W virtual ranks in one process, each with the graph a rank
of the sharded job runs -- its contiguous shard of the config-2 strips into Mixer(strips / W), and its row band of the 8-layer 1080p
cascade -- and one mx_exchange per rank on the library's in-process LOOPBACK transport (device-to-device copies where RCCL's
collectives would be; the packing, the slot pipelining, the time slicing and the rank-ordered combine are the RCCL path's own code,
mixlab_amd/csrc/mx_exchange.cpp).  Every rank's combined Master / Cue must equal, bit for bit, the oracle's run of the hierarchical
graph  W x Mixer(strips / W) -> Mixer(W, unity)  that defines the sharded job; the stitched bands must be the unsharded oracle picture.
The single-rank RCCL run of the same entry points is tests/test_gpu_exchange.py."""
import numpy as np
import pytest

import oracle
import oracle_video as ov
import synth
from mixlab_amd import abi, shard, video
from mixlab_amd.exchange import BusExchange, LoopbackGroup
from mixlab_amd.workspace import Workspace
from test_gpu_audio_parity import bits

pytestmark = pytest.mark.gpu

SR, SPT = 48000, 800
PER_RANK = 128
T, STEPS = 16, 3
FADERS = [1.0, 0.75, 0.5, 0.5, 0.25, 0.9, 0.1]
MATRIX = [3900, 150, 46, 4096, 60, 3980, 56, -2048, 20, 120, 3956, 0]


def gate_open(tick, k):
    return ((tick + k) // 30) % 2 == 1          # toggles every 30 ticks, phase k mod 60


def add_strips(ws, lo, n, total):
    """config-2 strips [lo, lo + n) of a `total`-strip job into a Mixer(n) of `ws` -> (mixer, sources, triggers)"""
    eq_g = synth.uniform(10, 3 * total, -24.0, 6.0)
    mg, mf = synth.uniform(11, total, -24.0, 6.0), synth.uniform(12, total, 0.0, 1.0)
    mix = ws.mixer([(float(mg[k]), float(mf[k]), k % 8 == 0) for k in range(lo, lo + n)])
    srcs, trigs = [], []
    for j, k in enumerate(range(lo, lo + n)):
        trig = ws.trigger(gate_open(0, k)); env = ws.envelope(); src = ws.source_mono()
        eq = ws.eq_three(float(eq_g[3 * k]), float(eq_g[3 * k + 1]), float(eq_g[3 * k + 2]))
        pan = ws.stereo_panner(); amp = ws.amplifier(1.0, 0.5)
        ws.connect(trig, 0, env, 0); ws.connect(src, 0, eq, 0)
        ws.connect(eq, 0, pan, 0); ws.connect(eq, 0, pan, 1)
        ws.connect(pan, 0, amp, 0); ws.connect(env, 0, amp, 1); ws.connect(amp, 0, mix, j)
        srcs.append(src); trigs.append(trig)
    return mix, srcs, trigs


_ORACLE = {}


def hierarchical_oracle(world):
    """Master / Cue of  world x Mixer(128) -> Mixer(world, unity)  over STEPS * T ticks, gates toggling between ticks."""
    if world in _ORACLE:
        return _ORACLE[world]
    total = world * PER_RANK
    ws = Workspace(SR, 60)
    fm, fc = ws.mixer(shard.combine_channels(world)), ws.mixer(shard.combine_channels(world))
    srcs, trigs = [], []
    for r in range(world):
        lo, n = shard.strip_range(r, world, total)
        sub, s, t = add_strips(ws, lo, n, total)
        ws.connect(sub, 0, fm, r); ws.connect(sub, 1, fc, r)
        srcs += s; trigs += t
    og = oracle.OracleGraph(ws)
    noise = [synth.noise(k, STEPS * T * SPT) for k in range(total)]
    m, c = [], []
    p_open, p_closed = abi.TriggerParams(1), abi.TriggerParams(0)
    for tick in range(STEPS * T):
        for k in range(total):
            if tick and (tick + k) % 30 == 0:
                og.update_params(trigs[k], p_open if gate_open(tick, k) else p_closed)
            og.set_source(srcs[k], noise[k][tick * SPT:(tick + 1) * SPT])
        og.run_tick(tick)
        m.append(og.output(fm, 0)); c.append(og.output(fc, 0))
    _ORACLE[world] = (np.concatenate(m), np.concatenate(c), noise)
    return _ORACLE[world]


def schedule_gates(g, trigs, lo, t0):
    p_open, p_closed = abi.TriggerParams(1), abi.TriggerParams(0)
    for j, tr in enumerate(trigs):
        k = lo + j
        for i in range(T):
            tick = t0 + i
            if tick and (tick + k) % 30 == 0:
                g.schedule_params(tr, i, p_open if gate_open(tick, k) else p_closed)


@pytest.mark.parametrize("mode", ["allgather", "slices"])
@pytest.mark.parametrize("world", [2, 8])
def test_virtual_ranks_through_the_exchange_equal_the_hierarchical_oracle_graph(world, mode):
    want_m, want_c, noise = hierarchical_oracle(world)
    total = world * PER_RANK
    grp = LoopbackGroup(world)
    ranks = []
    for r in range(world):
        lo, n = shard.strip_range(r, world, total)
        ws = Workspace(SR, 60)
        mix, srcs, trigs = add_strips(ws, lo, n, total)
        g = ws.build(max_ticks_per_run=T)
        ranks.append((g, mix, srcs, trigs, lo, BusExchange(g, mix, T, r, world, mode=mode, loopback=grp)))
    assert all(x[5].mode == mode and x[5].world == world for x in ranks)
    bus = 2 * 2 * SPT * T * 4
    assert ranks[0][5].bytes_received_per_step() == ((world - 1) * bus if mode == "allgather" else 2 * (world - 1) * bus // world)

    def run_step(i):
        for (g, mix, srcs, trigs, lo, ex) in ranks:
            for j, s in enumerate(srcs):
                g.write_source(s, noise[lo + j][i * T * SPT:(i + 1) * T * SPT], T)
            schedule_gates(g, trigs, lo, i * T)
            g.run_ticks(i * T, T)
            ex.submit(i)

    def check(i):
        sl = slice(i * T * 2 * SPT, (i + 1) * T * 2 * SPT)
        for r, x in enumerate(ranks):
            m, c = x[5].result(i)
            assert np.array_equal(bits(m), bits(want_m[sl])), f"rank {r}: Master of step {i}"
            assert np.array_equal(bits(c), bits(want_c[sl])), f"rank {r}: Cue of step {i}"

    run_step(0); run_step(1)          # two steps in flight (two slots)
    check(0)
    run_step(2)                       # reuses slot 0
    check(1); check(2)
    assert all(x[5].elapsed_ms(2) > 0 for x in ranks)
    # a rank's partial bus alone is NOT the whole bus (the exchange did something)
    assert not np.array_equal(bits(ranks[0][0].read_output(ranks[0][1], 0, T, True)), bits(want_m[2 * T * 2 * SPT:]))
    for x in ranks:
        x[5].close()
    grp.close()


def test_loopback_contract_errors():
    ws = Workspace(SR, 60)
    mix, _s, _t = add_strips(ws, 0, 4, 8)
    g = ws.build(max_ticks_per_run=6)
    grp = LoopbackGroup(2)
    with pytest.raises(abi.MxError):
        BusExchange(g, mix, 6, 0, 2, mode="allreduce", loopback=grp)       # RCCL's own order: not on the loopback transport
    with pytest.raises(abi.MxError):
        BusExchange(g, mix, 5, 0, 2, mode="slices", loopback=grp)          # 5 ticks do not cut into 2 time slices
    with pytest.raises(abi.MxError):
        BusExchange(g, mix, 7, 0, 2, mode="allgather", loopback=grp)       # more ticks than max_ticks_per_run
    with pytest.raises(abi.MxError):
        BusExchange(g, mix + 1, 6, 0, 2, mode="allgather", loopback=grp)   # not a Mixer
    with pytest.raises(abi.MxError):
        BusExchange(g, mix, 6, 0, 2, mode="allgather")                     # neither an id nor a group
    a = BusExchange(g, mix, 6, 0, 2, mode="allgather", loopback=grp)
    with pytest.raises(abi.MxError):
        BusExchange(g, mix, 6, 0, 2, mode="allgather", loopback=grp)       # rank 0 is taken
    g.run_ticks(0, 6)
    with pytest.raises(abi.MxError):
        a.submit(0)                                                        # rank 1 does not exist yet
    g2 = ws.build(max_ticks_per_run=6)
    b = BusExchange(g2, mix, 6, 1, 2, mode="allgather", loopback=grp)
    a.submit(0)
    with pytest.raises(abi.MxError):
        a.result(0)                                                        # rank 1 has not submitted step 0
    with pytest.raises(abi.MxError):
        a.submit(1)                                                        # ... and rank 0 may not run ahead
    g2.run_ticks(0, 6); b.submit(0)
    m0, _ = a.result(0); m1, _ = b.result(0)
    assert np.array_equal(bits(m0), bits(m1))
    with pytest.raises(abi.MxError):
        a.result(7)                                                        # never submitted
    a.close(); b.close(); grp.close()


def test_release_orders_slot_reuse_after_a_consumer_on_its_own_stream():
    """the result buffers of step i are rewritten by step i + 2; a consumer reading them on a stream of its own marks the end
    of its reads with mx_exchange_release and the exchange waits for that mark.  The consumer here is a second graph whose sources are
    BOUND to the combined bus (zero-copy) and which runs on its own stream long after submit(i + 2) was queued."""
    world = 2
    total = world * 8
    grp = LoopbackGroup(world)
    noise = [synth.noise(k, 4 * T * SPT) for k in range(total)]
    ranks = []
    for r in range(world):
        lo, n = shard.strip_range(r, world, total)
        ws = Workspace(SR, 60)
        mix, srcs, trigs = add_strips(ws, lo, n, total)
        g = ws.build(max_ticks_per_run=T)
        ranks.append((g, mix, srcs, lo, BusExchange(g, mix, T, r, world, mode="slices", loopback=grp)))

    def run_step(i):
        for (g, mix, srcs, lo, ex) in ranks:
            for j, s in enumerate(srcs):
                g.write_source(s, noise[lo + j][i * T * SPT:(i + 1) * T * SPT], T)
            g.run_ticks(i * T, T); ex.submit(i)

    run_step(0)
    ex0 = ranks[0][4]
    want0 = ex0.result(0)[0].copy()
    m_ptr, _c, n = ex0.device_result(0)
    cws = Workspace(SR, 60)
    s = cws.source_stereo(); amp = cws.amplifier(1.0, 0.0)
    cws.connect(s, 0, amp, 0)
    consumer = cws.build(max_ticks_per_run=T)
    consumer.bind_source_device(s, m_ptr)
    cs = consumer.stream()
    ex0.wait(0, cs)                   # the consumer's own stream waits for the combined bus of step 0 ...
    consumer.run_ticks(0, T)
    ex0.release(0, cs)                # ... and marks where its reads end
    run_step(1); run_step(2)          # step 2 rewrites the buffers the consumer read
    got = consumer.read_output(amp, 0, T, True)
    assert np.array_equal(bits(got), bits(want0))
    for x in ranks:
        x[4].close()
    consumer.close(); grp.close()


def test_rank_graphs_with_their_row_band_of_the_1080p_cascade_in_the_same_job():
    """configs[4] whole: every virtual rank's graph holds its 128 strips AND its row band of the 8-layer cascade (6 x 1080p + 2 x 720p,
    the 720p layers as halo slices scaled to the band by their source nodes); one submission of T ticks per step.  Audio through the
    exchange == the hierarchical oracle graph; the stitched RGBA bands of the last tick == the unsharded oracle picture."""
    from test_cpu_video_bands import rows_of, cascade as oracle_cascade
    world = 8
    want_m, want_c, noise = hierarchical_oracle(world)
    total = world * PER_RANK
    W, H, small = 1920, 1080, (1280, 720)
    sets = [[ov.HostFrame(W, H).fill(k, seed=4 + r) for k in range(6)] + [ov.HostFrame(*small).fill(k, seed=4 + r) for k in (6, 7)] for r in range(2)]
    last = sets[(T - 1) % 2]
    whole = []
    for f in last:
        o = f
        if (f.w, f.h) != (W, H):
            o = ov.HostFrame(W, H); ov.dynamic_scale(f, o)
        whole.append(o)
    want_rgba = ov.to_rgba(oracle_cascade(whole), MATRIX)
    got_rgba = np.zeros_like(want_rgba)
    grp = LoopbackGroup(world)
    ranks, keep = [], []
    for r, (row0, rows) in enumerate(shard.row_bands(H, world)):
        lo, n = shard.strip_range(r, world, total)
        ws = Workspace(SR, 60)
        mix, srcs, trigs = add_strips(ws, lo, n, total)
        vsrcs = [ws.source_video() for _ in range(8)]
        prev = vsrcs[0]
        for k in range(1, 8):
            m = ws.video_mixer(a=0, b=1, fader=FADERS[k - 1])
            ws.connect(prev, 0, m, 0); ws.connect(vsrcs[k], 0, m, 1)
            prev = m
        rgba = ws.video_to_rgba(MATRIX)
        ws.connect(prev, 0, rgba, 0)
        g = ws.build(max_ticks_per_run=T)
        for k in range(8):
            ring = []
            f0 = sets[0][k]
            scaled = (f0.w, f0.h) != (W, H)
            need = shard.band_source_rows((row0, rows), f0.w, f0.h, W, H) if scaled else (row0, rows)
            for q in range(2):
                hf = rows_of(sets[q][k], need[0], need[1])
                y, u, v = hf.visible()
                ring.append(video.DFrame(hf.w, hf.h).upload(y, u, v))
            if scaled:
                video.graph_set_video_source_band(g, vsrcs[k], f0.w, f0.h, need[0], need[1], W, H, row0, rows)
            keep.append(ring)
            video.graph_set_video_source_ring(g, vsrcs[k], ring, dur=(1, 60), off=(0, 1))
        ranks.append((g, mix, srcs, trigs, lo, BusExchange(g, mix, T, r, world, mode="auto", loopback=grp), rgba, row0, rows))
    assert ranks[0][5].mode == "slices"
    for (g, mix, srcs, trigs, lo, ex, rgba, row0, rows) in ranks:
        for j, s in enumerate(srcs):
            g.write_source(s, noise[lo + j][: T * SPT], T)
        schedule_gates(g, trigs, lo, 0)
        g.run_ticks(0, T)
        ex.submit(0)
    for r, (g, mix, srcs, trigs, lo, ex, rgba, row0, rows) in enumerate(ranks):
        m, c = ex.result(0)
        assert np.array_equal(bits(m), bits(want_m[: T * 2 * SPT])), f"rank {r}: Master"
        assert np.array_equal(bits(c), bits(want_c[: T * 2 * SPT])), f"rank {r}: Cue"
        got_rgba[row0:row0 + rows] = video.graph_rgba_output(g, rgba)
    assert np.array_equal(got_rgba, want_rgba)
    for x in ranks:
        x[5].close()
    grp.close()
