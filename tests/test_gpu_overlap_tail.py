"""MX_FLAG_OVERLAP_TAIL: the last Mixer bank of a batched run executes on a second stream beside the NEXT run's earlier groups; the
ports it reads are double-buffered and alternate per run.  Nothing observable may change: every run's buses are the oracle's bit for
bit, whether runs are queued back to back (really overlapping) or read back one by one, fused or not, and across a run that is cut
by a scheduled non-Trigger update (which falls back to one stream)."""
import ctypes as C

import numpy as np
import pytest

import oracle
import synth
from mixlab_amd import abi
from test_gpu_audio_parity import assert_bit_exact, strips
from test_gpu_schedule import gate_open, schedule_gates

pytestmark = pytest.mark.gpu

SR, SPT = 48000, 800


def oracle_runs(ws, mix, srcs, trigs, noise, n_runs, batch, eq_update=None):
    og = oracle.OracleGraph(ws)
    out = []
    for r in range(n_runs):
        m, c = [], []
        for kk in range(batch):
            tick = r * batch + kk
            if eq_update and eq_update[0] == tick:
                og.update_params(eq_update[1], eq_update[2])
            for k, tr in enumerate(trigs):
                og.update_params(tr, abi.TriggerParams(1 if gate_open(tick, k) else 0))
            for k, s in enumerate(srcs):
                og.set_source(s, noise[k][tick * SPT:(tick + 1) * SPT])
            og.run_tick(tick)
            m.append(og.output(mix, 0).copy()); c.append(og.output(mix, 1).copy())
        out.append((np.concatenate(m), np.concatenate(c)))
    return out


@pytest.mark.parametrize("flags", [0, abi.FLAG_NO_FUSE], ids=["fused", "unfused"])
@pytest.mark.parametrize("readback", ["every-run", "last-run-only"])
def test_overlapped_tail_runs_are_the_oracles(flags, readback):
    n_strips, batch, n_runs = 24, 16, 6
    ws, mix, srcs, trigs = strips(n_strips, SR)
    noise = [synth.noise(k, n_runs * batch * SPT) for k in range(n_strips)]
    want = oracle_runs(ws, mix, srcs, trigs, noise, n_runs, batch)
    g = ws.build(max_ticks_per_run=batch, flags=flags | abi.FLAG_OVERLAP_TAIL)
    assert g.tail_stream() is not None
    plain = ws.build(max_ticks_per_run=batch, flags=flags)
    assert plain.tail_stream() is None
    amp = mix + 6                                    # strip 0's Amplifier: a port the tail reads
    seen = set()
    for r in range(n_runs):
        schedule_gates(g, trigs, r * batch, batch)
        for k, s in enumerate(srcs):
            g.write_source(s, noise[k][r * batch * SPT:(r + 1) * batch * SPT], batch)
        g.run_ticks(r * batch, batch)
        if flags & abi.FLAG_NO_FUSE:                 # (fused, the port is stored one float per frame and has no public pointer)
            seen.add(g.output_device_ptr(amp, 0)[0])
        if readback == "every-run" or r == n_runs - 1:
            assert_bit_exact(g.read_output(mix, 0, batch, True), want[r][0], f"master of run {r}")
            assert_bit_exact(g.read_output(mix, 1, batch, True), want[r][1], f"cue of run {r}")
    assert len(seen) in (0, 2)                       # the strip ports alternate between two buffers


def test_a_run_cut_by_a_scheduled_update_between_overlapped_runs():
    n_strips, batch, n_runs = 12, 16, 5
    ws, mix, srcs, trigs = strips(n_strips, SR)
    eq0 = mix + 4                                    # strip 0 EqThree (test_gpu_audio_parity.strips layout)
    assert ws.nodes[eq0][0] == abi.KIND_EQ_THREE
    new_eq = abi.EqThreeParams(-6.0, 3.0, 1.5)
    cut_tick = 2 * batch + 5                         # inside run 2
    noise = [synth.noise(k, n_runs * batch * SPT) for k in range(n_strips)]
    want = oracle_runs(ws, mix, srcs, trigs, noise, n_runs, batch, eq_update=(cut_tick, eq0, new_eq))
    g = ws.build(max_ticks_per_run=batch, flags=abi.FLAG_OVERLAP_TAIL)
    for r in range(n_runs):
        schedule_gates(g, trigs, r * batch, batch)
        if r == 2:
            g.schedule_params(eq0, cut_tick - r * batch, new_eq)
        for k, s in enumerate(srcs):
            g.write_source(s, noise[k][r * batch * SPT:(r + 1) * batch * SPT], batch)
        g.run_ticks(r * batch, batch)
        if r in (1, 2, 4):
            assert_bit_exact(g.read_output(mix, 0, batch, True), want[r][0], f"master of run {r}")
            assert_bit_exact(g.read_output(mix, 1, batch, True), want[r][1], f"cue of run {r}")


def test_mode_is_refused_silently_where_it_does_not_apply():
    # a Mixer that reads a source directly, or no Mixer at the end: the flag is accepted and the graph runs on one stream
    from mixlab_amd.workspace import Workspace
    ws = Workspace(SR, 60)
    s = ws.source_stereo(); m = ws.mixer([(0.0, 1.0, False)])
    ws.connect(s, 0, m, 0)
    g = ws.build(max_ticks_per_run=4, flags=abi.FLAG_OVERLAP_TAIL)
    assert g.tail_stream() is None
    x = synth.noise(3, 4 * 2 * SPT)
    g.write_source(s, x, 4); g.run_ticks(0, 4)
    assert_bit_exact(g.read_output(m, 0, 4, True), x, "unity mixer")


def test_mode_is_automatic_for_short_submissions_of_many_strips_and_can_be_turned_off(monkeypatch):
    """Round 5: a graph with at least 64 EqThree strips and submissions of at least 16 ticks gets the Mixer bank on the second stream WITHOUT the flag (held back until the
    next run's EqThree launch has been placed: DESIGN.md 5.2) unless MX_OVERLAP_AUTO=0 or the second buffers would not be affordable; a tick or a few at a time and small
    graphs stay on one stream.  The results are the oracle's either way."""
    n_strips, batch, n_runs = 64, 16, 4
    ws, mix, srcs, trigs = strips(n_strips, SR)
    noise = [synth.noise(k, n_runs * batch * SPT) for k in range(n_strips)]
    want = oracle_runs(ws, mix, srcs, trigs, noise, n_runs, batch)
    g = ws.build(max_ticks_per_run=batch)
    assert g.tail_stream() is not None
    assert ws.build(max_ticks_per_run=2048).tail_stream() is not None        # long submissions too, since the bank's launch is held behind the next EqThree launch
    assert ws.build(max_ticks_per_run=4).tail_stream() is None               # a tick or a few at a time: the real-time regime is left alone
    monkeypatch.setenv("MX_OVERLAP_AUTO_MAX_GB", "0.001")
    assert ws.build(max_ticks_per_run=2048).tail_stream() is None            # second buffers over the budget: 64 ports x 1.6 M frames
    monkeypatch.delenv("MX_OVERLAP_AUTO_MAX_GB")
    monkeypatch.setenv("MX_OVERLAP_AUTO", "0")
    off = ws.build(max_ticks_per_run=batch)
    assert off.tail_stream() is None
    monkeypatch.delenv("MX_OVERLAP_AUTO")
    for gg in (g, off):
        for r in range(n_runs):
            schedule_gates(gg, trigs, r * batch, batch)
            for k, s in enumerate(srcs):
                gg.write_source(s, noise[k][r * batch * SPT:(r + 1) * batch * SPT], batch)
            gg.run_ticks(r * batch, batch)
            if r in (0, 2, 3):
                assert_bit_exact(gg.read_output(mix, 0, batch, True), want[r][0], f"master of run {r}")
                assert_bit_exact(gg.read_output(mix, 1, batch, True), want[r][1], f"cue of run {r}")


def test_a_bank_released_by_the_next_runs_eq_three_launch_is_the_oracles():
    """Round 5: the bank's launch of run k is held back and goes out behind run k + 1's speculative EqThree launch (k_tail_gate).  Every read-back joins the streams and would
    release a held launch itself, so THIS launch's result is read where only it can be seen: by a copy queued on the tail stream after run k + 1 was queued (the bank of
    run k is then on that stream, the bank of run k + 1 is still held).  mx_graph_tail_stream() releases what is held: the handle is taken before anything is."""
    n_strips, batch, n_runs = 64, 16, 5                      # 16 ticks: long enough for the speculative EqThree path whose last workgroup opens the gate
    ws, mix, srcs, trigs = strips(n_strips, SR)
    noise = [synth.noise(k, n_runs * batch * SPT) for k in range(n_strips)]
    want = oracle_runs(ws, mix, srcs, trigs, noise, n_runs, batch)
    g = ws.build(max_ticks_per_run=batch, flags=abi.FLAG_OVERLAP_TAIL)   # (the flag, not the automatism: taking the buses' device pointers would end that)
    tail = g.tail_stream()
    assert tail is not None
    hip = C.CDLL("libamdhip64.so")
    pm, nm = g.output_device_ptr(mix, 0)
    pc, _ = g.output_device_ptr(mix, 1)
    assert nm == 2 * SPT                                   # floats per tick; the buffer holds the run's ticks back to back
    nm *= batch
    got_m, got_c = np.empty(nm, np.float32), np.empty(nm, np.float32)
    ran0, _ = g.eq_spec_stats()
    for r in range(n_runs):
        schedule_gates(g, trigs, r * batch, batch)
        for k, s in enumerate(srcs):
            g.write_source(s, noise[k][r * batch * SPT:(r + 1) * batch * SPT], batch)
        g.run_ticks(r * batch, batch)                        # queues run r's EqThree launch and, behind its gate, the bank of run r - 1
        if r >= 1:
            for dst, src in ((got_m, pm), (got_c, pc)):
                assert hip.hipMemcpyAsync(dst.ctypes.data_as(C.c_void_p), C.c_void_p(src), C.c_size_t(nm * 4), 2, C.c_void_p(tail)) == 0   # 2 = hipMemcpyDeviceToHost
            assert hip.hipStreamSynchronize(C.c_void_p(tail)) == 0
            assert_bit_exact(got_m, want[r - 1][0], f"master of run {r - 1}, released by run {r}")
            assert_bit_exact(got_c, want[r - 1][1], f"cue of run {r - 1}, released by run {r}")
    assert_bit_exact(g.read_output(mix, 0, batch, True), want[-1][0], "master of the last run (released by the read-back)")
    ran1, _ = g.eq_spec_stats()
    assert ran1 > ran0                                       # the speculative kernel (the one that opens the gate) is what ran
    gated, at_once = g.debug_tail_releases()
    assert gated == n_runs - 1 and at_once == 1, (gated, at_once)   # every bank but the last went out behind the next run's gate


def test_per_group_times_with_a_held_back_bank():
    """mx_graph_profile_*: a bank that was held back is timed by its own pair of events on the tail stream; every kind that launched reports a positive time."""
    n_strips, batch = 64, 16
    ws, mix, srcs, trigs = strips(n_strips, SR)
    g = ws.build(max_ticks_per_run=batch)                    # automatic mode
    assert g.tail_stream() is not None
    for k, s in enumerate(srcs):
        g.write_source(s, synth.noise(k, batch * SPT), batch)
    g.run_ticks(0, batch)
    g.sync()
    g.profile_enable(True)
    for r in range(1, 5):
        g.run_ticks(r * batch, batch)
    by_kind, total, n = g.profile_collect()
    assert n == 4 and total > 0
    assert by_kind["eq_three"] > 0 and by_kind["mixer"] > 0
    assert by_kind["mixer"] < 50.0 and by_kind["eq_three"] < 50.0   # ms over four runs of 16 ticks: event pairs of one stream each, no garbage from unrecorded events


def test_a_gate_that_nobody_opens_times_out_and_the_results_stand():
    """The gate is an ordering hint, never a dependency: with the speculative EqThree launch in its direct form (MX_EQ_SPEC_DIRECT: a kernel that does not store the
    flag) and the gate armed all the same (MX_TAIL_GATE_TEST -- since round 6 the library arms a gate only for the tiled launch that opens it) every held-back bank
    goes behind a gate that only its bounded spin (300 us) opens.  Same buses, bit for bit.  In a process of its own."""
    import os
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    code = f"""
import sys
sys.path.insert(0, {str(root)!r}); sys.path.insert(0, {str(root / 'tests')!r})
import numpy as np
import synth
from mixlab_amd import abi
from test_gpu_audio_parity import assert_bit_exact, strips
from test_gpu_schedule import schedule_gates
from test_gpu_overlap_tail import oracle_runs, SR, SPT
n_strips, batch, n_runs = 64, 16, 4
ws, mix, srcs, trigs = strips(n_strips, SR)
noise = [synth.noise(k, n_runs * batch * SPT) for k in range(n_strips)]
want = oracle_runs(ws, mix, srcs, trigs, noise, n_runs, batch)
g = ws.build(max_ticks_per_run=batch, flags=abi.FLAG_OVERLAP_TAIL)
for r in range(n_runs):
    schedule_gates(g, trigs, r * batch, batch)
    for k, s in enumerate(srcs):
        g.write_source(s, noise[k][r * batch * SPT:(r + 1) * batch * SPT], batch)
    g.run_ticks(r * batch, batch)
assert_bit_exact(g.read_output(mix, 0, batch, True), want[-1][0], "master of the last run")
assert_bit_exact(g.read_output(mix, 1, batch, True), want[-1][1], "cue of the last run")
gated, at_once = g.debug_tail_releases()
assert gated == n_runs - 1, (gated, at_once)
ran, _ = g.eq_spec_stats()
assert ran > 0
print("ok gate-timeout")
"""
    env = dict(os.environ, MX_EQ_SPEC_DIRECT="1", MX_TAIL_GATE_TEST="1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "ok gate-timeout" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


def test_group_buses_the_tail_is_the_bank_and_the_master_above_it():
    """Hierarchical mix (64 strips -> four group Mixers of 16 -> a master Mixer of 4): the tail is BOTH Mixer groups, held back together and released in order behind the
    next run's EqThree launch.  Master, cue and the four group buses of every run against the oracle ticked, read back one run later on the tail stream where only
    gated banks can be seen (flag mode), and directly (automatic mode)."""
    import ctypes as C
    from mixlab_amd.workspace import Workspace
    n_groups, per, batch, n_runs = 4, 16, 16, 5
    n_strips = n_groups * per
    ws = Workspace(SR, 60)
    gains = synth.uniform(10, 3 * n_strips, -24.0, 6.0)
    master = ws.mixer([(-1.0 * j, 0.9, j % 2 == 0) for j in range(n_groups)])
    gms = [ws.mixer([(-0.5 * k, 1.0 - 0.02 * k, k % 5 == 0) for k in range(per)]) for _ in range(n_groups)]
    srcs, trigs = [], []
    for k in range(n_strips):
        trig = ws.trigger(False); env = ws.envelope(); src = ws.source_mono()
        eq = ws.eq_three(float(gains[3 * k]), float(gains[3 * k + 1]), float(gains[3 * k + 2])); pan = ws.stereo_panner(); amp = ws.amplifier(1.0, 0.5)
        ws.connect(trig, 0, env, 0); ws.connect(src, 0, eq, 0); ws.connect(eq, 0, pan, 0); ws.connect(eq, 0, pan, 1)
        ws.connect(pan, 0, amp, 0); ws.connect(env, 0, amp, 1); ws.connect(amp, 0, gms[k // per], k % per)
        srcs.append(src); trigs.append(trig)
    for j, gm in enumerate(gms):
        ws.connect(gm, 0, master, j)
    noise = [synth.noise(k, n_runs * batch * SPT) for k in range(n_strips)]
    og = oracle.OracleGraph(ws)
    want = []
    for r in range(n_runs):
        acc = {"m": [], "c": [], "g": [[] for _ in gms]}
        for kk in range(batch):
            tick = r * batch + kk
            for k, tr in enumerate(trigs):
                og.update_params(tr, abi.TriggerParams(1 if gate_open(tick, k) else 0))
            for k, s in enumerate(srcs):
                og.set_source(s, noise[k][tick * SPT:(tick + 1) * SPT])
            og.run_tick(tick)
            acc["m"].append(og.output(master, 0).copy()); acc["c"].append(og.output(master, 1).copy())
            for j, gm in enumerate(gms):
                acc["g"][j].append(og.output(gm, 0).copy())
        want.append((np.concatenate(acc["m"]), np.concatenate(acc["c"]), [np.concatenate(x) for x in acc["g"]]))
    hip = C.CDLL("libamdhip64.so")
    for flags in (abi.FLAG_OVERLAP_TAIL, 0):
        g = ws.build(max_ticks_per_run=batch, flags=flags)
        tail = g.tail_stream()
        assert tail is not None
        if flags:
            pm, fpt = g.output_device_ptr(master, 0)
            pg = [g.output_device_ptr(gm, 0)[0] for gm in gms]
            nfl = fpt * batch
            got = np.empty(nfl, np.float32)
        for r in range(n_runs):
            schedule_gates(g, trigs, r * batch, batch)
            for k, s in enumerate(srcs):
                g.write_source(s, noise[k][r * batch * SPT:(r + 1) * batch * SPT], batch)
            g.run_ticks(r * batch, batch)
            if flags and r >= 1:                      # the banks of run r - 1, released by run r, on the tail stream
                for ptr, w, what in [(pm, want[r - 1][0], "master")] + [(pg[j], want[r - 1][2][j], f"group bus {j}") for j in range(n_groups)]:
                    assert hip.hipMemcpyAsync(got.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(nfl * 4), 2, C.c_void_p(tail)) == 0
                    assert hip.hipStreamSynchronize(C.c_void_p(tail)) == 0
                    assert_bit_exact(got, w, f"{what} of run {r - 1}, released by run {r}")
            if not flags and r in (1, 3):
                assert_bit_exact(g.read_output(master, 0, batch, True), want[r][0], f"master of run {r}")
                assert_bit_exact(g.read_output(gms[2], 0, batch, True), want[r][2][2], f"group bus 2 of run {r}")
        assert_bit_exact(g.read_output(master, 0, batch, True), want[-1][0], "master of the last run")
        assert_bit_exact(g.read_output(master, 1, batch, True), want[-1][1], "cue of the last run")
        for j, gm in enumerate(gms):
            assert_bit_exact(g.read_output(gm, 0, batch, True), want[-1][2][j], f"group bus {j} of the last run")
        gated, at_once = g.debug_tail_releases()
        assert gated >= 2, (gated, at_once)


@pytest.mark.parametrize("runs_before", [1, 2], ids=["odd-parity", "even-parity"])
@pytest.mark.parametrize("what", ["bus", "strip-port"])
def test_taking_a_raw_pointer_ends_the_automatism_and_leaves_a_one_stream_graph(runs_before, what):
    """The automatic mode ends when a host takes mx_graph_output_device_ptr of a bus (the tail's output) or of a port the tail reads (double-buffered while the mode is on).
    From then on the graph is a one-stream graph in EVERY respect: after an odd number of overlapped runs (the ports at their second buffer) a parameter update, a run cut
    by a scheduled update and later runs still land in the descriptors the kernels read, the pointer names the buffer that holds the last run, and it stays fresh."""
    n_strips, batch, n_runs = 64, 16, runs_before + 3
    flags = abi.FLAG_NO_FUSE if what == "strip-port" else 0       # (fused, a strip port is stored one float per frame and has no public pointer)
    ws, mix, srcs, trigs = strips(n_strips, SR)
    eq0, amp0 = mix + 4, mix + 6
    assert ws.nodes[eq0][0] == abi.KIND_EQ_THREE and ws.nodes[amp0][0] == abi.KIND_AMPLIFIER
    new_eq = abi.EqThreeParams(-6.0, 3.0, 1.5)
    upd_tick = runs_before * batch                                # update_params between two runs, right after the pointer was taken
    cut_eq, cut_tick = abi.EqThreeParams(2.0, -4.0, 0.5), (runs_before + 1) * batch + 5     # and a run cut by a scheduled update after that
    noise = [synth.noise(k, n_runs * batch * SPT) for k in range(n_strips)]
    og = oracle.OracleGraph(ws)
    want, want_amp = [], []
    for r in range(n_runs):
        m, c, a = [], [], []
        for kk in range(batch):
            tick = r * batch + kk
            if tick == upd_tick:
                og.update_params(eq0, new_eq)
            if tick == cut_tick:
                og.update_params(eq0, cut_eq)
            for k, tr in enumerate(trigs):
                og.update_params(tr, abi.TriggerParams(1 if gate_open(tick, k) else 0))
            for k, s in enumerate(srcs):
                og.set_source(s, noise[k][tick * SPT:(tick + 1) * SPT])
            og.run_tick(tick)
            m.append(og.output(mix, 0).copy()); c.append(og.output(mix, 1).copy()); a.append(og.output(amp0, 0).copy())
        want.append((np.concatenate(m), np.concatenate(c))); want_amp.append(np.concatenate(a))
    g = ws.build(max_ticks_per_run=batch, flags=flags)
    assert g.tail_stream() is not None                            # automatic mode
    hip = C.CDLL("libamdhip64.so")
    ptr = n_fl = None

    def peek():                                                   # what a stream-ordered consumer on the graph's stream sees behind the raw pointer
        g.sync()
        buf = np.empty(n_fl, np.float32)
        assert hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(n_fl * 4), 2) == 0
        return buf

    for r in range(n_runs):
        schedule_gates(g, trigs, r * batch, batch)
        if r == runs_before + 1:
            g.schedule_params(eq0, cut_tick - r * batch, cut_eq)
        for k, s in enumerate(srcs):
            g.write_source(s, noise[k][r * batch * SPT:(r + 1) * batch * SPT], batch)
        g.run_ticks(r * batch, batch)
        if r == runs_before - 1:
            ptr, fpt = g.output_device_ptr(mix if what == "bus" else amp0, 0)
            n_fl = fpt * batch
            assert g.tail_stream() is None                        # the automatism ended, for good
            assert_bit_exact(peek(), want[r][0] if what == "bus" else want_amp[r], f"behind the pointer right after it was taken (run {r})")
            g.update_params(eq0, new_eq)
        elif ptr is not None:
            assert g.output_device_ptr(mix if what == "bus" else amp0, 0)[0] == ptr     # it does not move any more
            assert_bit_exact(peek(), want[r][0] if what == "bus" else want_amp[r], f"behind the pointer after run {r}")
        assert_bit_exact(g.read_output(mix, 0, batch, True), want[r][0], f"master of run {r}")
        assert_bit_exact(g.read_output(mix, 1, batch, True), want[r][1], f"cue of run {r}")


def test_a_direct_form_eq_three_launch_does_not_arm_a_gate(monkeypatch):
    """Only the tiled speculative kernel stores the flag k_tail_gate waits for.  Where the planner takes the direct form (MX_EQ_SPEC_DIRECT here; mixed-mode groups and short
    ragged ticks in the field) the held-back bank is released at once instead of spinning to the gate's 300 us limit -- and the results stand."""
    n_strips, batch, n_runs = 64, 16, 4
    ws, mix, srcs, trigs = strips(n_strips, SR)
    noise = [synth.noise(k, n_runs * batch * SPT) for k in range(n_strips)]
    want = oracle_runs(ws, mix, srcs, trigs, noise, n_runs, batch)
    monkeypatch.setenv("MX_EQ_SPEC_DIRECT", "1")
    g = ws.build(max_ticks_per_run=batch)
    assert g.tail_stream() is not None
    for r in range(n_runs):
        schedule_gates(g, trigs, r * batch, batch)
        for k, s in enumerate(srcs):
            g.write_source(s, noise[k][r * batch * SPT:(r + 1) * batch * SPT], batch)
        g.run_ticks(r * batch, batch)
    assert_bit_exact(g.read_output(mix, 0, batch, True), want[-1][0], "master of the last run")
    gated, at_once = g.debug_tail_releases()
    assert gated == 0 and at_once >= n_runs - 1, (gated, at_once)
