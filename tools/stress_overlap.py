"""Stress of the second-stream mode (the Mixer bank of run k held back for run k + 1's EqThree launch): config-2 strips, random run lengths (a tick or a few -- which stay on
one stream -- up to the graph's maximum), runs cut by a scheduled EqThree update, read-backs / mx_graph_sync / mx_graph_tail_stream at random points, Trigger updates between
runs, the automatic mode and MX_FLAG_OVERLAP_TAIL, exact and contracted order; every run's Master and Cue against the oracle ticked -- read back at once, or (flag mode) one
run later from a copy queued on the tail stream, which is where a bank released by the NEXT run's launch can be seen.   python tools/stress_overlap.py [first_seed] [count]"""
import ctypes as C, sys, pathlib, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle, synth
from mixlab_amd import abi
from test_gpu_audio_parity import strips
from test_gpu_schedule import gate_open

SR, SPT = 48000, 800
hip = C.CDLL("libamdhip64.so")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def bus_strips(n_strips, n_groups):
    """config-2 strips into n_groups group Mixers, those into a master Mixer(n_groups); -> ws, master, sources, triggers, first EqThree node"""
    from mixlab_amd.workspace import Workspace
    per = n_strips // n_groups
    ws = Workspace(SR, 60)
    gains = synth.uniform(10, 3 * n_strips, -24.0, 6.0)
    master = ws.mixer([(-1.0 * j, 1.0 - 0.1 * j, j % 2 == 0) for j in range(n_groups)])
    gms = [ws.mixer([(-0.5 * k, 1.0 - 0.01 * k, k % 5 == 0) for k in range(per)]) for _ in range(n_groups)]
    srcs, trigs, eq0 = [], [], None
    for k in range(per * n_groups):
        trig = ws.trigger(False); env = ws.envelope(); src = ws.source_mono()
        eq = ws.eq_three(float(gains[3 * k]), float(gains[3 * k + 1]), float(gains[3 * k + 2])); pan = ws.stereo_panner(); amp = ws.amplifier(1.0, 0.5)
        eq0 = eq if eq0 is None else eq0
        ws.connect(trig, 0, env, 0); ws.connect(src, 0, eq, 0); ws.connect(eq, 0, pan, 0); ws.connect(eq, 0, pan, 1)
        ws.connect(pan, 0, amp, 0); ws.connect(env, 0, amp, 1); ws.connect(amp, 0, gms[k // per], k % per)
        srcs.append(src); trigs.append(trig)
    for j, gm in enumerate(gms):
        ws.connect(gm, 0, master, j)
    return ws, master, srcs, trigs, eq0


def scenario(seed):
    rng = np.random.default_rng(seed)
    n_strips = int(rng.choice([64, 72, 96]))
    max_ticks = int(rng.choice([16, 24, 40]))
    n_runs = int(rng.integers(5, 9))
    contract = bool(rng.integers(0, 2))
    use_flag = bool(rng.integers(0, 2))
    lens = [int(rng.choice([1, 3, 16, max_ticks, max_ticks, int(rng.integers(16, max_ticks + 1))])) for _ in range(n_runs)]
    total = sum(lens)
    if rng.random() < 0.4:            # group buses: the tail is the bank of group Mixers AND the master above it
        ws, mix, srcs, trigs, eq0 = bus_strips(n_strips, int(rng.choice([2, 4, 8])))
    else:
        ws, mix, srcs, trigs = strips(n_strips, SR)
        eq0 = mix + 4
    n_strips = len(srcs)
    noise = [synth.noise(1000 * (seed % 97) + k, total * SPT) for k in range(n_strips)]
    cuts = {}                                           # run -> (tick in run, params)
    for r in range(n_runs):
        if lens[r] >= 4 and rng.random() < 0.25:
            cuts[r] = (int(rng.integers(1, lens[r])), abi.EqThreeParams(float(rng.uniform(-12, 6)), float(rng.uniform(-12, 6)), float(rng.uniform(-12, 6))))
    # the oracle, ticked
    import contextlib
    og = oracle.OracleGraph(ws)
    want, t = [], 0
    with (oracle.fp_contract() if contract else contextlib.nullcontext()):
      for r in range(n_runs):
          m, c = [], []
          for kk in range(lens[r]):
              if r in cuts and cuts[r][0] == kk:
                  og.update_params(eq0, cuts[r][1])
              for k, tr in enumerate(trigs):
                  og.update_params(tr, abi.TriggerParams(1 if gate_open(t, k) else 0))
              for k, s in enumerate(srcs):
                  og.set_source(s, noise[k][t * SPT:(t + 1) * SPT])
              og.run_tick(t)
              m.append(og.output(mix, 0).copy()); c.append(og.output(mix, 1).copy())
              t += 1
          want.append((np.concatenate(m), np.concatenate(c)))
    flags = (abi.FLAG_FP_CONTRACT if contract else 0) | (abi.FLAG_OVERLAP_TAIL if use_flag else 0)
    g = ws.build(max_ticks_per_run=max_ticks, flags=flags)
    tail = g.tail_stream()
    assert tail is not None, "the mode did not come up"
    pm = pc = None
    if use_flag:
        pm, fpt = g.output_device_ptr(mix, 0); pc, _ = g.output_device_ptr(mix, 1)
    t, prev_none = 0, False                             # prev_none: nothing joined the streams after the previous run (its bank was still held when this run was queued)
    checked = seen_released = 0
    for r in range(n_runs):
        n = lens[r]
        keep, events = [], []
        for k, tr in enumerate(trigs):
            g.update_params(tr, abi.TriggerParams(1 if gate_open(t, k) else 0))        # (a Trigger update does not join the streams)
            for cc in range(1, n):
                if gate_open(t + cc, k) != gate_open(t + cc - 1, k):
                    p = abi.TriggerParams(1 if gate_open(t + cc, k) else 0); keep.append(p)
                    events.append(abi.ParamEvent(tr, cc, C.cast(C.pointer(p), C.c_void_p), C.sizeof(p)))
        if events:
            g.schedule_params_batch((abi.ParamEvent * len(events))(*events))
        if r in cuts:
            g.schedule_params(eq0, cuts[r][0], cuts[r][1])
        for k, s in enumerate(srcs):
            g.write_source(s, noise[k][t * SPT:(t + n) * SPT], n)
        g.run_ticks(t, n)
        # the previous run's buses, where only a bank released by THIS run can be seen: a copy on the tail stream (flag mode; this run's own bank is still held or -- a short or
        # cut run -- already behind it on that stream, so the copy is queued BEFORE anything else is asked of the graph)
        if use_flag and prev_none and r not in cuts and r >= 1:
            n_prev = lens[r - 1]
            got = np.empty(2 * SPT * n_prev, np.float32)
            for ptr, w, nm in ((pm, want[r - 1][0], "master"), (pc, want[r - 1][1], "cue")):
                assert hip.hipMemcpyAsync(got.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(got.nbytes), 2, C.c_void_p(tail)) == 0
                assert hip.hipStreamSynchronize(C.c_void_p(tail)) == 0
                if not np.array_equal(bits(got), bits(w)):
                    raise AssertionError(f"run {r - 1} {nm}: differs (released by run {r}, read on the tail stream)")
            checked += 1; seen_released += 1
        act = rng.choice(["read", "sync", "tail", "none", "none", "none"])
        prev_none = act == "none" or (act == "tail" and not use_flag)
        if act == "read":
            for port, w in ((0, want[r][0]), (1, want[r][1])):
                got = g.read_output(mix, port, n, True)
                if not np.array_equal(bits(got), bits(w)):
                    raise AssertionError(f"run {r} port {port}: {int((bits(got) != bits(w)).sum())} samples differ (read-back)")
            checked += 1
        elif act == "sync":
            g.sync()
        elif act == "tail" and use_flag:
            ts = g.tail_stream()                                                      # releases this run's bank
            got = np.empty(2 * SPT * n, np.float32)
            for ptr, w, nm in ((pm, want[r][0], "master"), (pc, want[r][1], "cue")):
                assert hip.hipMemcpyAsync(got.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(got.nbytes), 2, C.c_void_p(ts)) == 0
                assert hip.hipStreamSynchronize(C.c_void_p(ts)) == 0
                if not np.array_equal(bits(got), bits(w)):
                    raise AssertionError(f"run {r} {nm}: differs (copy on the tail stream)")
            checked += 1
        t += n
    for port, w in ((0, want[-1][0]), (1, want[-1][1])):
        got = g.read_output(mix, port, lens[-1], True)
        if not np.array_equal(bits(got), bits(w)):
            raise AssertionError(f"last run port {port} differs")
    gated, at_once = g.debug_tail_releases()
    g.close()
    return checked + 1, gated, at_once, seen_released


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    fails = 0; tot_checked = tot_gated = tot_once = tot_seen = 0
    for seed in range(first, first + count):
        try:
            ch, ga, ao, sr = scenario(seed)
            tot_checked += ch; tot_gated += ga; tot_once += ao; tot_seen += sr
        except Exception as e:   # noqa: BLE001
            fails += 1
            print(f"seed {seed}: FAIL {e}"); traceback.print_exc()
    print(f"{count} scenarios from seed {first}, {fails} failures; runs checked {tot_checked}; banks released behind a gate {tot_gated} (or by the next run without one), at once {tot_once}; of those released by the next run, read on the tail stream and compared: {tot_seen}")
    sys.exit(1 if fails else 0)


main()
