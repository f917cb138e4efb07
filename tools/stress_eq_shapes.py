"""Shape stress of the speculative exact EqThree path (planner, ragged last chunks, repairs, inline Envelope modes): config-2 strips at
random strip counts, batch lengths, sample rates, gate periods, forced chunk counts and forced short warm-ups (so that the repair pass
works), master / cue against the oracle ticked.  Usage: python tools/stress_eq_shapes.py [first_seed] [count]"""
import ctypes as C, os, sys, pathlib, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle, synth
from mixlab_amd import abi
from test_gpu_audio_parity import strips

_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
first = int(_pos[0]) if len(_pos) > 0 else 0
count = int(_pos[1]) if len(_pos) > 1 else 100


def run(seed):
    rng = np.random.default_rng(seed)
    SR, SPT = [(44100, 735), (48000, 800)][int(rng.integers(0, 2))]
    n_strips = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 65]))
    batch = int(rng.choice([4, 5, 7, 16, 31, 64, 100, 257]))
    n_runs = 2 if batch < 100 else 1
    period = int(rng.choice([1, 2, 3, 7, 30]))
    os.environ["MX_EQ_SPEC_CHUNKS"] = str(int(rng.choice([0, 0, 2, 3, 5, 8, 13, 64])))
    os.environ["MX_EQ_SPEC_WARM"] = str(int(rng.choice([0, 0, 0, 128, 512])))
    flags = int(rng.choice([0, 0, abi.FLAG_NO_FUSE]))
    if "--fast" in sys.argv:
        flags |= abi.FLAG_EQ_FAST
    if "--contract" in sys.argv:      # MX_FLAG_FP_CONTRACT against the oracle's contract mode: bit for bit, like the exact order
        flags |= abi.FLAG_FP_CONTRACT
        oracle.lib.orc_set_fp_contract(1)
    desc = f"seed {seed}: {SR} Hz, {n_strips} strips, batch {batch} x {n_runs}, gate period {period}, chunks {os.environ['MX_EQ_SPEC_CHUNKS']}, warm {os.environ['MX_EQ_SPEC_WARM']}, flags {flags}"
    ws, mix, srcs, trigs = strips(n_strips, SR)
    og = oracle.OracleGraph(ws)
    g = ws.build(max_ticks_per_run=batch, flags=flags)
    gate = lambda tick, k: ((tick + k) // period) % 2 == 1
    noise = [synth.noise(1000 * seed + k, n_runs * batch * SPT) for k in range(n_strips)]
    for r in range(n_runs):
        t0 = r * batch
        keep, events = [], []
        for k, tr in enumerate(trigs):
            g.update_params(tr, abi.TriggerParams(1 if gate(t0, k) else 0))
            for c in range(1, batch):
                if gate(t0 + c, k) != gate(t0 + c - 1, k):
                    p = abi.TriggerParams(1 if gate(t0 + c, k) else 0); keep.append(p)
                    events.append(abi.ParamEvent(tr, c, C.cast(C.pointer(p), C.c_void_p), C.sizeof(p)))
        if events:
            g.schedule_params_batch((abi.ParamEvent * len(events))(*events))
        for k, s in enumerate(srcs):
            g.write_source(s, noise[k][t0 * SPT:(t0 + batch) * SPT], batch)
        g.run_ticks(t0, batch)
        got_m, got_c = g.read_output(mix, 0, batch, True), g.read_output(mix, 1, batch, True)
        for kk in range(batch):
            tick = t0 + kk
            for k, tr in enumerate(trigs):
                og.update_params(tr, abi.TriggerParams(1 if gate(tick, k) else 0))
            for k, s in enumerate(srcs):
                og.set_source(s, noise[k][tick * SPT:(tick + 1) * SPT])
            og.run_tick(tick)
            sl = slice(kk * 2 * SPT, (kk + 1) * 2 * SPT)
            for name, got, want in (("master", got_m[sl], og.output(mix, 0)), ("cue", got_c[sl], og.output(mix, 1))):
                if "--fast" in sys.argv:      # the opt-in scan: every strip within 1 ULP, so a bus of n strips within n ULP of its own magnitude scale
                    err = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)).max()
                    scale = max(1e-6, float(np.abs(np.asarray(want, np.float64)).max()))
                    if err > n_strips * 2.0 ** -22 * max(scale, 1.0):
                        raise AssertionError(f"{desc}: {name} off by {err} (scale {scale}) on tick {tick}")
                elif not np.array_equal(np.asarray(got).view(np.uint32), np.asarray(want, np.float32).view(np.uint32)):
                    raise AssertionError(f"{desc}: {name} differs on tick {tick}")
    ran, repaired = g.eq_spec_stats()
    return ran, repaired


bad = 0; tot_ran = tot_rep = 0
for seed in range(first, first + count):
    try:
        ran, rep = run(seed); tot_ran += ran; tot_rep += rep
    except Exception:
        bad += 1; traceback.print_exc(limit=3)
        if bad >= 3:
            break
print(f"{count} shapes, {bad} failures; speculative chunks run {tot_ran}, repaired {tot_rep}")
sys.exit(1 if bad else 0)
