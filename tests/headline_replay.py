"""Replay of bench.py's headline job against the CPU oracle AT ITS OWN SHAPE -- TEST INFRASTRUCTURE (the checker), never the product.

bench.py times 1024 strips x 2048 ticks per submission; the kernel instantiation the planner picks for that shape is not the one it picks
for 16 strips or for 6 ticks.  This module checks the outputs of exactly the timed submissions:

* a sample of strips (their fused Trigger -> Envelope / EqThree -> StereoPanner -> Amplifier output, the Mixer's input port) is replayed
  through the oracle's graph runner from tick 0 -- state carried across every submission, gates toggling between ticks as scheduled
  (ModuleT::update between two ticks, src/engine.rs:192-214, 277-398), sources re-read every submission -- and compared BIT FOR BIT with
  what the device left for the last submission (exact equality is the reference's own bar, src/module/eq_three.rs:150-167);
* Master and Cue of sampled ticks of that submission are compared with the oracle Mixer (src/module/mixer.rs:46-74) fed with the DEVICE's
  own strip outputs: the mix is the ordered f32 sum of the strips, so strips checked + ordered sum checked = buses checked.

One oracle graph per sampled strip, one thread per strip (ctypes releases the GIL inside orc_graph_run_ticks).
"""
from __future__ import annotations

import threading

import numpy as np


def gate_open(tick, k):
    """the Trigger of strip k toggles every 30 ticks with phase k mod 60."""
    return ((tick + k) // 30) % 2 == 1


def sample_strips(n_strips, k, seed=0x4D58):
    """k distinct strips: the first, the last, and a seeded random choice of the rest."""
    k = min(k, n_strips)
    rng = np.random.default_rng(seed)
    pick = {0, n_strips - 1}
    while len(pick) < k:
        pick.add(int(rng.integers(0, n_strips)))
    return sorted(pick)


def _replay_one(oracle, abi, ws, nodes, k_global, src_ring, ring_ticks, n_total, n_keep, toggling, out, idx):
    """Strip `k_global` alone (a one-strip Workspace with the job's seeded parameters) from tick 0 to n_total; keeps the Amplifier output of the
    last n_keep ticks."""
    mix, src, trig, amp = nodes
    og = oracle.OracleGraph(ws)
    og.set_source_ring(src, src_ring, ring_ticks)
    spt = ws.spt
    res = np.empty(n_keep * 2 * spt, dtype=np.float32)
    p_open, p_closed = abi.TriggerParams(1), abi.TriggerParams(0)
    t = 0
    first_keep = n_total - n_keep
    while t < first_keep:                                          # whole stretches between two toggles in one foreign call
        nxt = min(first_keep, t + (30 - (t + k_global) % 30)) if toggling else first_keep
        og.run_ticks(t, nxt - t)
        t = nxt
        if toggling and (t + k_global) % 30 == 0:
            og.update_params(trig, p_open if gate_open(t, k_global) else p_closed)
    for t in range(first_keep, n_total):
        if toggling and t > 0 and t != first_keep and (t + k_global) % 30 == 0:
            og.update_params(trig, p_open if gate_open(t, k_global) else p_closed)
        og.run_tick(t)
        res[(t - first_keep) * 2 * spt:(t - first_keep + 1) * 2 * spt] = og.output(amp, 0)
    out[idx] = res


def replay_and_compare(g, build_one_strip, strip_ids, first_strip, src_host, T, n_steps_run, mix_node, amp_node_of, toggling=True,
                       contract=False, check_ticks=6, all_amp_nodes=None, mixer_channels=None, seed=0x4D58):
    """g: the device graph, its LAST submission being step n_steps_run - 1 of T ticks (submission i covers ticks [i T, (i + 1) T) from tick 0).
    build_one_strip(k_global) -> (ws, (mix, src, trig, amp)): a Workspace holding strip k alone, same parameters as in the job.
    strip_ids: local indices (into the rank's strips) to replay; src_host[j]: the T-tick source buffer of local strip j as uploaded.
    all_amp_nodes / mixer_channels: every local strip's Amplifier node and the Mixer's (gain_db, fader, cue) rows, for the ordered-sum check.
    Returns the `headline_parity` record."""
    import oracle
    from mixlab_amd import abi

    n_total = n_steps_run * T
    out = [None] * len(strip_ids)
    threads = []
    with oracle.fp_contract(contract):
        for idx, j in enumerate(strip_ids):
            ws, nodes = build_one_strip(first_strip + j)
            th = threading.Thread(target=_replay_one, args=(oracle, abi, ws, nodes, first_strip + j, src_host[j], T, n_total, T, toggling, out, idx))
            th.start(); threads.append(th)
        for th in threads:
            th.join()
    spt = g.spt
    compared = 0
    bad_strips = []
    for idx, j in enumerate(strip_ids):
        if out[idx] is None:
            bad_strips.append({"strip": first_strip + j, "error": "oracle replay failed"})
            continue
        got = g.read_output(amp_node_of(j), 0, T, True)
        diff = np.flatnonzero(got.view(np.uint32) != out[idx].view(np.uint32))
        compared += int(got.size)
        if diff.size:
            i = int(diff[0])
            bad_strips.append({"strip": first_strip + j, "mismatching": int(diff.size), "first_index": i, "tick_in_step": i // (2 * spt),
                               "got": float(got[i]), "want": float(out[idx][i])})
    rec = {"verdict": "bit-exact" if not bad_strips else "MISMATCH", "strips_checked": len(strip_ids), "strips": [first_strip + j for j in strip_ids],
           "samples_compared": compared, "ticks_replayed_per_strip": n_total,
           "against": ("the CPU oracle's graph runner" + (" in its contract mode" if contract else "") + ", one strip per thread, replayed from tick 0 with the gate schedule; "
                       "compared with the device's fused strip outputs of the last timed submission")}
    if bad_strips:
        rec["mismatches"] = bad_strips[:4]
    # Master / Cue: the oracle Mixer over the DEVICE's own strips, on sampled ticks of the same submission
    if all_amp_nodes is not None and mixer_channels is not None and check_ticks > 0:
        rng = np.random.default_rng(seed + 1)
        ticks = sorted({0, T - 1} | {int(x) for x in rng.integers(0, T, size=max(0, check_ticks - 2))})
        bus_bad = []
        bus_cmp = 0
        for tk in ticks:
            dev = [g.read_output_window(a, 0, tk, 1, True) for a in all_amp_nodes]
            want_m, want_c = oracle.mixer_run(mixer_channels, dev, 2 * spt)
            for port, want, name in ((0, want_m, "master"), (1, want_c, "cue")):
                got = g.read_output_window(mix_node, port, tk, 1, True)
                bus_cmp += int(got.size)
                d = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
                if d.size:
                    bus_bad.append({"bus": name, "tick_in_step": tk, "mismatching": int(d.size)})
        rec["buses"] = {"verdict": "bit-exact" if not bus_bad else "MISMATCH", "ticks_checked": ticks, "samples_compared": bus_cmp,
                        "against": f"the oracle Mixer (ordered f32 sum, mixer.rs:46-74) over the device's own {len(all_amp_nodes)} strip outputs of those ticks"}
        if bus_bad:
            rec["buses"]["mismatches"] = bus_bad[:4]
            rec["verdict"] = "MISMATCH"
    return rec
