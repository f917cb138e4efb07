"""The headline job under other flags, topologies and material -- each on a graph of its own over the headline graph's resident sources (bound, not
copied), measured AFTER the timed region; none of them is `value`."""
from __future__ import annotations

import time

import numpy as np

from .common import build_strips, rounded, tiled_noise, timed_steps


def _evs(job, n, T=None, trigs=None, first=None):
    return [job.events(i, T=T, trigs=trigs, first=first) for i in range(n)]


def contract_leg(job, parity_fn):
    """MX_FLAG_FP_CONTRACT: every f32 within 1 ULP of the exact order, bit-exact vs the oracle's contract mode (tests/test_gpu_fp_contract.py)."""
    abi, args = job.abi, job.args
    g_fc = job.build(flags=(job.flags & ~abi.FLAG_OVERLAP_TAIL) | abi.FLAG_FP_CONTRACT)
    job.bind_resident_sources(g_fc)
    n_c = min(args.steps, 10)
    dt_c, k_ms = timed_steps(g_fc, _evs(job, 2 + n_c), job.T, 2, n_c, profile=not args.no_profile)
    ran, rep = g_fc.eq_spec_stats()
    out = {"flag": "MX_FLAG_FP_CONTRACT", "ms_per_step": dt_c / n_c * 1e3, "value": args.strips * job.T * n_c / dt_c, "unit": "channel-ticks/s", "steps": n_c,
           "kernel_ms_per_step": rounded(k_ms), "eq_spec": {"chunks_run": int(ran), "chunks_repaired": int(rep)},
           "headline_parity": parity_fn(g_fc, 2 + n_c, True) if parity_fn else None,
           "parity": "every f32 output within 1 ULP of the reference's order (NOT its bits); bit-exact vs the oracle's contract mode (tests/test_gpu_fp_contract.py)",
           "what": "the reference's f64 expressions with each multiply fused into the add that consumes it: EqThree 26 instead of 36 f64 instructions per sample "
                   "(eq_three.rs:76-88,117-124), Envelope decay and Amplifier depth() one fma each"}
    g_fc.close()
    return out


def one_stream_leg(job):
    """MX_OVERLAP_AUTO=0: every launch group on ONE stream -- what each kernel takes when it has the chip to itself."""
    abi, args = job.abi, job.args
    g1 = job.build(flags=job.flags & ~abi.FLAG_OVERLAP_TAIL, auto_overlap=False)
    job.bind_resident_sources(g1)
    n_1 = min(args.steps, 8)
    dt_1, k_ms = timed_steps(g1, _evs(job, 2 + n_1), job.T, 2, n_1, profile=True)
    g1.close()
    return {"env": "MX_OVERLAP_AUTO=0", "ms_per_step": round(dt_1 / n_1 * 1e3, 4), "value": args.strips * job.T * n_1 / dt_1, "unit": "channel-ticks/s", "steps": n_1,
            "kernel_ms_per_step": rounded(k_ms)}


def buses_leg(job):
    """The same strips mixed through GROUP BUSES (8 x Mixer(strips / 8) -> Mixer(8)): a topology the reference expresses with its own Mixer module,
    and the shape a console has.  The second-stream mode takes the bank AND the master above it as its tail."""
    abi, args, n8 = job.abi, job.args, job.args.strips // 8
    out = {}
    for label, auto in (("second_stream", True), ("one_stream", False)):
        wsb = job.Workspace(job.SR, 60); gm, sb, tb = [], [], []
        for j in range(8):
            wsb, m_, s_, t_ = build_strips(abi, job.Workspace, job.synth, n8, j * n8, job.SR, ws=wsb, total=args.strips, want_trigs=True)
            gm.append(m_); sb += s_; tb += t_
        master = wsb.mixer([(0.0, 1.0, False)] * 8)
        for j, m_ in enumerate(gm):
            wsb.connect(m_, 0, master, j)
        gb = job.build(ws=wsb, flags=job.flags & ~abi.FLAG_OVERLAP_TAIL, auto_overlap=auto)
        job.bind_resident_sources(gb, sb)
        kb = min(args.steps, 8)
        dtb, _k = timed_steps(gb, _evs(job, 2 + kb, trigs=tb, first=0), job.T, 2, kb)
        out[label] = {"ms_per_step": round(dtb / kb * 1e3, 4), "value": args.strips * job.T * kb / dtb, "unit": "channel-ticks/s", "steps": kb,
                      "mixer_groups_beside_next_eq_three": gb.tail_stream() is not None}
        gb.close()
    out["topology"] = f"8 x Mixer({n8}) -> Mixer(8, unity), {job.T} ticks per step, gates as in the headline"
    return out


def other_rate_leg(job, sample_rate, parity_fn_factory, steps=5):
    """The headline job at ANOTHER sample rate on a graph of its own -- 44.1 kHz is the reference's own rate (src/engine.rs SAMPLE_RATE), 48 kHz the
    one config 2 is written for.  Same strips, same gate schedule, same T."""
    abi, args, T = job.abi, job.args, job.T
    ws, mix, srcs, trigs = build_strips(abi, job.Workspace, job.synth, args.strips, 0, sample_rate, want_trigs=True)
    spt = ws.spt
    g = job.build(ws=ws)
    for j, sn in enumerate(srcs):
        g.write_source(sn, tiled_noise(job.synth, j, T, spt), T)
    dt, k_ms = timed_steps(g, _evs(job, steps + 1, trigs=trigs, first=0), T, 1, steps, profile=not args.no_profile)
    ran, repaired = g.eq_spec_stats()
    r_parity = parity_fn_factory(g, sample_rate, 1 + steps, mix, lambda j: tiled_noise(job.synth, j, T, spt)) if parity_fn_factory else None
    g.close()
    return {"sample_rate": sample_rate, "headline_parity": r_parity, "samples_per_tick": spt, "ticks_per_step": T, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
            "value": args.strips * T * steps / dt, "unit": "channel-ticks/s", "kernel_ms_per_step": rounded(k_ms),
            "eq_spec": {"chunks_run": int(ran), "chunks_repaired": int(repaired)},
            "note": "a channel-tick at 44.1 kHz is 735 samples against 800: per SAMPLE this is value x 735 / 800 of the headline's"}


def material_leg(job, step, timed_region, events):
    """Realistic material and the repair pass's worst case, on the headline graph itself (LAST: the poisoned strip's state stays NaN for ever).  The
    headline's sources are seeded noise, on which every chunk boundary of the speculative EqThree proves itself; a desk also carries muted strips
    (exact zeros) and programme that falls silent and comes back -- the one input class the proof fails on -- and may meet a NaN."""
    g, args, T, spt = job.g, job.args, job.T, job.spt
    out = {}
    rng = np.random.default_rng(0x4D58)
    seg = 48000 * 3                                                     # signal 3 s / silence 2 s / signal ...
    n_muted = n_gaps = 0
    for j, sn in enumerate(job.srcs):
        kind = j % 4                                                    # 0 muted, 1 programme with silences, 2 / 3 noise as in the headline
        if kind == 0:
            buf = np.zeros(T * spt, dtype=np.float32); n_muted += 1
        elif kind == 1:
            buf = tiled_noise(job.synth, job.first + j, T, spt).copy()
            pos = int(rng.integers(0, seg))
            while pos < buf.size:
                buf[pos: pos + 2 * 48000] = 0.0                         # two seconds of digital silence
                pos += seg + 2 * 48000
            n_gaps += 1
        else:
            continue
        g.write_source(sn, buf, T)
    ran0, rep0 = g.eq_spec_stats()
    rs0 = g.eq_repair_stats()
    base_i = job.nxt + 200
    n_m = min(args.steps, 10)
    for i in list(range(base_i, base_i + 2 + n_m)) + list(range(base_i + 20, base_i + 22 + n_m)):   # the schedules, before any clock starts
        events[i] = job.events(i)
    for i in range(2):
        step(base_i + i)
    g.profile_enable(not args.no_profile)
    dt_m = timed_region(base_i + 2, n_m)
    g.profile_enable(False)
    mk, _mt, mn = g.profile_collect()
    ran1, rep1 = g.eq_spec_stats()
    out["daw"] = {"what": f"{n_muted} strips muted (exact zeros), {n_gaps} with 3 s programme / 2 s digital silence alternating, the rest noise; gates toggling as in the headline",
                  "ms_per_step": round(dt_m / n_m * 1e3, 4), "value": args.strips * T * n_m / dt_m, "unit": "channel-ticks/s",
                  "kernel_ms_per_step": {k: round(v / max(1, mn), 5) for k, v in sorted(mk.items()) if v > 0},
                  "eq_spec": {"chunks_run": int(ran1 - ran0), "chunks_repaired": int(rep1 - rep0)},
                  "repair_pass": {k: v - rs0[k] for k, v in g.eq_repair_stats().items()}}
    # one strip poisoned: a NaN in its source.  Its poles are NaN from then on (the state is carried from step to step); in the step the NaN
    # arrives no chunk after it can prove itself and the repair pass fills the strip's remaining outputs with all 64 lanes of its wave
    bad = tiled_noise(job.synth, job.first + 2, T, spt).copy()
    bad[(T * spt) // 3] = np.float32("nan")
    g.write_source(job.srcs[2], bad, T)
    for i in range(2):
        step(base_i + 20 + i)
    ran2, rep2 = g.eq_spec_stats()
    dt_p = timed_region(base_i + 22, n_m)
    ran3, rep3 = g.eq_spec_stats()
    out["one_strip_poisoned_by_a_nan"] = {"ms_per_step": round(dt_p / n_m * 1e3, 4), "value": args.strips * T * n_m / dt_p, "unit": "channel-ticks/s",
                                          "eq_spec": {"chunks_run": int(ran3 - ran2), "chunks_repaired": int(rep3 - rep2)},
                                          "note": "on top of the daw material; from the second step on the poisoned strip CARRIES an all-NaN state, which stands still under any input"}
    return out
