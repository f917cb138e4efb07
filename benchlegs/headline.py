"""The headline job -- BASELINE.json configs[1] (SURVEY.md section 8d config 2 AS WRITTEN): 1024 channel strips
  Trigger -> Envelope ;  Source(noise) -> EqThree -> StereoPanner(L=R) -> Amplifier(ctl = Envelope) -> Mixer(1024)
at 48 kHz (SPT = 800), every strip's gate toggling every 30 ticks with phase k mod 60 -- applied BETWEEN ticks of the batch through
mx_graph_schedule_params_batch (the reference's client_update between two ticks, src/engine.rs:192-214) -- T ticks batched per submission ("step" =
one pass of the whole graph over T ticks of synthetic input already resident in HBM).  EqThree runs in the reference's exact order (the library
default).  N > 1 (configs[4]): the strips are sharded contiguously over the ranks (strong scaling), each rank runs Mixer(1024/N) over its strips,
and the partial buses are combined by the library's exchange (rank-ordered sum = the reference-expressible graph N x Mixer(1024/N) -> Mixer(N))."""
from __future__ import annotations

import time

import numpy as np

from .common import (dist_device, BYTES_PER_FRAME, BYTES_PER_FRAME_FUSED, F64_VALU_PEAK_TOPS, HBM_PEAK_GBS, PROFILE_TAG, build_strips, pmc_traffic, sq_profile,
                     sustained_clock_ghz, tiled_noise)
from .scaling import exchange_parity


def setup(job):
    """Build the rank's graph and upload its synthetic sources (resident in HBM before any clock starts; re-read every step)."""
    job.ws, job.mix, job.srcs, job.trigs = build_strips(job.abi, job.Workspace, job.synth, job.local_strips, job.first, job.SR, want_trigs=True)
    job.g = job.build()
    for j, s in enumerate(job.srcs):
        job.g.write_source(s, tiled_noise(job.synth, job.first + j, job.T, job.spt), job.T)


def source_of(job, spt=None):
    return lambda j: tiled_noise(job.synth, job.first + j, job.T, job.spt if spt is None else spt)


def headline_parity(job, g, sample_rate, n_steps_run, mix, contract, src_of):
    """The timed submissions' outputs against the CPU oracle at the job's own shape (tests/headline_replay.py): a sample of strips replayed from tick 0
    and compared bit for bit with the last submission's fused strip outputs; Master / Cue of sampled ticks against the oracle Mixer over the device's
    own strips.  Runs AFTER a timed region, outside every clock."""
    import headline_replay as hr    # test infrastructure: the checker

    abi, args, synth, T = job.abi, job.args, job.synth, job.T
    total = max(1024, args.strips)
    ids = hr.sample_strips(job.local_strips, args.parity_strips)
    mg = synth.uniform(11, total, -24.0, 6.0)
    mf = synth.uniform(12, total, 0.0, 1.0)

    def one(k):
        ws1, mix1, srcs1, trigs1 = build_strips(abi, job.Workspace, synth, 1, k, sample_rate, total=total, want_trigs=True)
        return ws1, (mix1, srcs1[0], trigs1[0], mix1 + 6)

    t0 = time.perf_counter()
    rec = hr.replay_and_compare(g, one, ids, job.first, {j: src_of(j) for j in ids}, T, n_steps_run, mix, lambda j: mix + 6 * j + 6, toggling=job.toggling,
                                contract=contract, check_ticks=6, all_amp_nodes=[mix + 6 * j + 6 for j in range(job.local_strips)],
                                mixer_channels=[(float(mg[k]), float(mf[k]), k % 8 == 0) for k in range(job.first, job.first + job.local_strips)])
    rec["shape"] = f"{job.local_strips} strips x {T} ticks per submission @ {sample_rate} Hz, submission {n_steps_run - 1} (the last one timed)"
    rec["seconds"] = round(time.perf_counter() - t0, 2)
    return rec


def moved_bytes_fn(job):
    bpf = BYTES_PER_FRAME if job.args.no_fuse else BYTES_PER_FRAME_FUSED
    frames = job.T * job.spt

    def moved_bytes(kind):   # bytes one launch of this kind's group has to move on this rank
        if kind == "mixer":
            return (bpf["mixer"] * job.local_strips + 16) * frames
        return bpf.get(kind, 0) * job.local_strips * frames
    return moved_bytes


def roofline(job, k_ms, overlap_active, one_stream, ms_per_step):
    """`roofline` of the line: the launch group that took the most device time in the timed region, ITS OWN algorithmic bytes over ITS OWN average
    duration (hipEvents on the graph's stream, recorded inside the timed region).  When the Mixer bank of the step before runs beside it (DESIGN.md
    5.2) the two-kernel figure is `window`, and `step_hbm_frac` is every byte the step moves over the step's wall time."""
    if not k_ms:
        return None
    args, use_dist = job.args, job.use_dist
    moved_bytes = moved_bytes_fn(job)
    dom = max(k_ms, key=k_ms.get)
    avg_ms, alg = k_ms[dom], moved_bytes(dom)
    ach = alg / (avg_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(dom, args, job.world, job.toggling)
    per_kernel = {k: {"moved_bytes_per_launch": moved_bytes(k), "ms": round(ms, 5), "hbm_frac": round(moved_bytes(k) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                  for k, ms in sorted(k_ms.items()) if moved_bytes(k)}
    group = "" if args.no_fuse or dom == "mixer" else " launch group (fused Trigger + Envelope + EqThree + StereoPanner + Amplifier: k_env_ticks + k_eq_three_spec_tiled + k_eq_three_repair)"
    roof = {"kernel": dom + group, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(avg_ms, 5), "algorithmic_bytes_per_launch": alg,
            "algorithmic_bytes_per_unit": "2M = 8 B per sample per strip (SURVEY 8d: EqThree channel-tick; source read + strip written as one float per frame)",
            "step_hbm_frac": round(sum(moved_bytes(k) for k in k_ms) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "kernel_ms_per_step": {k: round(v, 5) for k, v in sorted(k_ms.items())},
            "kernel_timing": "hipEvents inside the timed region" if not use_dist else "hipEvents on 3 extra steps after the timed region",
            "per_kernel": per_kernel}
    if overlap_active and dom == "eq_three" and "mixer" in k_ms:
        # the dominant launch does not have the chip to itself: the Mixer bank of the step before runs beside it from its first workgroup to (nearly) its last
        alg_mix = moved_bytes("mixer")
        roof["window"] = {"what": "the Mixer bank of the previous step runs beside this launch on the graph's second stream (held back until this launch's workgroups are placed): "
                                  "both kernels' bytes over the EqThree group's duration",
                          "eq_three_bytes": alg, "mixer_bytes": alg_mix, "frac": round((alg + alg_mix) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if one_stream is not None:
            o = one_stream["kernel_ms_per_step"]
            roof["one_stream"] = {"env": "MX_OVERLAP_AUTO=0 (each launch alone on the chip; same job, own graph, measured after the timed region)", "ms_per_step": one_stream["ms_per_step"],
                                  "per_kernel": {k: {"ms": o[k], "hbm_frac": round(moved_bytes(k) / (o[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} for k in sorted(o) if moved_bytes(k)}}
    if dom != "eq_three":
        roof["limiter"] = "hbm"
        return roof
    # the bound that applies: f64 VALU issue.  Reference arithmetic per strip-sample: EqThree 36 f64 operations (2 x 4 poles x (sub, mul, add) + VSA adds +
    # band split + gains + 2 conversions), Amplifier 6 (conversions, depth, 2 products), Envelope closed form ~13 on the ~70 % of samples where it is not
    # flat (25/500/0.8/200 ms, gates toggling every 30 ticks)
    fc = bool(args.fp_contract)
    samples = job.local_strips * job.T * job.spt
    ops = (26.0 + 5.0 + (12.0 * 0.7 if job.toggling else 0.0)) if fc else (36.0 + 6.0 + (13.0 * 0.7 if job.toggling else 0.0))
    tops = ops * samples / (avg_ms * 1e-3) / 1e12
    roof["limiter"] = "f64_valu"
    roof["f64_valu"] = {"ops_per_sample_reference": ops, "achieved_tops": round(tops, 2), "peak_tops": F64_VALU_PEAK_TOPS, "frac": round(tops / F64_VALU_PEAK_TOPS, 3),
                        "note": "the reference's f64 operations per second against the f64 VALU instruction rate at the 2.4 GHz peak clock; every chunk also re-runs a warm-up of "
                                "1 280 samples per 6 400, and the board's power limit holds the clock below 2.4 GHz under this kernel"}
    sq = sq_profile("k_eq_three_spec_tiled", fc, samples)
    if sq:
        roof["f64_valu"]["sq_profile"] = dict(sq, source=PROFILE_TAG)
    ghz = sustained_clock_ghz("k_eq_three_spec_tiled", fc)
    if ghz:
        roof["f64_valu"]["sustained_clock"] = {"ghz": ghz, "frac_at_that_clock": round(tops / (F64_VALU_PEAK_TOPS * ghz / 2.4), 3),
                                               "source": f"{PROFILE_TAG}/clock.json (a committed measurement of these kernel sources, not read live)"}
    return roof


def contract_roofline(job, contract):
    ck_ms = contract["kernel_ms_per_step"]
    if "eq_three" not in ck_ms:
        return None
    alg = moved_bytes_fn(job)("eq_three")
    sec = ck_ms["eq_three"] * 1e-3
    samples = job.local_strips * job.T * job.spt
    ops_fc = 26.0 + 5.0 + (12.0 * 0.7 if job.toggling else 0.0)      # f64 INSTRUCTIONS of the contracted order per strip-sample (an fma counts once)
    t, src = pmc_traffic("eq_three", job.args, job.world, job.toggling, fc=True)
    return {"kernel": "eq_three launch group, contracted order", "bound": "hbm", "achieved": round(alg / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(alg / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": t, "traffic_source": src, "avg_launch_ms": round(ck_ms["eq_three"], 5), "algorithmic_bytes_per_launch": alg,
            "sustained_clock_ghz": sustained_clock_ghz("k_eq_three_spec_tiled", True),
            "f64_valu": {"instructions_per_sample_contracted": ops_fc, "frac": round(ops_fc * samples / sec / 1e12 / F64_VALU_PEAK_TOPS, 3)}}


def exchange_section(job, ex, step_after):
    """What the exchange costs on its own stream (4 more steps, the library's own event pair around each) and its parity evidence."""
    torch, dist, g, T, world, rank = job.torch, job.dist, job.g, job.T, job.world, job.rank
    ex_ms_all = []
    for i in range(4):
        g.run_ticks((step_after + i) * T, T)
        ex.submit(step_after + i)
        torch.cuda.synchronize()
        ex_ms_all.append(ex.elapsed_ms(step_after + i))
    if ex.world != world:
        raise SystemExit(f"the exchange's communicator has {ex.world} ranks, the job {world}")
    exch = {"mode": ex.mode, "rccl_ranks": ex.world, "transport": "RCCL, called by libmixlab_gpu.so (mx_exchange_*)",
            "bytes_received_per_rank_per_step": ex.bytes_received_per_step(), "exchange_ms_per_step": round(sorted(ex_ms_all)[len(ex_ms_all) // 2], 4),
            "parity": "rank-ordered f32 sum (the graph N x Mixer(strips/N) -> Mixer(N))" if ex.mode != "allreduce" else "NONE: ncclAllReduce order is not a reference graph's"}
    if ex.mode != "allreduce":
        # parity evidence that needs none of the exchange's own code (collective: every rank takes part; rank 0 reports)
        exch["parity_check"] = exchange_parity(torch, dist, g, ex, job.mix, T, step_after + 3, world)
        if world > 1:
            ok = torch.tensor([1 if exch["parity_check"]["verdict"] == "bit-exact" else 0], device=dist_device())
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            exch["parity_check"]["all_ranks"] = "bit-exact" if int(ok.item()) == 1 else "MISMATCH on some rank"
    elif world > 1:
        # measured deviation of the all-reduce from the ordered sum of the same partial buses
        from mixlab_amd.exchange import BusExchange, unique_id
        box = [unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ordered = BusExchange(g, job.mix, T, rank, world, mode="allgather", nccl_id=box[0])
        g.run_ticks((step_after + 8) * T, T); ex.submit(step_after + 8); ordered.submit(0)
        torch.cuda.synchronize()
        exch["max_ulp_vs_ordered_sum"] = ex.max_ulp_vs(step_after + 8, *ordered.result(0))
        ordered.close()
    return exch
