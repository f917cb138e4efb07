#!/usr/bin/env python
"""bench.py -- throughput of the per-tick module-graph hot path on MI355X.

Headline (BASELINE.json configs[1]; benchlegs/headline.py): 1024 channel strips Trigger -> Envelope ; noise -> EqThree -> StereoPanner -> Amplifier ->
Mixer(1024) at 48 kHz, gates toggling every 30 ticks, T = 2048 ticks batched per submission ("step"), inputs resident in HBM.  Metric: audio
channels mixed per second = strip-ticks (one stereo strip processed and mixed for one 1/60 s tick) per second, whole job.  N > 1 (configs[4]): the
strips are sharded over the ranks, partial buses combined by the library's exchange (RCCL inside libmixlab_gpu.so); `value` is then the policy
whose per-rank chunk length equals the one-GPU job's (T x N ticks per step), the fixed-T number beside it (benchlegs/scaling.py).

stdout carries ONE compact JSON line (benchlegs/line.py: the contract's keys, `roofline` of the dominant launch group -- its own bytes over its own
hipEvent duration inside the timed region --, `cpu_baseline` = the CPU oracle on a bounded sample, one number per secondary leg); everything the
legs measured is written to bench_full.json beside this file (stderr only says where).  The legs: benchlegs/*.py.
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from benchlegs import cpu, headline, line as bench_line, realtime, scaling, variants  # noqa: E402
from benchlegs.common import HBM_PEAK_GBS, Job, build_strips, dist_backend, dist_device, gate_events, gate_open  # noqa: E402,F401  (re-exported: tests and tools import them from here)
from benchlegs.fir import fir_leg  # noqa: E402
from benchlegs.video import video_leg  # noqa: E402

_RESULT_FD = None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--strips", type=int, default=1024)
    ap.add_argument("--ticks-per-step", type=int, default=2048, help="ticks batched per submission on one GPU (SURVEY 8d: throughput mode; 1 = real-time mode)")
    ap.add_argument("--sample-rate", type=int, default=48000)
    ap.add_argument("--fixed-ticks", action="store_true", help="N > 1: keep T = --ticks-per-step per step as the HEADLINE policy (default for N > 1: T x N ticks per step, so that a "
                    "rank's chunk length -- and the share of warm-up samples its speculative EqThree runs -- stays what it is on one GPU); the line carries the other policy beside it")
    ap.add_argument("--scale-ticks", action="store_true", help="(accepted for older command lines: T x N is the default policy for N > 1)")
    ap.add_argument("--eq-fast", action="store_true", help="MX_FLAG_EQ_FAST: the time-parallel EqThree scan (<= 1 ULP, NOT bit-exact) instead of the exact default")
    ap.add_argument("--fp-contract", action="store_true", help="run the HEADLINE in the contracted order (MX_FLAG_FP_CONTRACT: <= 1 ULP, NOT the reference's bits)")
    ap.add_argument("--overlap-tail", action="store_true", help="MX_FLAG_OVERLAP_TAIL: the explicit form of the second-stream Mixer bank (automatic since round 5)")
    ap.add_argument("--hold-gates", action="store_true", help="no per-tick gate schedule: every gate held for the whole run (the round-1 configuration)")
    ap.add_argument("--no-fuse", action="store_true", help="materialise every port (MX_FLAG_NO_FUSE): module-boundary traffic")
    ap.add_argument("--exchange", choices=["auto", "slices", "allgather", "allreduce"], default="auto",
                    help="N > 1 bus exchange (mx_exchange_*): ordered reduce-scatter + all-gather over time slices (auto: N >= 4), one all-gather of the whole "
                         "partial buses (auto: N < 4), or ncclAllReduce (NOT the sum order of a reference graph: non-parity)")
    ap.add_argument("--force-combine", action="store_true", help="run the N > 1 exchange path at N = 1 (single-rank RCCL group)")
    ap.add_argument("--repeats", type=int, default=4, help="further repetitions of the K timed steps after the headline region (spread of the clock)")
    ap.add_argument("--parity-strips", type=int, default=16, help="strips replayed through the CPU oracle from tick 0 for headline_parity")
    ap.add_argument("--fir-ticks", type=int, default=128, help="ticks per step of the FIR + resampler leg (BASELINE configs[2]; 0 = skip)")
    ap.add_argument("--video-frames", type=int, default=4096, help="composited frames in the video leg (0 = skip)")
    ap.add_argument("--video-band-as", default=None, metavar="R/W", help="single GPU: run the video leg as rank R of a W-rank row-band job")
    ap.add_argument("--video-shard", choices=["replicas", "bands"], default="replicas",
                    help="N > 1: independent 8-layer streams per rank (weak scaling), or ONE stream cut into row bands over the ranks (strong scaling)")
    ap.add_argument("--full-out", default=str(ROOT / "bench_full.json"), help="where rank 0 writes everything the legs measured")
    ap.add_argument("--headline-only", action="store_true", help="every secondary leg off (profiler passes)")
    for flag, what in (("cpu-baseline", "the CPU oracle timed on this host"), ("buses-leg", "the group-bus topology leg"), ("one-stream-leg", "the MX_OVERLAP_AUTO=0 leg"),
                       ("held-leg", "the held-gates comparison leg"), ("other-policy-leg", "N > 1: the second tick policy"), ("scaling-probe", "the one-GPU model of 2 / 4 / 8 ranks"),
                       ("profile", "per-kernel hipEvents in the timed region (roofline omitted)"), ("t-sweep", "the shorter-submission legs"), ("realtime", "the one-tick-per-submission leg"),
                       ("north-star", "the 10 240-strip + 8-layer real-time leg"), ("contract-leg", "the MX_FLAG_FP_CONTRACT leg"), ("rate-leg", "the 44.1 kHz leg"),
                       ("headline-parity", "the oracle replay of the timed submissions"), ("material-leg", "the realistic-material and poisoned-strip legs")):
        ap.add_argument(f"--no-{flag}", action="store_true", help=f"skip {what}")
    ap.add_argument("--no-scaled-leg", dest="no_other_policy_leg", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    if args.headline_only:
        for k in ("cpu_baseline", "buses_leg", "one_stream_leg", "held_leg", "other_policy_leg", "scaling_probe", "t_sweep", "realtime", "north_star", "contract_leg", "rate_leg",
                  "material_leg"):
            setattr(args, "no_" + k, True)
        args.fir_ticks = args.video_frames = args.repeats = 0
    return args


def main(argv=None):
    # stdout carries the result line and nothing else: file descriptor 1 is pointed at stderr for the whole run (libraries print banners from C code, past
    # sys.stdout) and the line is written to the original descriptor at the end
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)
    args = parse_args(argv)

    import numpy as np
    import torch
    import torch.distributed as dist

    import synth
    from mixlab_amd import abi, shard
    from mixlab_amd.workspace import Workspace

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("MX_BENCH_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))   # (tests: every rank on GPU 0, with MX_BENCH_DIST_BACKEND=gloo and MX_RCCL_LIB)
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_combine
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if dist_backend() == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(dist_backend(), rank=rank, world_size=world)

    # N > 1 tick policy, decided here and named in config.ticks_policy: the headline runs T x N ticks per step unless --fixed-ticks
    scaled = world > 1 and not args.fixed_ticks
    if scaled:
        args.parity_strips = max(2, args.parity_strips // world)    # the oracle replays a strip from tick 0: N x the audio per step, so 1 / N of the strips (seconds, on rank 0, outside every clock)
    T, SR = args.ticks_per_step * (world if scaled else 1), args.sample_rate
    first, local_strips = shard.strip_range(rank, world, args.strips)
    flags = (abi.FLAG_EQ_FAST if args.eq_fast else 0) | (abi.FLAG_NO_FUSE if args.no_fuse else 0) | (abi.FLAG_FP_CONTRACT if args.fp_contract else 0)
    overlap = args.overlap_tail and not use_dist        # the exchange packs the buses on the compute stream: one stream there
    if overlap:
        flags |= abi.FLAG_OVERLAP_TAIL
    job = Job(torch=torch, dist=dist, abi=abi, shard=shard, Workspace=Workspace, synth=synth, args=args, rank=rank, world=world, local_rank=local_rank,
              stream=torch.cuda.Stream(), use_dist=use_dist, T=T, SR=SR, spt=SR // 60, first=first, local_strips=local_strips, toggling=not args.hold_gates, flags=flags)
    headline.setup(job)
    g, mix = job.g, job.mix

    def fresh_id():
        from mixlab_amd.exchange import unique_id
        box = [unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        return box[0]

    ex = None
    if use_dist:
        # the exchange is the library's (mx_exchange_*: RCCL called from libmixlab_gpu.so); torch.distributed only carries the job's ncclUniqueId from
        # rank 0 to the others, the barriers and the max-over-ranks of the clock
        from mixlab_amd.exchange import BusExchange
        ex = BusExchange(g, mix, T, rank, world, mode=args.exchange, nccl_id=fresh_id())

    # every step's gate toggles, built before any clock starts (the schedule is host data, like the params a UI would send)
    n_regions = 1 + (max(0, args.repeats) if not use_dist else 0)
    events = {i: job.events(i) for i in range(args.warmup + args.steps * n_regions + 4)}

    def step(i, scheduled=True):
        if scheduled and events[i] is not None:
            g.schedule_params_batch(events[i][0], events[i][1])
        g.run_ticks(i * T, T)
        if ex is not None:
            ex.submit(i)

    def barrier():
        if world > 1:
            dist.barrier()

    def timed_region(i0, k, scheduled=True):
        # g.sync() = mx_graph_sync: every launch of the graph, on both of its streams, INCLUDING a Mixer bank the library holds back for the next run's EqThree
        # launch: the region starts with nothing of this graph outstanding and ends when all K steps' launches have completed
        g.sync(); torch.cuda.synchronize(); barrier()
        t0 = time.perf_counter()
        for i in range(k):
            step(i0 + i, scheduled)
        g.sync(); torch.cuda.synchronize(); barrier()
        return time.perf_counter() - t0

    def parity_of(g2, n_steps_run, contract, sample_rate=SR, mix2=None, src_of=None):
        return headline.headline_parity(job, g2, sample_rate, n_steps_run, mix if mix2 is None else mix2, contract, src_of or headline.source_of(job))

    want_parity = not args.no_headline_parity and not args.eq_fast and not args.no_fuse and rank == 0
    full = {}
    with torch.cuda.stream(job.stream):
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        # per-kernel hipEvents cost a few us of stream time each: at N = 1 they sit inside the timed region (the roofline contract), at N > 1 -- where a
        # step is shorter -- they are taken on extra steps after it
        g.profile_enable(not args.no_profile and not use_dist)
        dt = timed_region(args.warmup, args.steps)                      # THE timed region: exactly K steps
        nxt = args.warmup + args.steps
        n_run = nxt                                                    # submissions the graph has run so far (the oracle replay compares the LAST one)
        if use_dist and not args.no_profile:
            g.profile_enable(True)
            for i in range(3):
                step(nxt + i)
            torch.cuda.synchronize()
            n_run = nxt + 3
        g.profile_enable(False)
        by_kind, _prof_total_ms, n_prof = g.profile_collect()
        spec_ran, spec_repaired = g.eq_spec_stats()
        # the timed submissions against the oracle at their own shape (outside every clock; before anything else overwrites the last step's outputs)
        parity = parity_of(g, n_run, bool(args.fp_contract)) if want_parity else None
        rep_ms = [dt / args.steps * 1e3]                               # the spread of the clock: the same K steps again, a few times (not part of `value`)
        if not use_dist:
            for _ in range(max(0, args.repeats)):
                rep_ms.append(timed_region(nxt, args.steps) / args.steps * 1e3)
                nxt += args.steps
        if job.toggling and not use_dist and not args.no_held_leg:     # the same job with every gate held where it stands (round 1 measured this)
            nh = min(args.steps, 10)
            g.profile_enable(not args.no_profile)
            dt_h = timed_region(nxt, nh, scheduled=False)
            g.profile_enable(False)
            hk, _ht, hn = g.profile_collect()
            full["held_gates"] = {"ms_per_step": dt_h / nh * 1e3, "value": args.strips * T * nh / dt_h, "unit": "channel-ticks/s",
                                  "kernel_ms_per_step": {k: round(v / max(1, hn), 5) for k, v in sorted(hk.items()) if v > 0}}
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dist_device())
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        job.nxt, job.dt = nxt, dt
        ms_per_step = dt / args.steps * 1e3
        overlap_active = g.tail_stream() is not None
        solo = not use_dist

        def leg(name, fn):
            """A secondary leg of the ONE-GPU run must not cost the run its line: its error goes to stderr and into `leg_errors`, the headline stands.  (N > 1 legs are
            collective: an exception there is every rank's, and is raised.)"""
            try:
                if os.environ.get("MX_BENCH_FAIL_LEG") == name:
                    raise RuntimeError("injected by MX_BENCH_FAIL_LEG (tests)")
                full[name] = fn()
            except Exception as e:
                import traceback
                traceback.print_exc(limit=8, file=sys.stderr)
                full[name] = None
                full.setdefault("leg_errors", {})[name] = f"{type(e).__name__}: {e}"[:300]

        if solo and not args.no_contract_leg and not args.eq_fast and not args.fp_contract:
            leg("fp_contract", lambda: variants.contract_leg(job, (lambda g2, n, c: parity_of(g2, n, c)) if parity is not None else None))
        if solo and overlap_active and not args.no_profile and not args.no_one_stream_leg:
            leg("one_stream", lambda: variants.one_stream_leg(job))
        if ex is not None:
            full["exchange"] = headline.exchange_section(job, ex, nxt)
            if not args.no_other_policy_leg:
                T2 = args.ticks_per_step * (1 if scaled else world)
                full["other_policy"] = scaling.other_policy_leg(job, T2, "fixed ticks per step (a rank's chunks shrink with N)" if scaled else
                                                                "ticks per step scaled with N (a rank's chunks keep their one-GPU length)", fresh_id)
        if solo and not args.no_buses_leg and not args.no_fuse and args.strips % 8 == 0:
            leg("group_buses", lambda: variants.buses_leg(job))
        if solo and not args.no_realtime:
            leg("realtime", lambda: realtime.realtime_leg(job))
        if solo and not args.no_t_sweep:
            leg("t_sweep", lambda: realtime.t_sweep_leg(job))
        if solo and not args.no_rate_leg and job.toggling and SR != 44100:
            pf = (lambda g2, sr, n, mix2, src: parity_of(g2, n, bool(args.fp_contract), sr, mix2, src)) if want_parity else None
            leg("rate_44100", lambda: variants.other_rate_leg(job, 44100, pf))
        if solo and not args.no_material_leg:
            leg("material", lambda: variants.material_leg(job, step, timed_region, events))
        if solo and not args.no_scaling_probe and not args.no_fuse and args.strips % 8 == 0:
            leg("scaling_model", lambda: scaling.scaling_probe(job, ms_per_step))
        if args.video_frames > 0:
            def run_video():
                return video_leg(torch, dist, world, job.stream, local_rank, args.video_frames, args.warmup, shard_mode=args.video_shard, rank=rank,
                                 band_as=tuple(int(x) for x in args.video_band_as.split("/")) if args.video_band_as else None)
            if solo:
                leg("video", run_video)
            else:
                full["video"] = run_video()
        if solo and not args.no_north_star:
            leg("north_star_realtime", lambda: realtime.north_star_leg(job))
        if solo and args.fir_ticks > 0:
            leg("fir_resample", lambda: fir_leg(torch, job.stream, local_rank, args.fir_ticks, 10, 2))

    if rank == 0:
        k_ms = {k: v / n_prof for k, v in by_kind.items() if v > 0} if n_prof else {}
        if full.get("fp_contract"):
            full["fp_contract"]["roofline"] = headline.contract_roofline(job, full["fp_contract"])
            full["fp_contract"]["speedup_vs_exact"] = round(ms_per_step / full["fp_contract"]["ms_per_step"], 3)
        rep_sorted = sorted(rep_ms)
        audio_s = T / 60.0
        policy = ("one GPU" if world == 1 else
                  (f"T x N = {T} ticks per step (a rank's chunks keep their one-GPU length; a step is {audio_s:.0f} s of audio, so results arrive {world} x {audio_s / world:.0f} s late)" if scaled
                   else f"fixed T = {T} ticks per step (a rank's chunks shrink with N: warm-up share grows)"))
        out = {
            "metric": "audio_ch_mixed_per_sec", "value": args.strips * T * args.steps / dt, "unit": "channel-ticks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32 (f64 intermediates)", "data": "synthetic",
            "config": {"workload": f"{args.strips}-channel Mixer + EqThree + Envelope chain (Trigger->Envelope; noise->EqThree->StereoPanner->Amplifier->Mixer), {SR} Hz f32",
                       "strips": args.strips, "ticks_per_step": T, "samples_per_tick": job.spt, "sample_rate": SR,
                       "gates": "toggle every 30 ticks, phase k mod 60, applied between ticks inside the batch" if job.toggling else "held for the whole run",
                       "eq_mode": "time-parallel scan (<= 1 ULP, MX_FLAG_EQ_FAST)" if args.eq_fast else ("CONTRACTED order (<= 1 ULP of the reference, NOT its bits)" if args.fp_contract
                                   else "exact order: speculative time-parallel kernel, verified bit-exact"),
                       "fusion": "off (every port materialised)" if args.no_fuse else "Trigger+Envelope+EqThree+StereoPanner+Amplifier in one kernel, L==R strips stored mono",
                       "overlap": "the Mixer bank of step k runs on a second stream beside step k + 1's EqThree group" if (overlap or overlap_active) else "off",
                       "parallelism": f"strips sharded x{world}" + (f", {ex.mode}" if ex is not None else ""), "ticks_policy": policy,
                       "rccl_ranks": ex.world if ex is not None else 0},
            "realtime_channels_equiv": args.strips * T * args.steps / dt / 60.0,
            "eq_spec": {"chunks_run": int(spec_ran), "chunks_repaired": int(spec_repaired)},
            "headline_parity": parity,
            "roofline": headline.roofline(job, k_ms, overlap_active, full.get("one_stream"), ms_per_step),
            "repeats": {"ms_per_step": [round(v, 4) for v in rep_ms], "median": round(rep_sorted[len(rep_sorted) // 2], 4),
                        "spread_pct": round((rep_sorted[-1] - rep_sorted[0]) / rep_sorted[len(rep_sorted) // 2] * 100.0, 2)},
        }
        out.update(full)
        out["cpu_baseline"] = None
        if not args.no_cpu_baseline and world == 1:
            note = cpu.native_oracle()
            if out.get("video"):
                out["video"]["cpu_baseline"] = dict(cpu.video(), build=note)
            if out.get("fir_resample"):
                out["fir_resample"]["cpu_baseline"] = dict(cpu.fir(), build=note)
            out["cpu_baseline"] = cpu.audio(Workspace, synth, abi, args.strips, SR, note)
            out["cpu_baseline_all_cores"] = dict(cpu.audio_all_cores(Workspace, synth, abi, shard, args.strips, SR), build=note)
        full_text = json.dumps(out, allow_nan=False)
        try:
            pathlib.Path(args.full_out).write_text(full_text + "\n")
            full_ref = os.path.relpath(args.full_out, ROOT) if str(args.full_out).startswith(str(ROOT)) else str(args.full_out)
        except OSError as e:
            full_ref = f"(not written: {e})"
        print(f"[bench.py] everything the legs measured: {full_ref} ({len(full_text)} bytes)", file=sys.stderr)   # (a pointer, not the record: a reader that keeps a tail of the merged streams must still find the line)
        text = bench_line.dumps_within_cap(bench_line.compact(out, full_ref))
        # RCCL prints a version banner through C stdio (flushed at exit when stdout is a pipe): drain it first so the JSON line is the LAST thing on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush(); sys.stderr.flush()
        os.write(_RESULT_FD if _RESULT_FD is not None else 1, (text + "\n").encode())

    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
