"""configs[4]: the strips sharded over N ranks, the partial Master / Cue buses combined by the library's own exchange
(mx_exchange_*: RCCL called from libmixlab_gpu.so).  Here: the parity evidence of the exchange, the second tick policy of an N > 1 run, and the
one-GPU measurement of what a rank of a 2 / 4 / 8-GPU job computes per step (a MODEL of N > 1 until a multi-GPU node runs the job)."""
from __future__ import annotations

import time

import numpy as np

from .common import build_strips, dist_device, gate_events, tiled_noise


def exchange_parity(torch, dist, g, ex, mix, T, step, world):
    """Is the exchange's combined bus the rank-ordered f32 sum of the partial buses (the graph N x Mixer(strips / N) -> Mixer(N, unity),
    src/module/mixer.rs:57-68: master starts at +0.0 and adds channel after channel)?  Checked without any of the exchange's own code: every
    rank's raw partial Master / Cue (read back from its graph) travels through ONE plain all_gather of torch.distributed (ncclAllGather), the
    sum is made on the host in rank order with numpy f32 adds, and compared bit for bit with mx_exchange_read_result.  Collective: every
    rank calls it; returns this rank's verdict."""
    part = np.concatenate([g.read_output(mix, 0, T, True), g.read_output(mix, 1, T, True)])
    mine = torch.from_numpy(part).to(dist_device())
    if world > 1:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        parts = [q.cpu().numpy() for q in parts]
    else:
        parts = [part]
    acc = np.zeros_like(part)                      # util::zero, then `master[i] += ...` per channel in order (mixer.rs:54-68); x * 1.0 is x
    for q in parts:
        acc = acc + q
    got_m, got_c = ex.result(step)
    got = np.concatenate([got_m, got_c])
    bad = np.flatnonzero(got.view(np.uint32) != acc.view(np.uint32))
    if bad.size == 0:
        return {"verdict": "bit-exact", "samples_compared": int(got.size), "against": f"host sum in rank order of {len(parts)} partial buses gathered by a plain ncclAllGather"}
    i = int(bad[0])
    return {"verdict": "MISMATCH", "samples_compared": int(got.size), "mismatching": int(bad.size), "first_index": i, "got": float(got[i]), "want": float(acc[i])}


def other_policy_leg(job, T, label, nccl_id_fn):
    """N > 1: the OTHER tick policy beside the one the headline ran, on a graph and an exchange of its own; barrier + max over ranks like the headline."""
    from mixlab_amd.exchange import BusExchange
    torch, dist, abi, args = job.torch, job.dist, job.abi, job.args
    ws, mix, srcs, trigs = build_strips(abi, job.Workspace, job.synth, job.local_strips, job.first, job.SR, want_trigs=True)
    g = job.build(ws=ws, T=T, flags=job.flags & ~abi.FLAG_OVERLAP_TAIL)
    for j, sn in enumerate(srcs):
        g.write_source(sn, tiled_noise(job.synth, job.first + j, T, job.spt), T)
    ex = BusExchange(g, mix, T, job.rank, job.world, mode=args.exchange, nccl_id=nccl_id_fn())
    steps, warm = min(args.steps, 6), 2
    events = [gate_events(abi, trigs, job.first, i * T, T) if job.toggling else None for i in range(warm + steps + 1)]

    def step(i):
        if events[i] is not None:
            g.schedule_params_batch(events[i][0], events[i][1])
        g.run_ticks(i * T, T)
        ex.submit(i)
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    if job.world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warm + i)
    torch.cuda.synchronize()
    if job.world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    parity = exchange_parity(torch, dist, g, ex, mix, T, warm + steps - 1, job.world) if ex.mode != "allreduce" else None
    if job.world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dist_device())
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    out = {"policy": label, "ticks_per_step": T, "steps": steps, "ms_per_step": dt / steps * 1e3,
           "value": args.strips * T * steps / dt, "unit": "channel-ticks/s", "exchange_mode": ex.mode, "parity": parity}
    ex.close(); g.close()
    return out


def scaling_probe(job, t1_ms, with_exchange=True):
    """What ONE rank of an N-GPU job computes per step, measured on this GPU: its strip share (strips / N) for T ticks (`fixed_ticks`) and for T x N
    ticks (`scale_ticks`, the policy bench.py --gpus N reports as `value`: the chunk length per lane of the speculative EqThree stays what it is at
    N = 1).  The exchange's wire time is modelled as bytes received per rank and step / 300 GB/s of xGMI and assumed hidden behind the next step's
    compute when shorter (mx_exchange runs on its own stream)."""
    torch, abi, args = job.torch, job.abi, job.args
    strips, T, spt = args.strips, job.T, job.spt
    flags = job.flags & ~abi.FLAG_OVERLAP_TAIL
    out = {"fixed_ticks": {}, "scale_ticks": {}}
    for policy, mult in (("fixed_ticks", lambda n: 1), ("scale_ticks", lambda n: n)):
        for n in (2, 4, 8):
            if strips % n:
                continue
            Tn, sn = T * mult(n), strips // n
            ws, mix, srcs, trigs = build_strips(abi, job.Workspace, job.synth, sn, 0, job.SR, want_trigs=True)
            g = job.build(ws=ws, T=Tn, flags=flags)
            gen = torch.Generator(device="cuda"); gen.manual_seed(0x4D58 + n)
            noise = (torch.rand(Tn * spt, generator=gen, device="cuda", dtype=torch.float32) * 2.0 - 1.0).contiguous()
            for s_ in srcs:
                g.bind_source_device(s_, noise.data_ptr())             # every strip of the probe reads the same device-resident noise
            k = 3
            evs = [gate_events(abi, trigs, 0, i * Tn, Tn) if job.toggling else None for i in range(2 + k)]
            for i in range(2 + k):
                if i == 2:
                    g.sync(); t0 = time.perf_counter()
                if evs[i] is not None:
                    g.schedule_params_batch(evs[i][0], evs[i][1])
                g.run_ticks(i * Tn, Tn)
            g.sync()
            ms = (time.perf_counter() - t0) / k * 1e3
            # the same rank with an exchange in the loop (fixed T only): a ONE-rank RCCL communicator -- the pack, the library's RCCL call and the combine
            # graph really run (behind the held-back Mixer bank); what no single GPU can show is the wire
            ms_x = None
            if policy == "fixed_ticks" and with_exchange:
                from mixlab_amd.exchange import BusExchange, unique_id
                ex1 = BusExchange(g, mix, Tn, 0, 1, mode="allgather", nccl_id=unique_id())
                kx = 4
                evx = [gate_events(abi, trigs, 0, (2 + k + i) * Tn, Tn) if job.toggling else None for i in range(2 + kx)]
                for i in range(2 + kx):
                    if i == 2:
                        g.sync(); ex1.sync(); tx0 = time.perf_counter()
                    if evx[i] is not None:
                        g.schedule_params_batch(evx[i][0], evx[i][1])
                    g.run_ticks((2 + k + i) * Tn, Tn)
                    ex1.submit(i)
                g.sync(); ex1.sync()
                ms_x = (time.perf_counter() - tx0) / kx * 1e3
                ex1.close()
            g.close(); del noise
            bus = 2 * 2 * spt * Tn * 4                                   # Master + Cue, interleaved stereo f32, per step
            recv = 2 * (n - 1) * bus // n if (n >= 4 and Tn % n == 0) else (n - 1) * bus
            ex_ms = recv / 300e9 * 1e3
            step_ms = max(ms, ex_ms)
            out[policy][str(n)] = {"strips_per_rank": sn, "ticks_per_step": Tn, "rank_compute_ms_per_step": round(ms, 4),
                                   **({"rank_step_ms_with_a_1_rank_rccl_exchange_in_the_loop": round(ms_x, 4)} if ms_x is not None else {}),
                                   "exchange_bytes_received_per_rank": recv, "exchange_ms_at_300GBps": round(ex_ms, 4),
                                   "predicted_job_value": strips * Tn / (step_ms * 1e-3),
                                   "predicted_speedup_vs_1_gpu": round((strips * Tn / step_ms) / (strips * T / t1_ms), 2)}
    out["what"] = ("one GPU playing one rank: rank_compute_ms is measured here, the exchange is modelled (bytes / 300 GB/s, hidden when shorter than the compute); "
                   "N > 1 is a model until a multi-GPU node runs the job")
    out["one_gpu_ms_per_step"] = round(t1_ms, 4)
    return out
