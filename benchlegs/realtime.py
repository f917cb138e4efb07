"""The short-submission regimes: one 60 Hz tick per submission synchronised every tick like a live engine, submissions of
64 / 256 / 1024 ticks, and the north-star's real-time statement (10 240 strips + the 8-layer video cascade in ONE graph, one tick per submission)."""
from __future__ import annotations

import time

from .common import HBM_PEAK_GBS, VIDEO_SIZES, build_strips, video_cascade


def realtime_leg(job):
    """One tick per submission on the headline graph itself, submit + wait per tick."""
    g = job.g
    base_t = (job.nxt + 16) * job.T
    for i in range(20):
        g.run_ticks(base_t + i, 1)
    g.sync()
    n_rt = 300
    t0 = time.perf_counter()
    for i in range(n_rt):
        g.run_ticks(base_t + 20 + i, 1)
        g.sync()
    tick_us = (time.perf_counter() - t0) / n_rt * 1e6
    return {"ticks_per_submission": 1, "tick_us": round(tick_us, 1), "tick_budget_us": round(1e6 / 60.0, 1), "headroom": round(1e6 / 60.0 / tick_us, 1),
            "note": "submit + wait per tick (host-paired), same graph as the headline, exact EqThree"}


def t_sweep_leg(job):
    """Shorter submissions of the same strips, gates toggling as in the headline.  Each length runs on a graph BUILT for it (max_ticks_per_run = T, as
    a host that submits T ticks at a time builds it) over the headline graph's resident sources: what the library decides from the submission length
    -- the chunk plan, and the Mixer bank beside the next submission's EqThree group (MX_OVERLAP_AUTO) -- is then what is measured."""
    abi = job.abi
    out = {}
    tick0 = (job.nxt + 64) * job.T

    def sweep(Ts, n_sub, auto):
        nonlocal tick0
        gs = job.build(T=Ts, flags=job.flags & ~abi.FLAG_OVERLAP_TAIL, auto_overlap=auto)
        job.bind_resident_sources(gs)
        from .common import gate_events
        evs = [gate_events(abi, job.trigs, job.first, tick0 + i * Ts, Ts) if job.toggling else None for i in range(n_sub + 3)]

        def sub(i):
            if evs[i] is not None:
                gs.schedule_params_batch(evs[i][0], evs[i][1])
            gs.run_ticks(tick0 + i * Ts, Ts)
        sub(0)
        gs.sync()
        th = time.perf_counter()
        for i in range(1, 3):
            sub(i)
        host_free_s = (time.perf_counter() - th) / 2     # two submissions into an idle queue: what the host needs when nothing makes it wait
        gs.sync()
        t0 = time.perf_counter()
        for i in range(3, n_sub + 3):
            sub(i)
        host_s = time.perf_counter() - t0                # the host's share: scheduling + enqueueing, before the device is waited for
        gs.sync()
        dts = time.perf_counter() - t0
        rec = {"ms_per_step": round(dts / n_sub * 1e3, 4), "value": job.args.strips * Ts * n_sub / dts, "unit": "channel-ticks/s", "submissions": n_sub,
               "host_ms_per_step": round(host_s / n_sub * 1e3, 4), "host_ms_per_step_idle_queue": round(host_free_s * 1e3, 4),
               "mixer_beside_next_eq_three": gs.tail_stream() is not None}
        tick0 += (n_sub + 3) * Ts
        gs.close()
        return rec

    for Ts in (64, 256, 1024):
        if Ts >= job.T:
            continue
        # (enough submissions for the steady state: the first few dozen of a new graph are slower -- first-use allocations, the clock settling)
        out[str(Ts)] = sweep(Ts, {64: 480, 256: 160}.get(Ts, 40), True)
    if "64" in out and out["64"]["mixer_beside_next_eq_three"]:
        out["64_one_stream"] = dict(sweep(64, 480, False), note="MX_OVERLAP_AUTO=0: the same submissions with every launch group on one stream")
    return out


def north_star_leg(job, n_strips=10240):
    """10 240 stereo channel strips mixed + the config-4 video cascade as ONE graph, one tick per submission, synchronised every tick.  Two mix
    topologies the reference can express: one flat Mixer(10 240) -- a single ordered chain per output sample, the strictest reading -- and ten group
    buses Mixer(1024) into a Mixer(10) master, how a desk of that size is wired.  Reports the tick time against the 16 667 us budget."""
    from mixlab_amd import video

    abi, Workspace, synth, sample_rate = job.abi, job.Workspace, job.synth, 48000
    frames_host = [synth.yuv_pattern(w, h, k, seed=3) for k, (w, h) in enumerate(VIDEO_SIZES)]
    blk = [synth.noise(k, sample_rate // 60) for k in range(64)]
    F, F720 = 1920 * 1080 * 3 // 2, 1280 * 720 * 3 // 2

    def one(topology):
        t_build = time.perf_counter()
        extra_bytes = 0
        if topology == "flat":
            ws, mix, srcs = build_strips(abi, Workspace, synth, n_strips, 0, sample_rate)
        else:
            # strips k*1024 .. k*1024+1023 into group bus k (same gains / faders / cue flags as the flat job), buses into a unity master
            n_bus = n_strips // 1024
            ws, srcs, buses = Workspace(sample_rate, 60), [], []
            for b in range(n_bus):
                _ws, bus, s_b = build_strips(abi, Workspace, synth, 1024, b * 1024, sample_rate, ws=ws, total=n_strips)
                buses.append(bus); srcs += s_b
            master = ws.mixer([(0.0, 1.0, False)] * n_bus)
            for b, bus in enumerate(buses):
                ws.connect(bus, 0, master, b)
            spt_ = sample_rate // 60
            extra_bytes = n_bus * 2 * 8 * spt_ + (n_bus + 2) * 8 * spt_     # the buses' outputs + the master Mixer(n_bus)
        vsrcs, _rgba = video_cascade(ws)
        g = ws.build(max_ticks_per_run=1, device=job.local_rank, stream=job.stream.cuda_stream)
        for j, s in enumerate(srcs):
            g.write_source(s, blk[j % 64], 1)
        keep = []
        for k, (w, h) in enumerate(VIDEO_SIZES):
            y, u, v = frames_host[k]
            d = video.DFrame(w, h).upload(y, u, v)
            keep.append(d)
            video.graph_set_video_source(g, vsrcs[k], d, dur=(1, 60), off=(0, 1), repeat=True)
        t_build = time.perf_counter() - t_build
        for i in range(20):
            g.run_ticks(i, 1)
        g.sync()
        n = 200
        t0 = time.perf_counter()
        for i in range(n):
            g.run_ticks(20 + i, 1)
            g.sync()
        tick_us = (time.perf_counter() - t0) / n * 1e6
        by_kind, _tot = g.profile_run(20 + n, 1)
        g.close()
        # module-boundary bytes of one tick: strips 51 200 B each (incl. their mixer input), the video cascade
        tick_bytes = 51200 * (sample_rate / 48000.0) * n_strips + extra_bytes + 7 * 3 * F + 2 * (F720 + F) + (F + 1920 * 1080 * 4)
        return {"tick_us": round(tick_us, 1), "headroom": round(1e6 / 60.0 / tick_us, 1),
                "device_ms_by_kind": {k: round(v, 4) for k, v in sorted(by_kind.items()) if v > 0},
                "hbm_frac_module_boundary_bytes": round(tick_bytes / (tick_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "graph_nodes": len(ws.nodes), "graph_build_s": round(t_build, 2)}

    out = {"workload": f"{n_strips} channel strips mixed + 8-layer 1080p cascade -> RGBA, one 1/60 s tick per submission, synchronised every tick",
           "tick_budget_us": round(1e6 / 60.0, 1)}
    out.update(one("flat"))                         # headline fields: the flat Mixer(10 240)
    out["mix_topology"] = f"flat Mixer({n_strips})"
    out["group_buses"] = dict(one("buses"), mix_topology=f"{n_strips // 1024} x Mixer(1024) -> Mixer({n_strips // 1024}, unity)")
    return out
