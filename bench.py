#!/usr/bin/env python
"""bench.py -- throughput of the per-tick module-graph hot path on MI355X.

Workload (BASELINE.json configs[1]; SURVEY.md section 8d config 2 AS WRITTEN): 1024 channel strips
  Trigger -> Envelope ;  Source(noise) -> EqThree -> StereoPanner(L=R) -> Amplifier(ctl = Envelope) -> Mixer(1024)
at 48 kHz (SPT = 800), every strip's gate toggling every 30 ticks with phase k mod 60 -- applied BETWEEN ticks of the
batch through mx_graph_schedule_params_batch (the reference's client_update between two ticks, src/engine.rs:192-214) --
T ticks batched per submission ("step" = one pass of the whole graph over T ticks of synthetic input already resident in
HBM).  EqThree runs in the reference's exact order (the library default).  Metric: audio channels mixed per second =
strip-ticks (one stereo strip processed and mixed for one 1/60 s tick) per second, whole job.

N > 1 (BASELINE.json configs[4]): the 1024 strips are sharded contiguously over the ranks (strong
scaling), each rank runs Mixer(1024/N) over its strips, and the partial Master / Cue buses are combined by
the library's own exchange (mx_exchange_*, RCCL called from libmixlab_gpu.so: rank-ordered sum = the reference-expressible
graph N x Mixer(1024/N) -> Mixer(N); --exchange allreduce is the north-star's non-parity collective).

One JSON line on rank 0; see the task contract for the fields.  `roofline` describes the launch group that took the most
device time in the timed region (hipEvents on the graph's stream, recorded inside the timed region); `roofline.per_kernel`
lists MOVED-byte fractions for every kernel family; `repeats` shows the spread of further repetitions of the same K steps;
`held_gates` is the same job with every gate held (the round-1 configuration); `cpu_baseline` is the CPU oracle (a C port of
the reference algorithms, one thread like the reference's engine thread, built on this host) on a bounded sample.
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

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)

# algorithmic (module-boundary) bytes per instance per frame: every input port read once + every output port written once
# (SURVEY.md section 8d); mixer is per input channel, +16 / frame for its two outputs.  Used only with --no-fuse, where every
# port really is materialised.
BYTES_PER_FRAME = {"trigger": 4, "envelope": 8, "eq_three": 8, "stereo_panner": 16, "amplifier": 20, "mixer": 8}
# default (graph-compiler fusion): Trigger + Envelope + EqThree + StereoPanner + Amplifier are ONE kernel that reads the source
# (4 B / frame) and writes the strip as one float per frame (L == R): 8 B / frame = SURVEY 8d's 2M per EqThree channel-tick;
# the Mixer reads those 4 B.  These are bytes that move.
BYTES_PER_FRAME_FUSED = {"eq_three": 4 + 4, "mixer": 4}
F64_VALU_PEAK_TOPS = 39.3   # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz f64 instructions / s (an FMA counts once)


def gate_open(tick, k):
    """the Trigger of strip k toggles every 30 ticks with phase k mod 60."""
    return ((tick + k) // 30) % 2 == 1


def gate_events(abi, trigs, first_strip, t0, n_ticks):
    """The toggles of every strip's Trigger that fall on ticks [t0, t0 + n_ticks) as one mx_param_event array for
    mx_graph_schedule_params_batch (tick_in_run 0 = the boundary before the submission's first tick: a strip whose toggle falls exactly
    on t0 gets it there).  Returns (ctypes pointer, count, keep-alive tuple) or None.  Built with numpy: ~70 000 events per 2048-tick step."""
    import ctypes as C
    p_open, p_closed = abi.TriggerParams(1), abi.TriggerParams(0)
    po, pc = C.addressof(p_open), C.addressof(p_closed)
    k = first_strip + np.arange(len(trigs), dtype=np.int64)
    first = (30 - (t0 + k) % 30) % 30                             # first toggle at or after t0, per strip
    n_ev = np.maximum(0, (n_ticks - first + 29) // 30)            # toggles at first, first + 30, ... < n_ticks
    total = int(n_ev.sum())
    if total == 0:
        return None
    strip = np.repeat(np.arange(len(trigs)), n_ev)
    j = np.arange(total) - np.repeat(np.cumsum(n_ev) - n_ev, n_ev)
    tick = first[strip] + 30 * j
    opens = ((t0 + tick + k[strip]) // 30) % 2 == 1
    ev = np.zeros(total, dtype=np.dtype([("node", "<u4"), ("tick_in_run", "<u4"), ("params", "<u8"), ("params_len", "<u8")], align=True))
    assert ev.dtype.itemsize == C.sizeof(abi.ParamEvent)
    ev["node"] = np.asarray(trigs, dtype=np.uint32)[strip]; ev["tick_in_run"] = tick
    ev["params"] = np.where(opens, po, pc); ev["params_len"] = C.sizeof(abi.TriggerParams)
    return ev.ctypes.data_as(C.POINTER(abi.ParamEvent)), total, (ev, p_open, p_closed)


def build_strips(abi, Workspace, synth, n_strips, first_strip, sample_rate, ws=None, total=None, want_trigs=False):
    """Config-2 strips [first_strip, first_strip + n_strips) with the global seeded parameters, into a Mixer(n_strips);
    `ws`: add them to an existing workspace (group buses), `total`: size of the whole job the parameters are drawn for."""
    if total is None:
        total = 1024 if first_strip + n_strips <= 1024 else first_strip + n_strips
    eq_g = synth.uniform(10, 3 * total, -24.0, 6.0)
    mg = synth.uniform(11, total, -24.0, 6.0)
    mf = synth.uniform(12, total, 0.0, 1.0)
    if ws is None:
        ws = Workspace(sample_rate, 60)
    mix = ws.mixer([(float(mg[k]), float(mf[k]), k % 8 == 0) for k in range(first_strip, first_strip + n_strips)])
    srcs, trigs = [], []
    for j, k in enumerate(range(first_strip, first_strip + n_strips)):
        trig = ws.trigger(gate_open(0, k))          # gate at tick 0; toggles every 30 ticks with phase k mod 60 (gate_events)
        trigs.append(trig)
        env = ws.envelope()                         # defaults 25/500/0.8/200 (protocol/src/lib.rs:318-327)
        src = ws.source_mono()
        eq = ws.eq_three(float(eq_g[3 * k]), float(eq_g[3 * k + 1]), float(eq_g[3 * k + 2]))
        pan = ws.stereo_panner()
        amp = ws.amplifier(1.0, 0.5)
        ws.connect(trig, 0, env, 0)
        ws.connect(src, 0, eq, 0)
        ws.connect(eq, 0, pan, 0); ws.connect(eq, 0, pan, 1)
        ws.connect(pan, 0, amp, 0); ws.connect(env, 0, amp, 1)
        ws.connect(amp, 0, mix, j)
        srcs.append(src)
    if want_trigs:
        return ws, mix, srcs, trigs
    return ws, mix, srcs


def headline_parity(g, abi, Workspace, synth, args, sample_rate, T, n_steps_run, first, local_strips, mix, toggling, contract, src_of):
    """The timed submissions' outputs against the CPU oracle at the job's own shape (tests/headline_replay.py): a sample of strips replayed from
    tick 0 and compared bit for bit with the last submission's fused strip outputs; Master / Cue of sampled ticks against the oracle Mixer over the
    device's own strips.  Runs AFTER a timed region, outside every clock.  `src_of(j)`: the T-tick source buffer of local strip j as uploaded."""
    import headline_replay as hr    # test infrastructure: the checker

    total = max(1024, args.strips)
    ids = hr.sample_strips(local_strips, args.parity_strips)
    mg = synth.uniform(11, total, -24.0, 6.0)
    mf = synth.uniform(12, total, 0.0, 1.0)

    def one(k):
        ws1, mix1, srcs1, trigs1 = build_strips(abi, Workspace, synth, 1, k, sample_rate, total=total, want_trigs=True)
        return ws1, (mix1, srcs1[0], trigs1[0], mix1 + 6)

    t0 = time.perf_counter()
    rec = hr.replay_and_compare(g, one, ids, first, {j: src_of(j) for j in ids}, T, n_steps_run, mix, lambda j: mix + 6 * j + 6, toggling=toggling,
                                contract=contract, check_ticks=6, all_amp_nodes=[mix + 6 * j + 6 for j in range(local_strips)],
                                mixer_channels=[(float(mg[k]), float(mf[k]), k % 8 == 0) for k in range(first, first + local_strips)])
    rec["shape"] = f"{local_strips} strips x {T} ticks per submission @ {sample_rate} Hz, submission {n_steps_run - 1} (the last one timed)"
    rec["seconds"] = round(time.perf_counter() - t0, 2)
    return rec


def native_oracle():
    """Build the CPU oracle ON THIS HOST with -O3 -march=native (same sources, same -ffp-contract=off -fno-fast-math: same
    results) for the timed baselines; falls back to the library shipped with the repo.  Must run before `import oracle`."""
    import subprocess
    import tempfile
    out = pathlib.Path(tempfile.gettempdir()) / f"libmixlab_oracle_native_{os.getpid()}.so"
    try:
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "ARCH=native", f"OUT={out}"], check=True, capture_output=True)
        os.environ["MIXLAB_ORACLE_LIB"] = str(out)
        return "gcc -O3 -march=native -ffp-contract=off -fno-fast-math, built on this host"
    except (OSError, subprocess.CalledProcessError):
        return "library shipped with the repo (-O3 -march=x86-64-v3)"


def cpu_baseline(Workspace, synth, abi, n_strips, sample_rate, build_note, target_seconds=12.0):
    """Time the CPU oracle's graph runner (C, one thread) on a bounded sample of the same workload, gates toggling every
    30 ticks (ModuleT::update between ticks, as the reference's client_update does)."""
    import oracle  # test infrastructure: used here only as the timed CPU baseline

    ws, mix, srcs, trigs = build_strips(abi, Workspace, synth, n_strips, 0, sample_rate, want_trigs=True)
    og = oracle.OracleGraph(ws)
    spt = ws.spt
    noise = [synth.noise(k, spt) for k in range(n_strips)]
    for s, nz in zip(srcs, noise):
        og.set_source(s, nz)
    p_open, p_closed = abi.TriggerParams(1), abi.TriggerParams(0)

    def tick(t):
        for k in range(n_strips):                      # the strips whose gate toggles before this tick
            if t and (t + k) % 30 == 0:
                og.update_params(trigs[k], p_open if gate_open(t, k) else p_closed)
        og.run_tick(t)

    t0 = time.perf_counter()
    for t in range(4):
        tick(t)
    per_tick = (time.perf_counter() - t0) / 4
    n_ticks = int(max(8, min(4000, target_seconds / max(per_tick, 1e-6))))
    t0 = time.perf_counter()
    for t in range(4, 4 + n_ticks):
        tick(t)
    dt = time.perf_counter() - t0
    return {
        "value": n_strips * n_ticks / dt, "unit": "channel-ticks/s", "cores": 1, "kind": "port", "build": build_note,
        "sample": f"{n_strips} strips x {n_ticks} ticks @ {sample_rate} Hz, gates toggling every 30 ticks, single thread (the reference engine is one thread, src/engine.rs:78), {dt:.1f} s",
        "cpu_model": _cpu_model(), "host_cores": os.cpu_count(),
    }


def cpu_baseline_all_cores(Workspace, synth, abi, shard, n_strips, sample_rate, per_strip_tick_s, target_seconds=6.0):
    """The same CPU oracle, one graph shard per host core (SURVEY.md section 8d "(ii)"): the strips are partitioned like
    the multi-GPU job (contiguous shards, each with its own sub-Mixer); ctypes releases the GIL, so plain threads run
    the C runners concurrently.  The final Mixer(shards) over the partial buses is not included (negligible)."""
    import threading

    import oracle  # test infrastructure: used here only as the timed CPU baseline

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = None
    try:   # a container may see every host CPU and still be throttled to a few cores' worth of time
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:   # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        cores = max(1, min(cores, int(quota + 0.5)))
    n_thr = max(1, min(cores, n_strips))
    shards = []
    for r in range(n_thr):
        first, cnt = shard.strip_range(r, n_thr, n_strips)
        ws, _mix, srcs = build_strips(abi, Workspace, synth, cnt, first, sample_rate)
        og = oracle.OracleGraph(ws)
        for j, sn in enumerate(srcs):
            og.set_source(sn, synth.noise(first + j, ws.spt))
        shards.append(og)

    def timed(n_ticks):
        go = threading.Barrier(n_thr + 1)

        def work(og):
            go.wait()
            og.run_ticks(0, n_ticks)     # one foreign call per thread: the GIL is released for its whole duration

        th = [threading.Thread(target=work, args=(og,)) for og in shards]
        for t in th:
            t.start()
        go.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    probe = 16
    dt_probe = timed(probe)                                   # calibrate under the real contention, then run the bounded sample
    n_ticks = int(max(probe, min(400000, target_seconds / max(dt_probe / probe, 1e-7))))
    dt = timed(n_ticks)
    return {"value": n_strips * n_ticks / dt, "unit": "channel-ticks/s", "cores": n_thr, "kind": "port",
            "cpu_quota_cores": quota, "host_logical_cpus": os.cpu_count(),
            "sample": f"{n_strips} strips in {n_thr} contiguous shards (one thread each) x {n_ticks} ticks @ {sample_rate} Hz, gates held, {dt:.1f} s"}


VIDEO_FADERS = [1.0, 0.75, 0.5, 0.5, 0.25, 0.9, 0.1]
VIDEO_MATRIX = [3900, 150, 46, 4096, 60, 3980, 56, -2048, 20, 120, 3956, 0]


def video_leg(torch, dist, world, stream, local_rank, frames, warmup, n_sets=16, shard_mode="replicas", rank=0, band_as=None, only=None):
    """BASELINE.json configs[3]: 8 layers (6 x 1080p + 2 x 720p) every tick ->
    cascade of 7 reference VideoMixer cross-fades (scale + letterbox for the 720p layers) ->
    build-specified YUV420P->RGBA + colour matrix.  One composited 1080p RGBA frame per tick.
    Every source delivers a NEW frame each tick out of a ring of `n_sets` distinct frames (16 sets x 21.4 MB = 342 MB > the
    256 MiB Infinity Cache), so the layers come from HBM, not from cache.
    N > 1: every rank composites its own independent 8-layer stream (independent VideoMixer
    instances) -- no exchange step, weak scaling.
    --video-shard bands: ONE picture stream over all ranks (strong scaling): rank r composites row band r of every frame
    (mixlab_amd/shard.py: whole chroma rows; the 720p layers arrive as the halo slice the band's vertical taps reach and are scaled to
    the band by their source nodes, mx_graph_set_video_source_band); no exchange step either -- each band goes to its own sink.
    N = 1 adds two variants of the same job beside the headline one: `no_rest_fader` (every fader inside its travel: all eight layers are read) and
    `alpha` (three layers carry a coverage plane -- BASELINE's "alpha composite", build-specified: DESIGN.md "Per-pixel alpha")."""
    import alpha_patterns   # seeded coverage planes (numpy only)
    import synth   # seeded synthetic patterns (numpy only)
    from mixlab_amd import shard, video
    from mixlab_amd.workspace import Workspace

    sizes = [(1920, 1080)] * 6 + [(1280, 720)] * 2
    bands = shard_mode == "bands"
    row0, rows = shard.row_bands(1080, world)[rank] if bands else (0, 1080)
    if band_as:                                                        # one GPU plays rank R of W (what a rank of the sharded job costs)
        bands = True
        rank, of = band_as
        row0, rows = shard.row_bands(1080, of)[rank]
    ALPHA_LAYERS = (2, 5, 7)                                           # two 1080p layers and a scaled 720p one carry coverage in the `alpha` variant
    variants = not bands and world == 1 and only != "main"       # only: "main" = the headline job alone; "alpha" / "no_rest_fader" = that variant AS the measured job (tools/vleg.py)
    T = 256   # ticks per submission (a throughput knob like the audio leg's: the video pipeline fills and drains once per run)
    cuts, rings, alpha_rings = [], [], {}
    for k, (w, h) in enumerate(sizes):
        ring = []
        cut = (0, h)                                                   # the luma rows of this layer the rank holds
        if bands:
            cut = (row0, rows) if (w, h) == (1920, 1080) else shard.band_source_rows((row0, rows), w, h, 1920, 1080)
        cuts.append(cut)
        for r in range(n_sets):
            y, u, v = synth.yuv_pattern(w, h, k, seed=3 + r)
            if bands:
                y, u, v = y[cut[0]:cut[0] + cut[1]], u[cut[0] // 2:(cut[0] + cut[1]) // 2], v[cut[0] // 2:(cut[0] + cut[1]) // 2]
            ring.append(video.DFrame(w, cut[1]).upload(y, u, v))
            if variants and k in ALPHA_LAYERS:                         # the same picture once more as yuva420p with a seeded coverage plane
                pat = ("soft-disc", "random")[(k + r) % 2]
                alpha_rings.setdefault(k, []).append(video.DFrame(w, h, fmt=video.PIXFMT_YUVA420P).upload(y, u, v).upload_alpha(alpha_patterns.alpha_plane(w, h, pat, r)))
        rings.append(ring)

    def run(faders, alpha_layers, n_frames_wanted, warm):
        ws = Workspace(48000, 60)
        srcs = [ws.source_video() for _ in sizes]
        prev = srcs[0]
        for k in range(1, 8):
            m = ws.video_mixer(a=0, b=1, fader=faders[k - 1])
            ws.connect(prev, 0, m, 0); ws.connect(srcs[k], 0, m, 1)
            prev = m
        rgba = ws.video_to_rgba(VIDEO_MATRIX)
        ws.connect(prev, 0, rgba, 0)
        g = ws.build(max_ticks_per_run=T, device=local_rank, stream=stream.cuda_stream)
        for k, (w, h) in enumerate(sizes):
            if bands and (w, h) != (1920, 1080):
                video.graph_set_video_source_band(g, srcs[k], w, h, cuts[k][0], cuts[k][1], 1920, 1080, row0, rows)
            video.graph_set_video_source_ring(g, srcs[k], alpha_rings[k] if k in alpha_layers else rings[k], dur=(1, 60), off=(0, 1))
        steps = max(1, n_frames_wanted // T)
        for i in range(max(1, warm)):
            g.run_ticks(i * T, T)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        g.profile_enable(True)
        t0 = time.perf_counter()
        for i in range(steps):
            g.run_ticks((warm + i) * T, T)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        g.profile_enable(False)
        by_kind, _tot, n_prof = g.profile_collect()
        g.close()
        return dt, steps, by_kind.get("video_mixer", 0.0) / max(1, n_prof) / T    # wall seconds, steps, device ms per composited frame (scaler + chain)

    if only == "alpha":
        dt, steps, dev_ms = run(VIDEO_FADERS, ALPHA_LAYERS, frames, warmup)
    elif only == "no_rest_fader":
        dt, steps, dev_ms = run([0.95 if f == 1.0 else f for f in VIDEO_FADERS], (), frames, warmup)
    else:
        dt, steps, dev_ms = run(VIDEO_FADERS, (), frames, warmup)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    F = 1920 * 1080 * 3 // 2
    F720 = 1280 * 720 * 3 // 2
    RGBA = 1920 * 1080 * 4
    # bytes per composited frame.  Module-boundary accounting (every VideoMixer output materialised):
    # 7 cross-fades x 3F + 2 scales (F720 in + F out) + RGBA (F in + 4wh out).  MOVED by the two fused kernels: the batched scaler
    # reads 2 x F720 and writes 2 x F; the chain kernel reads 8 F (six layers + the two scaled ones) and writes the RGBA frame.
    # FUSED MINIMUM (SURVEY 8d asks for it beside whichever figure is used): every layer read once at its own size + the RGBA frame written.
    alg = 7 * 3 * F + 2 * (F720 + F) + (F + RGBA)
    moved_scaler = 2 * (F720 + F)
    fused_min = 6 * F + 2 * F720 + RGBA

    def moved_chain_of(faders):
        # a step whose fader rests at an end of its travel returns one of its inputs exactly: the launcher drops it and never reads the other
        # layer (mx_k_video.hip chain_matrix_mode).  SURVEY's config-4 faders start with 1.0, so 7 of the 8 layers are read.
        layers_read = 8 - sum(1 for f in faders if f == 1.0)
        return layers_read, layers_read * F + RGBA

    layers_read, moved_chain = moved_chain_of(VIDEO_FADERS)
    n_frames = steps * T * (1 if bands else world)
    if bands:
        return {"metric": "1080p_composited_fps", "value": n_frames / dt, "unit": "frames/s", "scaling": "strong",
                "shard": f"row bands: rank {rank} of {band_as[1] if band_as else world} composites luma rows [{row0}, {row0 + rows}) of every frame; the 720p layers enter as halo slices and are scaled to the band per tick (two-pass kernel)",
                "workload": "8 layers (6x1080p + 2x720p yuv420p) -> 7 VideoMixer cross-fades (+2 bicubic letterbox scales) -> YUV->RGBA + 3x4 matrix, ONE stream over all ranks",
                "frames": n_frames, "device_us_per_frame_rank0": round(dev_ms * 1e3, 2),
                "note": "a 1080p frame is ~15 us of device work on one GPU: cut 8 ways a band is launch-sized (~4.5 us), so this mode pays for pictures far larger than 1080p"}

    def fracs(moved, ms):
        return round(moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None

    out = {
        "metric": "1080p_composited_fps", "value": n_frames / dt, "unit": "frames/s", "scaling": "weak",
        "workload": "8 layers (6x1080p + 2x720p yuv420p) -> 7 VideoMixer cross-fades (+2 bicubic letterbox scales) -> YUV->RGBA + 3x4 matrix",
        "inputs": f"a new frame per layer per tick out of rings of {n_sets} distinct device frames ({n_sets * (6 * F + 2 * F720) / 1e6:.0f} MB in all: HBM-resident, beyond the 256 MiB Infinity Cache)",
        "frames": n_frames, "realtime_1080p60_streams_equiv": n_frames / dt / 60.0,
        "device_us_per_frame": round(dev_ms * 1e3, 2),
        "moved_bytes_per_frame": moved_scaler + moved_chain, "module_boundary_bytes_per_frame": alg, "fused_minimum_bytes_per_frame": fused_min,
        "hbm_frac_moved_bytes_device": fracs(moved_scaler + moved_chain, dev_ms),
        "hbm_frac_fused_min": fracs(fused_min, dev_ms),
        "hbm_frac_moved_bytes_wall": round((moved_scaler + moved_chain) * n_frames / world / dt / 1e9 / HBM_PEAK_GBS, 4),
        "per_kernel_moved_bytes": {"scaler tiles (2 layers)": moved_scaler, "chain tiles": moved_chain,
                                   "layers_read_by_the_chain": layers_read,
                                   "launches": "k_video_batch: the chains of 16 ticks and the scaler tiles of the 16 ticks after them in ONE launch (MX_VIDEO_BATCH; DESIGN.md 5.3)",
                                   "ticks_per_submission": T,
                                   "note": "per-kernel durations and PMC traffic: profiles/r05 (the hipEvents here bracket the whole per-tick video section); fused_minimum = every layer read once at its own size + the RGBA frame written: what the round trip of the two scaled layers costs on top is moved - fused_minimum"},
    }
    if variants and only is None:
        nf = min(frames, 1024)
        # every fader inside its travel: nothing is pruned, the chain reads all eight layers
        f2 = [0.95 if f == 1.0 else f for f in VIDEO_FADERS]
        dt2, st2, ms2 = run(f2, (), nf, 1)
        lr2, mc2 = moved_chain_of(f2)
        out["no_rest_fader"] = {"faders": f2, "value": st2 * T / dt2, "unit": "frames/s", "device_us_per_frame": round(ms2 * 1e3, 2), "layers_read_by_the_chain": lr2,
                                "moved_bytes_per_frame": moved_scaler + mc2, "hbm_frac_moved_bytes_device": fracs(moved_scaler + mc2, ms2), "hbm_frac_fused_min": fracs(fused_min, ms2)}
        # BASELINE configs[3]'s "alpha composite" (build-specified): layers 2, 5 (1080p) and 7 (720p, scaled with its coverage) carry a coverage plane
        dt3, st3, ms3 = run(VIDEO_FADERS, ALPHA_LAYERS, nf, 1)
        a1080, a720 = 1920 * 1080, 1280 * 720
        moved_alpha = 2 * a1080 + (a720 + a1080) + a1080          # two planes read by the chain; the 720p plane read + written by the scaler, then read by the chain
        out["alpha"] = {"layers_with_coverage": list(ALPHA_LAYERS), "value": st3 * T / dt3, "unit": "frames/s", "device_us_per_frame": round(ms3 * 1e3, 2),
                        "moved_bytes_per_frame": moved_scaler + moved_chain + moved_alpha, "hbm_frac_moved_bytes_device": fracs(moved_scaler + moved_chain + moved_alpha, ms3),
                        "cost_vs_headline_us": round((ms3 - dev_ms) * 1e3, 2),
                        "parity": "bit-exact vs the oracle's rule (tests/test_gpu_video_alpha.py); constant-255 coverage reproduces the headline picture bit for bit",
                        "what": "wa = aA*fade/255, wb = aB*(255-wa)/255, out = (A*(255-wb) + B*wb)/255 per sample in fade_line's u16 arithmetic (DESIGN.md 'Per-pixel alpha')"}
        # the experiment VERDICT r4 asked for: the Q12 colour matrix on the matrix cores (v_mfma_i32_4x4x4_16b_i8, bit-exact) instead of packed f32 FMAs
        os.environ["MX_VIDEO_MFMA_MATRIX"] = "1"
        try:
            dt4, st4, ms4 = run(VIDEO_FADERS, (), nf, 1)
        finally:
            os.environ.pop("MX_VIDEO_MFMA_MATRIX", None)
        out["mfma_matrix"] = {"env": "MX_VIDEO_MFMA_MATRIX=1", "value": st4 * T / dt4, "unit": "frames/s", "device_us_per_frame": round(ms4 * 1e3, 2),
                              "vs_headline_us": round((ms4 - dev_ms) * 1e3, 2), "parity": "bit-exact (integer; tests/test_gpu_video_graph.py)",
                              "counters": "profiles/r05/video_sq_mfma_{0,1}.txt (SQ_INSTS_VALU per launch of 16 frames)"}
    return out


def north_star_realtime_leg(torch, stream, local_rank, abi, Workspace, synth, n_strips=10240, sample_rate=48000):
    """The north-star's real-time statement as ONE graph, one tick per submission: 10 240 stereo channel strips (config-2 strips)
    mixed, plus the config-4 video cascade (8 layers -> 7 VideoMixers -> RGBA), every tick synchronised like a live engine.
    Two mix topologies the reference can express: one flat Mixer(10 240) -- a single ordered chain per output sample, the
    strictest reading -- and ten group buses Mixer(1024) into a Mixer(10) master, how a desk of that size is wired.
    Reports the tick time against the 16 667 us budget."""
    from mixlab_amd import video

    sizes = [(1920, 1080)] * 6 + [(1280, 720)] * 2
    frames_host = [synth.yuv_pattern(w, h, k, seed=3) for k, (w, h) in enumerate(sizes)]
    blk = None
    F, F720 = 1920 * 1080 * 3 // 2, 1280 * 720 * 3 // 2

    def one(topology):
        nonlocal blk
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
        vsrcs = [ws.source_video() for _ in sizes]
        prev = vsrcs[0]
        for k in range(1, 8):
            m = ws.video_mixer(a=0, b=1, fader=VIDEO_FADERS[k - 1])
            ws.connect(prev, 0, m, 0); ws.connect(vsrcs[k], 0, m, 1)
            prev = m
        rgba = ws.video_to_rgba(VIDEO_MATRIX)
        ws.connect(prev, 0, rgba, 0)
        g = ws.build(max_ticks_per_run=1, device=local_rank, stream=stream.cuda_stream)
        spt = ws.spt
        if blk is None:
            blk = [synth.noise(k, spt) for k in range(64)]
        for j, s in enumerate(srcs):
            g.write_source(s, blk[j % 64], 1)
        keep = []
        for k, (w, h) in enumerate(sizes):
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
        # module-boundary bytes of one tick: strips 51 200 B each (incl. their mixer input), the video cascade
        tick_bytes = 51200 * (sample_rate / 48000.0) * n_strips + extra_bytes + 7 * 3 * F + 2 * (F720 + F) + (F + 1920 * 1080 * 4)
        return {"tick_us": round(tick_us, 1), "headroom": round(1e6 / 60.0 / tick_us, 1),
                "device_ms_by_kind": {k: round(v, 4) for k, v in sorted(by_kind.items()) if v > 0},
                "hbm_frac_module_boundary_bytes": round(tick_bytes / (tick_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "graph_nodes": len(ws.nodes), "graph_build_s": round(t_build, 2)}

    flat = one("flat")
    buses = one("buses")
    out = {"workload": f"{n_strips} channel strips mixed + 8-layer 1080p cascade -> RGBA, one 1/60 s tick per submission, synchronised every tick",
           "tick_budget_us": round(1e6 / 60.0, 1)}
    out.update(flat)                         # headline fields: the flat Mixer(10 240)
    out["mix_topology"] = f"flat Mixer({n_strips})"
    out["group_buses"] = dict(buses, mix_topology=f"{n_strips // 1024} x Mixer(1024) -> Mixer({n_strips // 1024}, unity)")
    return out


def fir_leg(torch, stream, local_rank, T, steps, warmup, flags=0, with_contract=True):
    """BASELINE.json configs[2] (build-specified): 256 stereo channels @44.1 kHz ->
    128-tap FIR reverb -> 160/147 polyphase resampler (16 taps per phase) -> 48 kHz-domain Mixer(256).
    f64 accumulation in ascending tap order with separate multiply and add, one rounding to f32."""
    import synth
    from mixlab_amd.workspace import Workspace

    n_ch, SPT = 256, 735
    up, down, tpp = 160, 147, 16
    n = up * tpp
    m = np.arange(n) - (n - 1) / 2.0
    fc = 0.5 / max(up, down) * 0.92
    table = np.ascontiguousarray((2 * fc * np.sinc(2 * fc * m) * np.kaiser(n, 8.6) * up).reshape(tpp, up).T)
    ws = Workspace(44100, 60)
    srcs, rs = [], []
    for k in range(n_ch):
        taps = (synth.uniform(20 + k, 128, -1.0, 1.0) * np.exp(-np.arange(128) / 24.0) * 0.35).astype(np.float64)
        s = ws.source_stereo(); f = ws.fir(taps); r = ws.resample(up, down, table)
        ws.connect(s, 0, f, 0); ws.connect(f, 0, r, 0)
        srcs.append(s); rs.append(r)
    mix = ws.mixer([(0.0, 1.0, k % 2 == 0) for k in range(n_ch)])
    for k, r in enumerate(rs):
        ws.connect(r, 0, mix, k)
    g = ws.build(max_ticks_per_run=T, flags=flags, device=local_rank, stream=stream.cuda_stream)
    for k, s in enumerate(srcs):
        blk = synth.noise(60 + k, 2 * SPT * min(T, 64))
        g.write_source(s, np.tile(blk, (T + 63) // 64)[: 2 * SPT * T], T)
    for i in range(max(1, warmup)):
        g.run_ticks(i * T, T)
    torch.cuda.synchronize()
    g.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(steps):
        g.run_ticks((warmup + i) * T, T)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g.profile_enable(False)
    by_kind, _tot, n_prof = g.profile_collect()
    frames_in = T * SPT
    # f64 operations the spec prescribes: per output frame 2 channels x taps x (mul + add)
    fir_ops = n_ch * frames_in * 2 * 128 * 2
    rs_ops = n_ch * (T * 800) * 2 * tpp * 2
    k_ms = {k: v / max(1, n_prof) for k, v in by_kind.items() if v > 0}
    out = {"metric": "fir_resample_stereo_ch_ticks_per_sec", "value": n_ch * T * steps / dt, "unit": "channel-ticks/s",
           "workload": "256 stereo channels @44.1 kHz: 128-tap FIR -> 160/147 polyphase resampler (16 taps/phase) -> Mixer(256) @48 kHz",
           "ticks_per_step": T, "ms_per_step": dt / steps * 1e3, "kernel_ms_per_step": {k: round(v, 5) for k, v in sorted(k_ms.items())},
           "realtime_stereo_channels_equiv": n_ch * T * steps / dt / 60.0}
    # per-kernel roofs: the f64 operations the spec prescribes against the f64 VALU rate, the bytes a kernel has to move against HBM, and the
    # HBM traffic of the committed PMC passes (profiles/rNN/fir_pmc_traffic.json) while the kernel sources are the ones it was collected on
    traffic = {}
    try:
        rec = json.load(open(PROFILE_DIR / "fir_pmc_traffic.json"))
        if rec.get("kernel_sources_sha16") == _kernel_hash("fir") and rec.get("config", {}).get("ticks_per_step") == T:
            traffic = rec.get("bytes_per_launch", {})
    except (OSError, ValueError):
        pass
    moved = {"fir": n_ch * frames_in * 8 * 2, "resample": n_ch * (frames_in + T * 800) * 8}
    ops = {"fir": fir_ops, "resample": rs_ops}
    roof = {}
    for k in ("fir", "resample"):
        if k in k_ms:
            sec = k_ms[k] * 1e-3
            roof[k] = {"ms": round(k_ms[k], 5), "f64_ops_per_launch": ops[k], "f64_tops": round(ops[k] / sec / 1e12, 2), "f64_frac": round(ops[k] / sec / 1e12 / F64_VALU_PEAK_TOPS, 3),
                       "moved_bytes_per_launch": moved[k], "hbm_frac": round(moved[k] / sec / 1e9 / HBM_PEAK_GBS, 4),
                       "traffic": traffic.get(k), "bound": "f64 VALU (prescribed mul + add, no FMA by spec)" if k == "fir" else "on-chip: LDS issue (three 8-byte reads per tap step and lane against four f64 operations) and the latency between a group's barriers; neither the f64 rate nor HBM"}
    out["roofline"] = {"per_kernel": roof, "f64_peak_tops": F64_VALU_PEAK_TOPS, "hbm_peak_gbs": HBM_PEAK_GBS,
                       "traffic_source": f"{PROFILE_TAG}/fir_pmc_traffic.json" if traffic else None}
    if "fir" in k_ms:
        out["fir_f64_valu"] = {"ops_per_launch": fir_ops, "achieved_tops": round(fir_ops / (k_ms["fir"] * 1e-3) / 1e12, 2), "peak_tops": 39.3,
                               "frac": round(fir_ops / (k_ms["fir"] * 1e-3) / 1e12 / 39.3, 3), "note": "prescribed f64 mul + add only (no FMA by spec)"}
    if "resample" in k_ms:
        out["resample_f64_tops"] = round(rs_ops / (k_ms["resample"] * 1e-3) / 1e12, 2)
    if with_contract:
        # the same leg in the contracted order (MX_FLAG_FP_CONTRACT: acc = fma(h[k], x, acc), half the f64 instructions; <= 1 ULP of the spec)
        from mixlab_amd import abi
        g.close()
        fc = fir_leg(torch, stream, local_rank, T, steps, warmup, flags=abi.FLAG_FP_CONTRACT, with_contract=False)
        out["fp_contract"] = {"flag": "MX_FLAG_FP_CONTRACT", "parity": "<= 1 ULP of the separate multiply-and-add spec; bit-exact vs the oracle's contract mode (tests/test_gpu_fp_contract.py)",
                              "value": fc["value"], "unit": fc["unit"], "ms_per_step": fc["ms_per_step"], "kernel_ms_per_step": fc["kernel_ms_per_step"],
                              "roofline": {k: {kk: v[kk] for kk in ("ms", "f64_ops_per_launch", "f64_tops", "f64_frac", "moved_bytes_per_launch", "hbm_frac")}
                                           for k, v in fc["roofline"]["per_kernel"].items()},
                              "note": "f64_ops counts the spec's mul and add separately (an fma does two of them): f64_frac can approach 2 x the instruction-rate roof"}
    return out


def video_cpu_baseline(target_seconds=4.0):
    """CPU oracle (C, one thread) on the config-4 cascade: 7 reference VideoMixer cross-fades (+2 bicubic letterbox scales)
    + YUV->RGBA + matrix per composited 1080p frame; a bounded number of frames."""
    import oracle_video as ov   # test infrastructure: used here only as the timed CPU baseline
    import synth

    sizes = [(1920, 1080)] * 6 + [(1280, 720)] * 2
    layers = []
    for k, (w, h) in enumerate(sizes):
        hf = ov.HostFrame(w, h)
        for pl, a in zip(hf.visible(), synth.yuv_pattern(w, h, k, seed=3)):
            pl[:] = a
        layers.append(hf)
    oms = [ov.OracleVideoMixer(a=0, b=1, fader=VIDEO_FADERS[k]) for k in range(7)]

    def one(tick):
        prev = (layers[0], (1, 60), (0, 1))
        for k in range(7):
            out = oms[k].run_tick(tick * 800, [prev, (layers[k + 1], (1, 60), (0, 1)), None, None])
            prev = (out, (1, 60), (0, 1))
        return ov.to_rgba(prev[0], VIDEO_MATRIX)

    t0 = time.perf_counter(); one(0); per = time.perf_counter() - t0
    n = int(max(2, min(200, target_seconds / max(per, 1e-3))))
    t0 = time.perf_counter()
    for i in range(1, n + 1):
        one(i)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port", "sample": f"{n} composited 1080p frames, single thread, {dt:.1f} s"}


def fir_cpu_baseline(T_ref_ticks=8, n_ch=8):
    """CPU oracle (C, one thread) on config 3, a bounded slice: n_ch of the 256 stereo channels for a few ticks."""
    import oracle  # test infrastructure: used here only as the timed CPU baseline
    import synth
    from mixlab_amd.workspace import Workspace

    up, down, tpp = 160, 147, 16
    n = up * tpp
    m = np.arange(n) - (n - 1) / 2.0
    fc = 0.5 / max(up, down) * 0.92
    table = np.ascontiguousarray((2 * fc * np.sinc(2 * fc * m) * np.kaiser(n, 8.6) * up).reshape(tpp, up).T)
    ws = Workspace(44100, 60)
    srcs, rs = [], []
    for k in range(n_ch):
        taps = (synth.uniform(20 + k, 128, -1.0, 1.0) * np.exp(-np.arange(128) / 24.0) * 0.35).astype(np.float64)
        s = ws.source_stereo(); f = ws.fir(taps); r = ws.resample(up, down, table)
        ws.connect(s, 0, f, 0); ws.connect(f, 0, r, 0)
        srcs.append(s); rs.append(r)
    mix = ws.mixer([(0.0, 1.0, k % 2 == 0) for k in range(n_ch)])
    for k, r in enumerate(rs):
        ws.connect(r, 0, mix, k)
    og = oracle.OracleGraph(ws)
    for k, s in enumerate(srcs):
        og.set_source(s, synth.noise(60 + k, 2 * 735))
    t0 = time.perf_counter(); og.run_ticks(0, 4); per = (time.perf_counter() - t0) / 4
    n_ticks = int(max(T_ref_ticks, min(20000, 3.0 / max(per, 1e-6))))
    t0 = time.perf_counter(); og.run_ticks(4, n_ticks); dt = time.perf_counter() - t0
    return {"value": n_ch * n_ticks / dt, "unit": "channel-ticks/s", "cores": 1, "kind": "port",
            "sample": f"{n_ch} of the 256 stereo channels x {n_ticks} ticks, single thread, {dt:.1f} s"}


def _profile_dir():
    """the newest profiles/rNN that holds counter summaries (tools/profile_round.sh copies them there before it runs the default command)"""
    ds = sorted(d for d in (ROOT / "profiles").glob("r[0-9][0-9]") if (d / "pmc_traffic.json").exists())
    return ds[-1] if ds else ROOT / "profiles" / "r05"


PROFILE_DIR = _profile_dir()
PROFILE_TAG = f"profiles/{PROFILE_DIR.name}"


def _kernel_hash(family):
    sys.path.insert(0, str(ROOT / "tools"))
    from kernel_hash import kernel_hash
    return kernel_hash(family)


def pmc_traffic(kernel, args, world, toggling, fc=None):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes (profiles/rNN/pmc_traffic.json of the newest round, collected
    with this same command under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for wide streaming reads); None when the run's configuration differs from the profiled one
    OR the kernel sources have changed since the profile was collected (their hash is recorded in the JSON) -- counters cannot be
    read from inside the process, and a stale figure is worse than none."""
    fc = bool(args.fp_contract) if fc is None else fc
    name = "pmc_traffic_fc.json" if fc else "pmc_traffic.json"
    try:
        rec = json.load(open(PROFILE_DIR / name))
    except (OSError, ValueError):
        return None, None
    if rec.get("kernel_sources_sha16") != _kernel_hash("audio"):
        return None, f"{PROFILE_TAG}/{name} is STALE (kernel sources changed since it was collected): not copied"
    c = rec.get("config", {})
    same = (c.get("strips") == args.strips and c.get("ticks_per_step") == args.ticks_per_step and c.get("sample_rate") == args.sample_rate
            and c.get("fused") == (not args.no_fuse) and c.get("eq_fast") == bool(args.eq_fast) and c.get("n_gpus") == world
            and c.get("gates_toggle") == bool(toggling) and bool(c.get("fp_contract", False)) == fc)
    if not same or kernel not in rec.get("bytes_per_launch", {}):
        return None, None
    return rec["bytes_per_launch"][kernel], f"{PROFILE_TAG}/{name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; the x2 confirmed on this kernel's whole-line reads by tools/fetch_probe.hip; kernel sources unchanged since)"


def sq_profile(kernel_substr, fc, samples):
    """What the committed SQ counter pass (profiles/rNN/pmc_sq_toggle.json / pmc_sq_fc.json: means per dispatch) says about the dominant kernel, per
    OUTPUT sample of the launch: VALU wave-instructions x 64 lanes / samples.  None when the profile was collected on other kernel sources."""
    try:
        rec = json.load(open(PROFILE_DIR / ("pmc_sq_fc.json" if fc else "pmc_sq_toggle.json")))
    except (OSError, ValueError):
        return None
    if rec.get("kernel_sources_sha16") != _kernel_hash("audio"):
        return None
    for k, v in rec.get("mean_per_dispatch", {}).items():
        if kernel_substr in k and v.get("SQ_INSTS_VALU", 0) > 1e6:
            out = {"kernel": k[-70:], "valu_instructions_per_output_sample": round(v["SQ_INSTS_VALU"] * 64.0 / samples, 2)}
            if v.get("SQ_ACTIVE_INST_VALU"):
                out["valu_active_quad_cycles_per_dispatch"] = v["SQ_ACTIVE_INST_VALU"]   # x 4 cycles / (1024 SIMDs x the dispatch's cycles) = the share of time the VALU pipes are busy
            return out
    return None


def sustained_clock_ghz(kernel_substr, fc=False):
    """The clock the chip held under a kernel in the committed counter pass (profiles/rNN/clock.json, clock_fc.json), or None when that profile was
    collected on other kernel sources."""
    try:
        rec = json.load(open(PROFILE_DIR / ("clock_fc.json" if fc else "clock.json")))
    except (OSError, ValueError):
        return None
    if rec.get("kernel_sources_sha16") != _kernel_hash("audio"):
        return None
    for k, v in rec.get("ghz_by_kernel", {}).items():
        if kernel_substr in k:
            return v["ghz"]
    return None


def scaling_probe(torch, stream, local_rank, abi, Workspace, synth, strips, T, SR, toggling, flags, t1_ms, with_exchange=True):
    """What ONE rank of an N-GPU job computes per step, measured on this GPU: its strip share (strips / N) for T ticks (strong scaling as the
    driver runs it) and for T x N ticks (--scale-ticks: the chunk length per lane of the speculative EqThree stays what it is at N = 1).  The
    exchange is not run here (one GPU): its time is modelled as bytes received per rank and step / 300 GB/s of xGMI and assumed hidden behind
    the next step's compute when shorter (mx_exchange runs on its own stream).  N > 1 is a MODEL until a multi-GPU node runs it."""
    spt = SR // 60
    out = {"fixed_ticks": {}, "scale_ticks": {}}
    for policy, mult in (("fixed_ticks", lambda n: 1), ("scale_ticks", lambda n: n)):
        for n in (2, 4, 8):
            if strips % n:
                continue
            Tn, sn = T * mult(n), strips // n
            ws, mix, srcs, trigs = build_strips(abi, Workspace, synth, sn, 0, SR, want_trigs=True)
            g = ws.build(max_ticks_per_run=Tn, flags=flags, device=local_rank, stream=stream.cuda_stream)
            gen = torch.Generator(device="cuda"); gen.manual_seed(0x4D58 + n)
            noise = (torch.rand(Tn * spt, generator=gen, device="cuda", dtype=torch.float32) * 2.0 - 1.0).contiguous()
            for s_ in srcs:
                g.bind_source_device(s_, noise.data_ptr())             # every strip of the probe reads the same device-resident noise
            k = 3
            evs = [gate_events(abi, trigs, 0, i * Tn, Tn) if toggling else None for i in range(2 + k)]
            for i in range(2 + k):
                if i == 2:
                    g.sync(); t0 = time.perf_counter()
                if evs[i] is not None:
                    g.schedule_params_batch(evs[i][0], evs[i][1])
                g.run_ticks(i * Tn, Tn)
            g.sync()
            ms = (time.perf_counter() - t0) / k * 1e3
            # the same rank with an exchange in the loop (fixed T only): a ONE-rank RCCL communicator -- the pack, the library's RCCL call and the combine graph really run
            # (behind the held-back Mixer bank); what no single GPU can show is the wire
            ms_x = None
            if policy == "fixed_ticks" and with_exchange:
                from mixlab_amd.exchange import BusExchange, unique_id
                ex1 = BusExchange(g, mix, Tn, 0, 1, mode="allgather", nccl_id=unique_id())
                kx = 4
                evx = [gate_events(abi, trigs, 0, (2 + k + i) * Tn, Tn) if toggling else None for i in range(2 + kx)]
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


def exchange_parity(torch, dist, np, g, ex, mix, T, step, world):
    """Is the exchange's combined bus the rank-ordered f32 sum of the partial buses (the graph N x Mixer(strips / N) -> Mixer(N, unity),
    src/module/mixer.rs:57-68: master starts at +0.0 and adds channel after channel)?  Checked without any of the exchange's own code: every
    rank's raw partial Master / Cue (read back from its graph) travels through ONE plain all_gather of torch.distributed (ncclAllGather), the
    sum is made on the host in rank order with numpy f32 adds, and compared bit for bit with mx_exchange_read_result.  Collective: every
    rank calls it; returns this rank's verdict."""
    part = np.concatenate([g.read_output(mix, 0, T, True), g.read_output(mix, 1, T, True)])
    mine = torch.from_numpy(part).cuda()
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
    return {"verdict": "MISMATCH", "samples_compared": int(got.size), "mismatching": int(bad.size), "first_index": i,
            "got": float(got[i]), "want": float(acc[i])}


def other_rate_leg(torch, np, synth, abi, Workspace, args, local_rank, stream, sample_rate, T, steps=5):
    """The headline job at ANOTHER sample rate on a graph of its own -- 44.1 kHz is the reference's own rate (src/engine.rs SAMPLE_RATE),
    48 kHz the one config 2 is written for.  Same strips, same gate schedule, same T; not part of `value`."""
    ws, mix, srcs, trigs = build_strips(abi, Workspace, synth, args.strips, 0, sample_rate, want_trigs=True)
    spt = ws.spt
    flags = abi.FLAG_FP_CONTRACT if args.fp_contract else 0
    with torch.cuda.stream(stream):
        g = ws.build(max_ticks_per_run=T, flags=flags, device=local_rank, stream=stream.cuda_stream)
        base_ticks = min(T, 256)
        for j, sn in enumerate(srcs):
            blk = synth.noise(j, base_ticks * spt)
            g.write_source(sn, np.tile(blk, (T + base_ticks - 1) // base_ticks)[: T * spt], T)
        evs = [gate_events(abi, trigs, 0, i * T, T) for i in range(steps + 1)]

        def step(i):
            if evs[i] is not None:
                g.schedule_params_batch(evs[i][0], evs[i][1])
            g.run_ticks(i * T, T)
        step(0)
        g.sync(); torch.cuda.synchronize()
        g.profile_enable(not args.no_profile)
        t0 = time.perf_counter()
        for i in range(1, steps + 1):
            step(i)
        g.sync(); torch.cuda.synchronize()      # (mx_graph_sync: a held-back Mixer bank of the last step included)
        dt = time.perf_counter() - t0
        g.profile_enable(False)
        by_kind, _tot, n_prof = g.profile_collect()
        ran, repaired = g.eq_spec_stats()
        r_parity = None
        if not args.no_headline_parity and not args.eq_fast and not args.no_fuse:
            def src_of(j):
                blk = synth.noise(j, base_ticks * spt)
                return np.tile(blk, (T + base_ticks - 1) // base_ticks)[: T * spt]
            r_parity = headline_parity(g, abi, Workspace, synth, args, sample_rate, T, 1 + steps, 0, args.strips, mix, True, bool(args.fp_contract), src_of)
        g.close()
    return {"sample_rate": sample_rate, "headline_parity": r_parity, "samples_per_tick": spt, "ticks_per_step": T, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
            "value": args.strips * T * steps / dt, "unit": "channel-ticks/s",
            "kernel_ms_per_step": {k: round(v / max(1, n_prof), 5) for k, v in sorted(by_kind.items()) if v > 0},
            "eq_spec": {"chunks_run": int(ran), "chunks_repaired": int(repaired)},
            "note": "a channel-tick at 44.1 kHz is 735 samples against 800: per SAMPLE this is value x 735 / 800 of the headline's"}


def scaled_ticks_leg(torch, dist, np, synth, abi, shard, Workspace, args, rank, world, local_rank, stream, nccl_id_fn, toggling):
    """N > 1: the OTHER tick policy beside the one the headline ran -- T x N ticks per step, so that a rank's chunk length (and the share of
    warm-up samples its speculative EqThree runs) is what it is on one GPU.  Own graph, own exchange; barrier + max over ranks like the headline."""
    from mixlab_amd.exchange import BusExchange
    T, SR = args.ticks_per_step * world, args.sample_rate
    spt = SR // 60
    first, local_strips = shard.strip_range(rank, world, args.strips)
    ws, mix, srcs, trigs = build_strips(abi, Workspace, synth, local_strips, first, SR, want_trigs=True)
    g = ws.build(max_ticks_per_run=T, flags=(abi.FLAG_EQ_FAST if args.eq_fast else 0) | (abi.FLAG_FP_CONTRACT if args.fp_contract else 0), device=local_rank, stream=stream.cuda_stream)
    base_ticks = min(T, 256)
    for j, sn in enumerate(srcs):
        blk = synth.noise(first + j, base_ticks * spt)
        g.write_source(sn, np.tile(blk, (T + base_ticks - 1) // base_ticks)[: T * spt], T)
    ex = BusExchange(g, mix, T, rank, world, mode=args.exchange, nccl_id=nccl_id_fn())
    steps, warm = min(args.steps, 6), 2
    events = [gate_events(abi, trigs, first, i * T, T) if toggling else None for i in range(warm + steps + 1)]

    def step(i):
        if events[i] is not None:
            g.schedule_params_batch(events[i][0], events[i][1])
        g.run_ticks(i * T, T)
        ex.submit(i)
    with torch.cuda.stream(stream):
        for i in range(warm):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warm + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        parity = exchange_parity(torch, dist, np, g, ex, mix, T, warm + steps - 1, world) if ex.mode != "allreduce" else None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    out = {"policy": "ticks per step scaled with N (a rank's chunks keep their one-GPU length)", "ticks_per_step": T, "steps": steps, "ms_per_step": dt / steps * 1e3,
           "value": args.strips * T * steps / dt, "unit": "channel-ticks/s", "exchange_mode": ex.mode, "parity": parity}
    ex.close(); g.close()
    return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_RESULT_FD = None


def main():
    # stdout carries the result line and nothing else: file descriptor 1 is pointed at stderr for the whole run (libraries print banners from C code, past sys.stdout)
    # and the line is written to the original descriptor at the end
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--strips", type=int, default=1024)
    ap.add_argument("--ticks-per-step", type=int, default=2048, help="ticks batched per submission (SURVEY 8d: throughput mode; 1 = real-time mode)")
    ap.add_argument("--sample-rate", type=int, default=48000)
    ap.add_argument("--eq-fast", action="store_true", help="MX_FLAG_EQ_FAST: the time-parallel EqThree scan (<= 1 ULP, NOT bit-exact) instead of the exact default")
    ap.add_argument("--no-buses-leg", action="store_true", help="skip the group-bus topology leg (8 x Mixer(strips / 8) -> Mixer(8))")
    ap.add_argument("--no-one-stream-leg", action="store_true", help="skip the MX_OVERLAP_AUTO=0 leg (each kernel alone on the chip) that the roofline block quotes")
    ap.add_argument("--overlap-tail", action="store_true",
                    help="MX_FLAG_OVERLAP_TAIL: the Mixer bank of step k on a second stream beside step k + 1's EqThree group (measured SLOWER: 6.53 vs 6.00 ms per step, DESIGN.md 5.2)")
    ap.add_argument("--no-held-leg", action="store_true", help="skip the held-gates comparison leg (counter passes: keep the dispatches of one kind)")
    ap.add_argument("--hold-gates", action="store_true", help="no per-tick gate schedule: every gate held for the whole run (the round-1 configuration)")
    ap.add_argument("--no-fuse", action="store_true", help="materialise every port (MX_FLAG_NO_FUSE): module-boundary traffic")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", choices=["auto", "slices", "allgather", "allreduce"], default="auto",
                    help="N > 1 bus exchange (mx_exchange_*): ordered reduce-scatter + all-gather over time slices (auto: N >= 4), one all-gather of "
                         "the whole partial buses (auto: N < 4), or ncclAllReduce (NOT the sum order of a reference graph: non-parity)")
    ap.add_argument("--force-combine", action="store_true", help="run the N > 1 exchange path at N = 1 (single-rank RCCL group)")
    ap.add_argument("--scale-ticks", action="store_true", help="N > 1: T = --ticks-per-step x N ticks per step, so that a rank's chunk length (and with it the share of "
                    "warm-up samples its speculative EqThree runs) stays what it is on one GPU; the model line shows both policies")
    ap.add_argument("--no-scaled-leg", action="store_true", help="N > 1: skip the second tick policy (T x N ticks per step) that the line carries beside the headline's")
    ap.add_argument("--no-scaling-probe", action="store_true", help="skip the one-GPU measurement of what a rank of a 2 / 4 / 8-GPU job computes per step (scaling_model)")
    ap.add_argument("--no-profile", action="store_true", help="debug: no per-kernel hipEvents in the timed region (roofline omitted)")
    ap.add_argument("--repeats", type=int, default=4, help="further repetitions of the K timed steps after the headline region (spread of the clock)")
    ap.add_argument("--no-t-sweep", action="store_true", help="skip the shorter-submission legs (T = 64 and 1024 ticks, SURVEY 8d)")
    ap.add_argument("--no-realtime", action="store_true", help="skip the one-tick-per-submission leg (hundreds of tiny dispatches: slow under a counter-collecting profiler)")
    ap.add_argument("--no-north-star", action="store_true", help="skip the 10 240-strip + 8-layer real-time leg")
    ap.add_argument("--no-contract-leg", action="store_true", help="skip the MX_FLAG_FP_CONTRACT leg (the same graph in the contracted order, <= 1 ULP)")
    ap.add_argument("--fp-contract", action="store_true", help="run the HEADLINE in the contracted order (MX_FLAG_FP_CONTRACT: <= 1 ULP, NOT the reference's bits); "
                    "the default line reports it as the `fp_contract` leg beside the exact headline")
    ap.add_argument("--no-rate-leg", action="store_true", help="skip the 44.1 kHz leg (the headline job at the reference's own sample rate)")
    ap.add_argument("--no-headline-parity", action="store_true", help="skip the oracle replay of the timed submissions at their own shape (headline_parity)")
    ap.add_argument("--parity-strips", type=int, default=16, help="strips replayed through the CPU oracle from tick 0 for headline_parity")
    ap.add_argument("--no-material-leg", action="store_true", help="skip the realistic-material (muted strips, silences) and poisoned-strip legs")
    ap.add_argument("--fir-ticks", type=int, default=128, help="ticks per step of the FIR + resampler leg (BASELINE configs[2]; 0 = skip)")
    ap.add_argument("--video-frames", type=int, default=1920, help="composited frames in the video leg (0 = skip)")
    ap.add_argument("--video-band-as", default=None, metavar="R/W", help="single GPU: run the video leg as rank R of a W-rank row-band job")
    ap.add_argument("--video-shard", choices=["replicas", "bands"], default="replicas",
                    help="N > 1: independent 8-layer streams per rank (weak scaling), or ONE stream cut into row bands over the ranks (strong scaling)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import synth
    from mixlab_amd import abi, shard
    from mixlab_amd.workspace import Workspace

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_combine
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    T, SR = args.ticks_per_step * (world if args.scale_ticks else 1), args.sample_rate
    spt = SR // 60
    first, local_strips = shard.strip_range(rank, world, args.strips)
    toggling = not args.hold_gates

    stream = torch.cuda.Stream()
    flags = (abi.FLAG_EQ_FAST if args.eq_fast else 0) | (abi.FLAG_NO_FUSE if args.no_fuse else 0) | (abi.FLAG_FP_CONTRACT if args.fp_contract else 0)
    overlap = args.overlap_tail and not (world > 1 or args.force_combine)   # the exchange packs the buses on the compute stream: one stream there
    if overlap:
        flags |= abi.FLAG_OVERLAP_TAIL
    ws, mix, srcs, trigs = build_strips(abi, Workspace, synth, local_strips, first, SR, want_trigs=True)
    g = ws.build(max_ticks_per_run=T, flags=flags, device=local_rank, stream=stream.cuda_stream)

    # synthetic sources, resident in HBM before the timed region (uploaded once, re-read every step)
    # (a seeded 256-tick noise block per strip, repeated to fill the step: host-side generation stays in seconds)
    base_ticks = min(T, 256)
    for j, s in enumerate(srcs):
        blk = synth.noise(first + j, base_ticks * spt)
        g.write_source(s, np.tile(blk, (T + base_ticks - 1) // base_ticks)[: T * spt], T)

    ex = None
    nccl_id = None
    if use_dist:
        # the exchange is the library's (mx_exchange_*: RCCL called from libmixlab_gpu.so); torch.distributed only carries the
        # job's ncclUniqueId from rank 0 to the others, the barriers and the max-over-ranks of the clock
        from mixlab_amd.exchange import BusExchange, unique_id
        box = [unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        nccl_id = box[0]
        ex = BusExchange(g, mix, T, rank, world, mode=args.exchange, nccl_id=nccl_id)

    # every step's gate toggles, built before any clock starts (the schedule is host data, like the params a UI would send)
    n_regions = 1 + (max(0, args.repeats) if not use_dist else 0)
    n_sched = args.warmup + args.steps * n_regions + 4
    events = {i: (gate_events(abi, trigs, first, i * T, T) if toggling else None) for i in range(n_sched)}

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
        # g.sync() = mx_graph_sync: every launch of the graph, on both of its streams, INCLUDING a Mixer bank the library holds back for the next run's EqThree launch
        # (automatic overlap): the region starts with nothing of this graph outstanding and ends when all K steps' launches -- K EqThree groups and K Mixer
        # banks, the last bank alone -- have completed
        g.sync()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for i in range(k):
            step(i0 + i, scheduled)
        g.sync()
        torch.cuda.synchronize()
        barrier()
        return time.perf_counter() - t0

    held = None
    with torch.cuda.stream(stream):
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        # per-kernel hipEvents cost a few us of stream time each: at N = 1 they sit inside the timed region (the
        # roofline contract), at N > 1 -- where a step is ~8x shorter -- they are taken on extra steps after it
        prof_in_region = not args.no_profile and not use_dist
        g.profile_enable(prof_in_region)
        dt = timed_region(args.warmup, args.steps)                      # THE timed region: exactly K steps
        nxt = args.warmup + args.steps
        if use_dist and not args.no_profile:
            g.profile_enable(True)
            for i in range(3):
                step(nxt + i)
            torch.cuda.synchronize()
        g.profile_enable(False)
        by_kind, prof_total_ms, n_prof = g.profile_collect()
        spec_ran, spec_repaired = g.eq_spec_stats()
        # the timed submissions against the oracle at their own shape (outside every clock; before anything else overwrites the last step's outputs)
        parity = None
        if not args.no_headline_parity and not args.eq_fast and not args.no_fuse and rank == 0:
            def src_of(j):
                blk = synth.noise(first + j, base_ticks * spt)
                return np.tile(blk, (T + base_ticks - 1) // base_ticks)[: T * spt]
            parity = headline_parity(g, abi, Workspace, synth, args, SR, T, args.warmup + args.steps, first, local_strips, mix, toggling,
                                     bool(args.fp_contract), src_of)
        # the spread of the clock: the same K steps again, a few times (not part of `value`)
        rep_ms = [dt / args.steps * 1e3]
        if not use_dist:
            for r in range(max(0, args.repeats)):
                rep_ms.append(timed_region(nxt, args.steps) / args.steps * 1e3)
                nxt += args.steps
        # the same job with every gate held where it stands (round 1 measured this): the Envelope is flat most of the time
        if toggling and not use_dist and not args.no_held_leg:
            g.profile_enable(not args.no_profile)
            dt_h = timed_region(nxt, min(args.steps, 10), scheduled=False)
            g.profile_enable(False)
            hk, _ht, hn = g.profile_collect()
            held = {"ms_per_step": dt_h / min(args.steps, 10) * 1e3, "value": args.strips * T * min(args.steps, 10) / dt_h, "unit": "channel-ticks/s",
                    "kernel_ms_per_step": {k: round(v / max(1, hn), 5) for k, v in sorted(hk.items()) if v > 0},
                    "note": "gates held for the whole run: the inline Envelope is flat (sustain / silent) except for the first seconds"}

    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # The same job in the CONTRACTED order (MX_FLAG_FP_CONTRACT): a second graph over the SAME resident sources (bound, not copied), own
    # state; every f32 within 1 ULP of the exact order, bit-exact vs the oracle's contract mode (tests/test_gpu_fp_contract.py).  Not `value`.
    contract = None
    if not use_dist and not args.no_contract_leg and not args.eq_fast and not args.fp_contract:
        with torch.cuda.stream(stream):
            g_fc = ws.build(max_ticks_per_run=T, flags=(flags & ~abi.FLAG_OVERLAP_TAIL) | abi.FLAG_FP_CONTRACT, device=local_rank, stream=stream.cuda_stream)
            for sn in srcs:
                g_fc.bind_source_device(sn, g.output_device_ptr(sn, 0)[0])
            n_c = min(args.steps, 10)
            evs = [gate_events(abi, trigs, first, i * T, T) if toggling else None for i in range(2 + n_c)]

            def step_fc(i):
                if evs[i] is not None:
                    g_fc.schedule_params_batch(evs[i][0], evs[i][1])
                g_fc.run_ticks(i * T, T)
            for i in range(2):
                step_fc(i)
            g_fc.sync(); torch.cuda.synchronize()
            g_fc.profile_enable(not args.no_profile)
            t0 = time.perf_counter()
            for i in range(n_c):
                step_fc(2 + i)
            g_fc.sync(); torch.cuda.synchronize()
            dt_c = time.perf_counter() - t0
            g_fc.profile_enable(False)
            ck, _ct, cn = g_fc.profile_collect()
            c_ran, c_rep = g_fc.eq_spec_stats()
            c_parity = None
            if parity is not None:
                c_parity = headline_parity(g_fc, abi, Workspace, synth, args, SR, T, 2 + n_c, first, local_strips, mix, toggling, True, src_of)
            contract = {"flag": "MX_FLAG_FP_CONTRACT", "ms_per_step": dt_c / n_c * 1e3, "value": args.strips * T * n_c / dt_c, "unit": "channel-ticks/s", "steps": n_c,
                        "kernel_ms_per_step": {k: v / max(1, cn) for k, v in sorted(ck.items()) if v > 0},
                        "eq_spec": {"chunks_run": c_ran, "chunks_repaired": c_rep},
                        "headline_parity": c_parity,
                        "parity": "every f32 output within 1 ULP of the reference's order (NOT its bits); bit-exact vs the oracle's contract mode (tests/test_gpu_fp_contract.py)",
                        "what": "the reference's f64 expressions with each multiply fused into the add that consumes it: EqThree 26 instead of 36 f64 instructions per sample "
                                "(eq_three.rs:76-88,117-124), Envelope decay and Amplifier depth() one fma each"}
            g_fc.close()

    # The same job with every launch group on ONE stream (MX_OVERLAP_AUTO=0): what each kernel takes when it has the chip to itself -- the figures the roofline block
    # quotes beside those of the timed region, where the Mixer bank of step k runs beside step k + 1's EqThree group.  Own graph over the same resident sources; not `value`.
    one_stream = None
    overlap_active = g.tail_stream() is not None
    if overlap_active and not use_dist and not args.no_profile and not args.no_one_stream_leg:
        with torch.cuda.stream(stream):
            os.environ["MX_OVERLAP_AUTO"] = "0"
            try:
                g1 = ws.build(max_ticks_per_run=T, flags=flags & ~abi.FLAG_OVERLAP_TAIL, device=local_rank, stream=stream.cuda_stream)
            finally:
                os.environ.pop("MX_OVERLAP_AUTO", None)
            for sn in srcs:
                g1.bind_source_device(sn, g.output_device_ptr(sn, 0)[0])
            n_1 = min(args.steps, 8)
            evs1 = [gate_events(abi, trigs, first, i * T, T) if toggling else None for i in range(2 + n_1)]

            def step_1(i):
                if evs1[i] is not None:
                    g1.schedule_params_batch(evs1[i][0], evs1[i][1])
                g1.run_ticks(i * T, T)
            for i in range(2):
                step_1(i)
            g1.sync(); torch.cuda.synchronize()
            g1.profile_enable(True)
            t0 = time.perf_counter()
            for i in range(n_1):
                step_1(2 + i)
            g1.sync(); torch.cuda.synchronize()
            dt_1 = time.perf_counter() - t0
            g1.profile_enable(False)
            k1, _t1, nn1 = g1.profile_collect()
            one_stream = {"env": "MX_OVERLAP_AUTO=0", "ms_per_step": round(dt_1 / n_1 * 1e3, 4), "value": args.strips * T * n_1 / dt_1, "unit": "channel-ticks/s", "steps": n_1,
                          "kernel_ms_per_step": {k: round(v / max(1, nn1), 5) for k, v in sorted(k1.items()) if v > 0}}
            g1.close()

    exch = None
    if ex is not None:
        # what the exchange costs on its own stream: 4 more steps, the library's own event pair around the exchange of each
        with torch.cuda.stream(stream):
            for i in range(4):
                g.run_ticks((nxt + i) * T, T)
                ex.submit(nxt + i)
                torch.cuda.synchronize()
                if i == 0:
                    ex_ms_all = []
                ex_ms_all.append(ex.elapsed_ms(nxt + i))
        ex_ms = sorted(ex_ms_all)[len(ex_ms_all) // 2]
        if ex.world != world:
            raise SystemExit(f"the exchange's communicator has {ex.world} ranks, the job {world}")
        exch = {"mode": ex.mode, "rccl_ranks": ex.world, "transport": "RCCL, called by libmixlab_gpu.so (mx_exchange_*)",
                "bytes_received_per_rank_per_step": ex.bytes_received_per_step(),
                "exchange_ms_per_step": round(ex_ms, 4), "parity": "rank-ordered f32 sum (the graph N x Mixer(strips/N) -> Mixer(N))" if ex.mode != "allreduce"
                else "NONE: ncclAllReduce order is not a reference graph's"}
        if ex.mode != "allreduce":
            # parity evidence that needs none of the exchange's own code (collective: every rank takes part; rank 0 reports)
            with torch.cuda.stream(stream):
                exch["parity_check"] = exchange_parity(torch, dist, np, g, ex, mix, T, nxt + 3, world)
            if world > 1:
                ok = torch.tensor([1 if exch["parity_check"]["verdict"] == "bit-exact" else 0], device="cuda")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                exch["parity_check"]["all_ranks"] = "bit-exact" if int(ok.item()) == 1 else "MISMATCH on some rank"
        if ex.mode == "allreduce" and world > 1:
            # measured deviation of the all-reduce from the ordered sum of the same partial buses
            box = [unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ordered = BusExchange(g, mix, T, rank, world, mode="allgather", nccl_id=box[0])
            with torch.cuda.stream(stream):
                g.run_ticks((nxt + 8) * T, T); ex.submit(nxt + 8); ordered.submit(0)
                torch.cuda.synchronize()
            exch["max_ulp_vs_ordered_sum"] = ex.max_ulp_vs(nxt + 8, *ordered.result(0))
            ordered.close()

    other_policy = None
    if use_dist and not args.scale_ticks and not args.no_scaled_leg and (world > 1 or args.force_combine):
        def fresh_id():
            box = [unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            return box[0]
        other_policy = scaled_ticks_leg(torch, dist, np, synth, abi, shard, Workspace, args, rank, world, local_rank, stream, fresh_id, toggling)

    # The same strips mixed through GROUP BUSES (8 x Mixer(strips / 8) -> Mixer(8)): a topology the reference expresses with its own Mixer module, and the shape a console
    # has.  The second-stream mode takes the bank AND the master above it as its tail.  Own graphs over the headline graph's resident sources; not `value`.
    buses_leg = None
    if not use_dist and not args.no_buses_leg and not args.no_fuse and args.strips % 8 == 0:
        with torch.cuda.stream(stream):
            buses_leg = {}
            for label, auto in (("second_stream", True), ("one_stream", False)):
                if auto:
                    os.environ.pop("MX_OVERLAP_AUTO", None)
                else:
                    os.environ["MX_OVERLAP_AUTO"] = "0"
                try:
                    wsb = Workspace(SR, 60); gm, sb, tb = [], [], []
                    for j in range(8):
                        wsb, m_, s_, t_ = build_strips(abi, Workspace, synth, args.strips // 8, j * (args.strips // 8), SR, ws=wsb, total=args.strips, want_trigs=True)
                        gm.append(m_); sb += s_; tb += t_
                    master = wsb.mixer([(0.0, 1.0, False)] * 8)
                    for j, m_ in enumerate(gm):
                        wsb.connect(m_, 0, master, j)
                    gb = wsb.build(max_ticks_per_run=T, flags=flags & ~abi.FLAG_OVERLAP_TAIL, device=local_rank, stream=stream.cuda_stream)
                finally:
                    os.environ.pop("MX_OVERLAP_AUTO", None)
                for sn_b, sn in zip(sb, srcs):
                    gb.bind_source_device(sn_b, g.output_device_ptr(sn, 0)[0])
                kb = min(args.steps, 8)
                evb = [gate_events(abi, tb, 0, i * T, T) if toggling else None for i in range(2 + kb)]
                for i in range(2 + kb):
                    if i == 2:
                        gb.sync(); tb0 = time.perf_counter()
                    if evb[i] is not None:
                        gb.schedule_params_batch(evb[i][0], evb[i][1])
                    gb.run_ticks(i * T, T)
                gb.sync()
                dtb = (time.perf_counter() - tb0) / kb
                buses_leg[label] = {"ms_per_step": round(dtb * 1e3, 4), "value": args.strips * T / dtb, "unit": "channel-ticks/s", "steps": kb,
                                    "mixer_groups_beside_next_eq_three": gb.tail_stream() is not None}
                gb.close()
            buses_leg["topology"] = f"8 x Mixer({args.strips // 8}) -> Mixer(8, unity), {T} ticks per step, gates as in the headline"

    # real-time regime: one 60 Hz tick per submission, synchronised every tick like a live engine
    realtime = None
    if not use_dist and not args.no_realtime:
        with torch.cuda.stream(stream):
            base_t = (nxt + 16) * T
            for i in range(20):
                g.run_ticks(base_t + i, 1)
            g.sync()
            n_rt = 300
            t0 = time.perf_counter()
            for i in range(n_rt):
                g.run_ticks(base_t + 20 + i, 1)
                g.sync()
            tick_us = (time.perf_counter() - t0) / n_rt * 1e6
        realtime = {"ticks_per_submission": 1, "tick_us": round(tick_us, 1), "tick_budget_us": round(1e6 / 60.0, 1),
                    "headroom": round(1e6 / 60.0 / tick_us, 1), "note": "submit + wait per tick (host-paired), same 1024-strip graph, exact EqThree"}

    # shorter submissions of the same strips (T in {1, 64, 1024}; T = 1 is the real-time leg above), gates toggling as in the headline.  Each length runs on a graph
    # BUILT for it (max_ticks_per_run = T, as a host that submits T ticks at a time builds it), over the headline graph's resident sources: what the library decides from the
    # submission length -- the chunk plan, and for submissions of at most one EqThree wave per SIMD the Mixer bank beside the next submission's EqThree group (automatic since
    # round 5, MX_OVERLAP_AUTO) -- is then what is measured.
    t_sweep = None
    if not use_dist and not args.no_t_sweep:
        t_sweep = {}
        with torch.cuda.stream(stream):
            tick0 = (nxt + 64) * T

            def sweep(Ts, n_sub, auto):
                nonlocal tick0
                if auto:
                    os.environ.pop("MX_OVERLAP_AUTO", None)
                else:
                    os.environ["MX_OVERLAP_AUTO"] = "0"
                try:
                    gs = ws.build(max_ticks_per_run=Ts, flags=flags & ~abi.FLAG_OVERLAP_TAIL, device=local_rank, stream=stream.cuda_stream)
                finally:
                    os.environ.pop("MX_OVERLAP_AUTO", None)
                for sn in srcs:
                    gs.bind_source_device(sn, g.output_device_ptr(sn, 0)[0])
                evs = [gate_events(abi, trigs, first, tick0 + i * Ts, Ts) if toggling else None for i in range(n_sub + 3)]

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
                host_s = time.perf_counter() - t0        # the host's share: scheduling + enqueueing, before the device is waited for
                gs.sync()
                dts = time.perf_counter() - t0
                rec = {"ms_per_step": round(dts / n_sub * 1e3, 4), "value": args.strips * Ts * n_sub / dts, "unit": "channel-ticks/s", "submissions": n_sub,
                       "host_ms_per_step": round(host_s / n_sub * 1e3, 4), "host_ms_per_step_idle_queue": round(host_free_s * 1e3, 4),
                       "mixer_beside_next_eq_three": gs.tail_stream() is not None}
                tick0 += (n_sub + 3) * Ts
                gs.close()
                return rec
            for Ts in (64, 256, 1024):
                if Ts >= T:
                    continue
                # (enough submissions for the steady state: the first few dozen of a new graph are slower -- first-use allocations, the clock settling)
                t_sweep[str(Ts)] = sweep(Ts, {64: 480, 256: 160}.get(Ts, 40), True)
            if "64" in t_sweep and t_sweep["64"]["mixer_beside_next_eq_three"]:
                t_sweep["64_one_stream"] = dict(sweep(64, 480, False), note="MX_OVERLAP_AUTO=0: the same submissions with every launch group on one stream (round 4's default)")

    # the same job at the reference's own sample rate (config 2 is written for 48 kHz; the reference runs at 44.1 kHz)
    rate_leg = None
    if not use_dist and not args.no_rate_leg and toggling and args.sample_rate != 44100:
        rate_leg = other_rate_leg(torch, np, synth, abi, Workspace, args, local_rank, stream, 44100, T)

    # Realistic material and the repair pass's worst case, on the same graph (LAST: the poisoned strip's state stays NaN for ever).
    # The headline's sources are seeded noise, on which every chunk boundary of the speculative EqThree proves itself; a desk also carries
    # muted strips (exact zeros) and programme that falls silent and comes back -- the one input class the proof fails on (poles stall a few
    # ulps from their fixed point) -- and may meet a NaN.  Not part of `value`.
    material = None
    if not use_dist and not args.no_material_leg:
        material = {}
        with torch.cuda.stream(stream):
            rng = np.random.default_rng(0x4D58)
            seg = 48000 * 3                                                     # signal 3 s / silence 2 s / signal ...
            base_ticks = min(T, 256)
            n_muted = n_gaps = 0
            for j, sn in enumerate(srcs):
                kind = j % 4                                                    # 0 muted, 1 programme with silences, 2 / 3 noise as in the headline
                if kind == 0:
                    buf = np.zeros(T * spt, dtype=np.float32); n_muted += 1
                elif kind == 1:
                    blk = synth.noise(first + j, base_ticks * spt)
                    buf = np.tile(blk, (T + base_ticks - 1) // base_ticks)[: T * spt].copy()
                    off = int(rng.integers(0, seg))
                    pos = off
                    while pos < buf.size:
                        buf[pos: pos + 2 * 48000] = 0.0                         # two seconds of digital silence
                        pos += seg + 2 * 48000
                    n_gaps += 1
                else:
                    continue
                g.write_source(sn, buf, T)
            ran0, rep0 = g.eq_spec_stats()
            rs0 = g.eq_repair_stats()
            base_i = nxt + 200
            n_m = min(args.steps, 10)
            for i in list(range(base_i, base_i + 2 + n_m)) + list(range(base_i + 20, base_i + 22 + n_m)):   # the schedules, before any clock starts
                events[i] = gate_events(abi, trigs, first, i * T, T) if toggling else None
            for i in range(2):
                step(base_i + i)
            g.profile_enable(not args.no_profile)
            dt_m = timed_region(base_i + 2, n_m)
            g.profile_enable(False)
            mk, _mt, mn = g.profile_collect()
            ran1, rep1 = g.eq_spec_stats()
            material["daw"] = {"what": f"{n_muted} strips muted (exact zeros), {n_gaps} with 3 s programme / 2 s digital silence alternating, the rest noise; gates toggling as in the headline",
                               "ms_per_step": round(dt_m / n_m * 1e3, 4), "value": args.strips * T * n_m / dt_m, "unit": "channel-ticks/s",
                               "kernel_ms_per_step": {k: round(v / max(1, mn), 5) for k, v in sorted(mk.items()) if v > 0},
                               "eq_spec": {"chunks_run": ran1 - ran0, "chunks_repaired": rep1 - rep0},
                               "repair_pass": {k: v - rs0[k] for k, v in g.eq_repair_stats().items()}}
            # one strip poisoned: a NaN in its source.  Its poles are NaN from then on (the state is carried from step to step); in the step the NaN
            # arrives no chunk after it can prove itself and the repair pass fills the strip's remaining outputs with all 64 lanes of its wave
            bad = np.array(synth.noise(first + 2, min(T, 256) * spt), dtype=np.float32)
            bad = np.tile(bad, (T + min(T, 256) - 1) // min(T, 256))[: T * spt].copy()
            bad[(T * spt) // 3] = np.float32("nan")
            g.write_source(srcs[2], bad, T)
            for i in range(2):
                step(base_i + 20 + i)
            ran2, rep2 = g.eq_spec_stats()
            dt_p = timed_region(base_i + 22, n_m)
            ran3, rep3 = g.eq_spec_stats()
            material["one_strip_poisoned_by_a_nan"] = {"ms_per_step": round(dt_p / n_m * 1e3, 4), "value": args.strips * T * n_m / dt_p, "unit": "channel-ticks/s",
                                                       "eq_spec": {"chunks_run": ran3 - ran2, "chunks_repaired": rep3 - rep2},
                                                       "note": "on top of the daw material; from the second step on the poisoned strip CARRIES an all-NaN state, which stands still under any input: its speculative lanes start from it and prove themselves (the step the NaN arrives in is finished by the repair wave's parallel fill)"}

    scaling = None
    if not use_dist and not args.no_scaling_probe and not args.no_fuse and args.strips % 8 == 0:
        with torch.cuda.stream(stream):
            scaling = scaling_probe(torch, stream, local_rank, abi, Workspace, synth, args.strips, T, SR, toggling, flags & ~abi.FLAG_OVERLAP_TAIL, dt / args.steps * 1e3)

    video = None
    if args.video_frames > 0:
        with torch.cuda.stream(stream):
            video = video_leg(torch, dist, world, stream, local_rank, args.video_frames, args.warmup, shard_mode=args.video_shard, rank=rank,
                              band_as=tuple(int(x) for x in args.video_band_as.split("/")) if args.video_band_as else None)

    north = None
    if not args.no_north_star and not use_dist:
        with torch.cuda.stream(stream):
            north = north_star_realtime_leg(torch, stream, local_rank, abi, Workspace, synth)

    fir = None
    if args.fir_ticks > 0 and not use_dist:
        with torch.cuda.stream(stream):
            fir = fir_leg(torch, stream, local_rank, args.fir_ticks, 10, 2)

    if rank == 0:
        units = args.strips * T * args.steps
        value = units / dt
        frames = T * spt
        bpf = BYTES_PER_FRAME if args.no_fuse else BYTES_PER_FRAME_FUSED
        k_ms = {k: v / n_prof for k, v in by_kind.items() if v > 0} if n_prof else {}

        def moved_bytes(kind):   # bytes one launch of this kind's group has to move on this rank
            if kind == "mixer":
                return (bpf["mixer"] * local_strips + 16) * frames
            return bpf.get(kind, 0) * local_strips * frames

        dom = max(k_ms, key=k_ms.get) if k_ms else None
        roof = None
        if dom is not None:
            avg_ms = k_ms[dom]
            alg = moved_bytes(dom)
            ach = alg / (avg_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(dom, args, world, toggling)
            per_kernel = {}
            for k, ms in sorted(k_ms.items()):
                b = moved_bytes(k)
                if b:
                    per_kernel[k] = {"moved_bytes_per_launch": b, "ms": round(ms, 5), "tb_per_s": round(b / (ms * 1e-3) / 1e12, 3),
                                     "hbm_frac": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            shared = overlap_active and dom == "eq_three" and "mixer" in k_ms
            if shared:
                # The dominant launch does not have the chip to itself: the Mixer bank of the step before runs beside it from its first workgroup to (nearly) its last.
                # The roofline of that WINDOW is what the chip moves in it -- the EqThree group's algorithmic bytes and the bank's -- over the EqThree group's duration.
                alg_eq, alg_mix = alg, moved_bytes("mixer")
                alg = alg_eq + alg_mix
                ach = alg / (avg_ms * 1e-3) / 1e9
                t_mix, _src_mix = pmc_traffic("mixer", args, world, toggling)
                traffic = (traffic + t_mix) if (traffic and t_mix) else None
            roof = {"kernel": dom + ("" if args.no_fuse or dom == "mixer" else " launch group (fused Trigger + Envelope + EqThree + StereoPanner + Amplifier: k_env_ticks + k_eq_three_spec_tiled + k_eq_three_repair)") +
                              (" WITH the Mixer bank of the previous step beside it on the graph's second stream (k_mixer, held back until this launch's workgroups are placed: DESIGN.md 5.2)" if shared else ""),
                    "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_ms": round(avg_ms, 5), "algorithmic_bytes_per_launch": alg,
                    "algorithmic_bytes_per_unit": "2M = 8 B per sample per strip (SURVEY 8d: EqThree channel-tick; source read + strip written as one float per frame)",
                    "kernel_ms_per_step": {k: round(v, 5) for k, v in sorted(k_ms.items())},
                    "kernel_timing": "hipEvents inside the timed region" if not use_dist else "hipEvents on 3 extra steps after the timed region",
                    "per_kernel": per_kernel}
            if shared:
                roof["window"] = {"what": "achieved / frac / algorithmic_bytes_per_launch / traffic above are those of the WINDOW: both kernels' bytes over the EqThree group's duration (avg_launch_ms); "
                                          "per_kernel lists each kernel's own bytes over its own duration in the timed region (they overlap: the durations do not add up to a step)",
                                  "eq_three_bytes": alg_eq, "mixer_bytes": alg_mix,
                                  "eq_three_alone_frac_in_this_window": round(alg_eq / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                if one_stream is not None:
                    o = one_stream["kernel_ms_per_step"]
                    roof["one_stream"] = {"env": "MX_OVERLAP_AUTO=0 (each launch alone on the chip; same job, own graph, measured after the timed region)", "ms_per_step": one_stream["ms_per_step"],
                                          "value": one_stream["value"],
                                          "per_kernel": {k: {"ms": o[k], "moved_bytes_per_launch": moved_bytes(k), "hbm_frac": round(moved_bytes(k) / (o[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                                                         for k in sorted(o) if moved_bytes(k)}}
            if dom == "eq_three":
                sq = sq_profile("k_eq_three_spec_tiled", bool(args.fp_contract), local_strips * frames)
                mand = (26.0 + 5.0 + 12.0 * 0.7) if args.fp_contract else (36.0 + 8.0 + 15.0)
                roof["limiter"] = ("f64 VALU issue, not HBM: " + (f"{sq['valu_instructions_per_output_sample']} VALU instructions per output sample (PMC SQ_INSTS_VALU, {PROFILE_TAG}), " if sq else "") +
                                   f"{mand:.0f} of them the reference's own operations" + (" with each multiply fused into its add" if args.fp_contract else " in the reference's order") +
                                   "; every chunk re-runs a warm-up of 1 280 samples per 6 400; " +
                                   (f"HBM traffic {traffic / alg:.2f}x the algorithmic bytes (PMC; the warm-up re-read is 1.10x of that by construction)" if traffic else "HBM traffic: no current PMC pass") +
                                   "; the board's power limit holds the clock below 2.4 GHz under this kernel (f64_valu.sustained_clock)")
                if sq:
                    roof["sq_profile"] = sq
            else:
                roof["limiter"] = "HBM"
            if dom == "eq_three":
                # the bound that applies: f64 VALU.  Reference arithmetic per strip-sample: EqThree 36 f64 operations (2 x 4 poles x (sub, mul, add)
                # + VSA adds + band split + gains + 2 conversions), Amplifier 6 (conversions, depth, 2 products), Envelope closed form ~13 on the
                # ~70 % of samples where it is not flat (25/500/0.8/200 ms, gates toggling every 30 ticks)
                ops = 36.0 + 6.0 + (13.0 * 0.7 if toggling else 0.0)
                f64_ops = ops * local_strips * frames
                roof["f64_valu"] = {"ops_per_sample_reference": ops, "ops_per_launch": f64_ops, "achieved_tops": round(f64_ops / (avg_ms * 1e-3) / 1e12, 2),
                                    "peak_tops": F64_VALU_PEAK_TOPS, "frac": round(f64_ops / (avg_ms * 1e-3) / 1e12 / F64_VALU_PEAK_TOPS, 3),
                                    "note": "f64 operations of the reference's arithmetic per second against the f64 VALU instruction rate at the 2.4 GHz peak clock (an FMA would count once; none is allowed here)",
                                    }
                ghz = sustained_clock_ghz("k_eq_three_spec_tiled", bool(args.fp_contract))
                if ghz:
                    roof["f64_valu"]["sustained_clock"] = {"ghz": ghz, "peak_tops_at_that_clock": round(F64_VALU_PEAK_TOPS * ghz / 2.4, 1),
                                                           "frac_at_that_clock": round(f64_ops / (avg_ms * 1e-3) / 1e12 / (F64_VALU_PEAK_TOPS * ghz / 2.4), 3),
                                                           "source": f"{PROFILE_TAG}/clock.json: GRBM_GUI_ACTIVE / XCDs / kernel duration under k_eq_three_spec_tiled; a committed measurement of these kernel sources, not read live"}
        if contract is not None:
            ck_ms = contract["kernel_ms_per_step"]
            if "eq_three" in ck_ms:
                alg = moved_bytes("eq_three")
                sec = ck_ms["eq_three"] * 1e-3
                ops_fc = 26.0 + 5.0 + (12.0 * 0.7 if toggling else 0.0)      # f64 INSTRUCTIONS of the contracted order per strip-sample (an fma counts once)
                contract["roofline"] = {"kernel": "eq_three launch group, contracted order (k_env_ticks<true> + k_eq_three_spec_tiled<32, ., ., true, 1> + k_eq_three_repair<true>)",
                                        "bound": "hbm", "achieved": round(alg / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / sec / 1e9 / HBM_PEAK_GBS, 4),
                                        "traffic": pmc_traffic("eq_three", args, world, toggling, fc=True)[0], "traffic_source": pmc_traffic("eq_three", args, world, toggling, fc=True)[1],
                                        "avg_launch_ms": round(ck_ms["eq_three"], 5), "algorithmic_bytes_per_launch": alg,
                                        "sq_profile": sq_profile("k_eq_three_spec_tiled", True, local_strips * frames),
                                        "sustained_clock_ghz": sustained_clock_ghz("k_eq_three_spec_tiled", True),
                                        "f64_valu": {"instructions_per_sample_contracted": ops_fc, "achieved_tops": round(ops_fc * local_strips * frames / sec / 1e12, 2), "peak_tops": F64_VALU_PEAK_TOPS,
                                                     "frac": round(ops_fc * local_strips * frames / sec / 1e12 / F64_VALU_PEAK_TOPS, 3)},
                                        "limiter": "f64 VALU issue, as the exact order: 51.5 VALU instructions per sample in the hot loop (exact: 63), 20 per warm-up sample (28)"}
            contract["kernel_ms_per_step"] = {k: round(v, 5) for k, v in ck_ms.items()}
            contract["speedup_vs_exact"] = round((dt / args.steps * 1e3) / contract["ms_per_step"], 3)
        moved = sum(moved_bytes(k) for k in k_ms)
        rep_sorted = sorted(rep_ms)
        out = {
            "metric": "audio_ch_mixed_per_sec", "value": value, "unit": "channel-ticks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32 (f64 intermediates)",
            "data": "synthetic",
            "config": {"workload": f"{args.strips}-channel Mixer + EqThree + Envelope chain (Trigger->Envelope; noise->EqThree->StereoPanner->Amplifier->Mixer), {SR} Hz f32",
                       "strips": args.strips, "ticks_per_step": T, "samples_per_tick": spt,
                       "gates": "toggle every 30 ticks, phase k mod 60, applied between ticks inside the batch (mx_graph_schedule_params_batch)" if toggling else "held for the whole run",
                       "eq_mode": "time-parallel scan (<= 1 ULP, MX_FLAG_EQ_FAST)" if args.eq_fast else ("CONTRACTED order (MX_FLAG_FP_CONTRACT): <= 1 ULP of the reference, NOT its bits" if args.fp_contract
                                   else "exact order (default): speculative time-parallel kernel, verified bit-exact"),
                       "fusion": "off (every port materialised)" if args.no_fuse else "Trigger+Envelope+EqThree+StereoPanner+Amplifier in one kernel, L==R strips stored mono",
                       "overlap": ("MX_FLAG_OVERLAP_TAIL: " if overlap else "automatic (MX_OVERLAP_AUTO): ") + "the Mixer bank of step k runs on a second stream beside step k + 1's EqThree group (strip ports double-buffered)"
                                  if (overlap or overlap_active) else "off",
                       "parallelism": f"strips sharded x{world}" + (f", {ex.mode}" if ex is not None else "") + (", ticks per step scaled with N (--scale-ticks)" if args.scale_ticks else ""),
                       "rccl_ranks": ex.world if ex is not None else 0},
            "realtime_channels_equiv": value / 60.0,
            "graph_hbm_frac_moved_bytes": round(moved / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
            "eq_spec": {"chunks_run": spec_ran, "chunks_repaired": spec_repaired},
            "headline_parity": parity,
            "roofline": roof,
            "repeats": {"what": f"ms per step of {len(rep_ms)} consecutive regions of {args.steps} steps (the first is the timed region)", "ms_per_step": [round(v, 4) for v in rep_ms],
                        "median": round(rep_sorted[len(rep_sorted) // 2], 4), "min": round(rep_sorted[0], 4), "max": round(rep_sorted[-1], 4),
                        "spread_pct": round((rep_sorted[-1] - rep_sorted[0]) / rep_sorted[len(rep_sorted) // 2] * 100.0, 2)},
            "held_gates": held,
            "one_stream": one_stream,
            "fp_contract": contract,
            "exchange": exch,
            "scaled_ticks": other_policy,
            "realtime": realtime,
            "t_sweep": t_sweep,
            "group_buses": buses_leg,
            "rate_44100": rate_leg,
            "material": material,
            "scaling_model": scaling,
            "north_star_realtime": north,
            "video": video,
            "fir_resample": fir,
        }
        if args.no_cpu_baseline or world > 1:
            out["cpu_baseline"] = None
        else:
            note = native_oracle()
            if video is not None:
                video["cpu_baseline"] = dict(video_cpu_baseline(), build=note)
            if fir is not None:
                fir["cpu_baseline"] = dict(fir_cpu_baseline(), build=note)
            out["cpu_baseline"] = cpu_baseline(Workspace, synth, abi, args.strips, SR, note)
            out["cpu_baseline_all_cores"] = dict(cpu_baseline_all_cores(Workspace, synth, abi, shard, args.strips, SR,
                                                                        1.0 / max(out["cpu_baseline"]["value"], 1.0)), build=note)
        # RCCL prints a version banner through C stdio (flushed at exit when stdout is a pipe):
        # drain it first so the JSON line is the LAST line of stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        # the ONE line of this program's stdout (everything else any library prints while it runs -- RCCL's version banner, for one -- went to stderr: see main())
        sys.stdout.flush()
        os.write(_RESULT_FD if _RESULT_FD is not None else 1, (json.dumps(out) + "\n").encode())

    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
