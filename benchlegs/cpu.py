"""cpu_baseline: the CPU oracle (oracle/*.c, a C restatement of the reference's algorithms -- TEST INFRASTRUCTURE) timed on this host's cores on a
bounded sample of each workload.  The only place outside tests/ and smoke() that touches oracle/; never inside a GPU clock."""
from __future__ import annotations

import os
import pathlib
import subprocess
import tempfile
import threading
import time

import numpy as np

from .common import ROOT, VIDEO_FADERS, VIDEO_MATRIX, VIDEO_SIZES, build_strips, cpu_model, gate_open


def native_oracle():
    """Build the CPU oracle ON THIS HOST with -O3 -march=native (same sources, same -ffp-contract=off -fno-fast-math: same results) for the
    timed baselines; falls back to the library shipped with the repo.  Must run before `import oracle`."""
    out = pathlib.Path(tempfile.gettempdir()) / f"libmixlab_oracle_native_{os.getpid()}.so"
    try:
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "ARCH=native", f"OUT={out}"], check=True, capture_output=True)
        os.environ["MIXLAB_ORACLE_LIB"] = str(out)
        return "gcc -O3 -march=native -ffp-contract=off -fno-fast-math, built on this host"
    except (OSError, subprocess.CalledProcessError):
        return "library shipped with the repo (-O3 -march=x86-64-v3)"


def audio(Workspace, synth, abi, n_strips, sample_rate, build_note, target_seconds=12.0):
    """The oracle's graph runner (C, one thread) on a bounded sample of the headline workload, gates toggling every 30 ticks
    (ModuleT::update between ticks, as the reference's client_update does)."""
    import oracle  # the checker, here as the timed CPU baseline

    ws, mix, srcs, trigs = build_strips(abi, Workspace, synth, n_strips, 0, sample_rate, want_trigs=True)
    og = oracle.OracleGraph(ws)
    for k, s in enumerate(srcs):
        og.set_source(s, synth.noise(k, ws.spt))
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
    return {"value": n_strips * n_ticks / dt, "unit": "channel-ticks/s", "cores": 1, "kind": "port", "build": build_note,
            "sample": f"{n_strips} strips x {n_ticks} ticks @ {sample_rate} Hz, gates toggling every 30 ticks, one thread (the reference engine is one thread, src/engine.rs:78), {dt:.1f} s",
            "cpu_model": cpu_model(), "host_cores": os.cpu_count()}


def usable_cores():
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
    return cores, quota


def audio_all_cores(Workspace, synth, abi, shard, n_strips, sample_rate, target_seconds=6.0):
    """The same oracle, one graph shard per host core (SURVEY.md section 8d "(ii)"): the strips are partitioned like the multi-GPU job
    (contiguous shards, each with its own sub-Mixer); ctypes releases the GIL, so plain threads run the C runners concurrently."""
    import oracle  # the checker, here as the timed CPU baseline

    cores, quota = usable_cores()
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
    return {"value": n_strips * n_ticks / dt, "unit": "channel-ticks/s", "cores": n_thr, "kind": "port", "cpu_quota_cores": quota, "host_logical_cpus": os.cpu_count(),
            "sample": f"{n_strips} strips in {n_thr} contiguous shards (one thread each) x {n_ticks} ticks @ {sample_rate} Hz, gates held, {dt:.1f} s"}


def video(target_seconds=4.0):
    """The oracle (C, one thread) on the config-4 cascade: 7 reference VideoMixer cross-fades (+2 bicubic letterbox scales) + YUV->RGBA + matrix per frame."""
    import oracle_video as ov   # the checker, here as the timed CPU baseline
    import synth

    layers = []
    for k, (w, h) in enumerate(VIDEO_SIZES):
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
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port", "sample": f"{n} composited 1080p frames, one thread, {dt:.1f} s"}


def fir(n_ch=8):
    """The oracle (C, one thread) on config 3, a bounded slice: n_ch of the 256 stereo channels."""
    import oracle  # the checker, here as the timed CPU baseline
    import synth

    from .fir import fir_graph
    ws, srcs, _mix = fir_graph(synth, n_ch)
    og = oracle.OracleGraph(ws)
    for k, s in enumerate(srcs):
        og.set_source(s, synth.noise(60 + k, 2 * 735))
    t0 = time.perf_counter(); og.run_ticks(0, 4); per = (time.perf_counter() - t0) / 4
    n_ticks = int(max(8, min(20000, 3.0 / max(per, 1e-6))))
    t0 = time.perf_counter(); og.run_ticks(4, n_ticks); dt = time.perf_counter() - t0
    return {"value": n_ch * n_ticks / dt, "unit": "channel-ticks/s", "cores": 1, "kind": "port",
            "sample": f"{n_ch} of the 256 stereo channels x {n_ticks} ticks, one thread, {dt:.1f} s"}
