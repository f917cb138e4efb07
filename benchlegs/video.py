"""BASELINE.json configs[3]: 8 layers (6 x 1080p + 2 x 720p) every tick -> cascade of 7 reference VideoMixer
cross-fades (scale + letterbox for the 720p layers) -> build-specified YUV420P->RGBA + colour matrix.  One composited 1080p RGBA frame per tick.
Every source delivers a NEW frame each tick out of a ring of `n_sets` distinct frames (16 sets x 21.4 MB = 342 MB > the 256 MiB Infinity Cache),
so the layers come from HBM, not from cache.
N > 1: every rank composites its own independent 8-layer stream (independent VideoMixer instances) -- no exchange step, weak
scaling.  --video-shard bands: ONE picture stream over all ranks (strong scaling): rank r composites row band r of every frame (mixlab_amd/shard.py).
N = 1 adds variants of the same job beside the headline one: `no_rest_fader` (every fader inside its travel: all eight layers are read), `alpha`
(three layers carry a coverage plane -- BASELINE's "alpha composite", build-specified: DESIGN.md "Per-pixel alpha") and `mfma_matrix`."""
from __future__ import annotations

import os
import time

from .common import HBM_PEAK_GBS, VIDEO_FADERS, VIDEO_SIZES, dist_device, video_cascade

F = 1920 * 1080 * 3 // 2
F720 = 1280 * 720 * 3 // 2
RGBA = 1920 * 1080 * 4
ALPHA_LAYERS = (2, 5, 7)     # two 1080p layers and a scaled 720p one carry coverage in the `alpha` variant
T_DEFAULT = 1024             # ticks per submission (a throughput knob like the audio leg's 2048: the video pipeline fills -- one scale-only launch -- and drains once per run;
                             # 256 until round 5: 0.28 us per frame of that fill, profiles/r06/video_experiments.md)


def video_leg(torch, dist, world, stream, local_rank, frames, warmup, n_sets=16, shard_mode="replicas", rank=0, band_as=None, only=None, ticks_per_submission=None):
    import alpha_patterns   # seeded coverage planes (numpy only)
    import synth            # seeded synthetic patterns (numpy only)
    from mixlab_amd import shard, video
    from mixlab_amd.workspace import Workspace

    T = int(os.environ.get("VLEG_T", "0")) or ticks_per_submission or min(T_DEFAULT, max(16, frames))
    bands = shard_mode == "bands"
    row0, rows = shard.row_bands(1080, world)[rank] if bands else (0, 1080)
    if band_as:                                                        # one GPU plays rank R of W (what a rank of the sharded job costs)
        bands = True
        rank, of = band_as
        row0, rows = shard.row_bands(1080, of)[rank]
    variants = not bands and world == 1 and only != "main"       # only: "main" = the headline job alone; "alpha" / "no_rest_fader" = that variant AS the measured job
    cuts, rings, alpha_rings = [], [], {}
    for k, (w, h) in enumerate(VIDEO_SIZES):
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
        srcs, _rgba = video_cascade(ws, faders)
        g = ws.build(max_ticks_per_run=T, device=local_rank, stream=stream.cuda_stream)
        for k, (w, h) in enumerate(VIDEO_SIZES):
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

    no_rest = [0.95 if f == 1.0 else f for f in VIDEO_FADERS]
    if only == "alpha":
        dt, steps, dev_ms = run(VIDEO_FADERS, ALPHA_LAYERS, frames, warmup)
    elif only == "no_rest_fader":
        dt, steps, dev_ms = run(no_rest, (), frames, warmup)
    else:
        dt, steps, dev_ms = run(VIDEO_FADERS, (), frames, warmup)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dist_device())
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # bytes per composited frame.  Module-boundary accounting (every VideoMixer output materialised): 7 cross-fades x 3F + 2 scales
    # (F720 in + F out) + RGBA (F in + 4wh out).  MOVED by the two fused kernels: the batched scaler reads 2 x F720 and writes 2 x F; the chain kernel
    # reads the layers its faders leave in play and writes the RGBA frame.  FUSED MINIMUM: every layer read once at its own size + the RGBA frame written.
    alg = 7 * 3 * F + 2 * (F720 + F) + (F + RGBA)
    moved_scaler = 2 * (F720 + F)
    fused_min = 6 * F + 2 * F720 + RGBA

    def moved_chain_of(faders):
        # a step whose fader rests at an end of its travel returns one of its inputs exactly: the launcher drops it and never reads the other layer
        # (mx_k_video.hip chain_matrix_mode).  SURVEY's config-4 faders start with 1.0, so 7 of the 8 layers are read.
        layers_read = 8 - sum(1 for f in faders if f == 1.0)
        return layers_read, layers_read * F + RGBA

    layers_read, moved_chain = moved_chain_of(VIDEO_FADERS)
    n_frames = steps * T * (1 if bands else world)
    workload = "8 layers (6x1080p + 2x720p yuv420p) -> 7 VideoMixer cross-fades (+2 bicubic letterbox scales) -> YUV->RGBA + 3x4 matrix"
    if bands:
        return {"metric": "1080p_composited_fps", "value": n_frames / dt, "unit": "frames/s", "scaling": "strong",
                "shard": f"row bands: rank {rank} of {band_as[1] if band_as else world} composites luma rows [{row0}, {row0 + rows}) of every frame; the 720p layers enter as halo slices and are scaled to the band per tick",
                "workload": workload + ", ONE stream over all ranks", "frames": n_frames, "device_us_per_frame_rank0": round(dev_ms * 1e3, 2)}

    def frac(moved, ms):
        return round(moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None

    out = {"metric": "1080p_composited_fps", "value": n_frames / dt, "unit": "frames/s", "scaling": "weak", "workload": workload,
           "inputs": f"a new frame per layer per tick out of rings of {n_sets} distinct device frames ({n_sets * (6 * F + 2 * F720) / 1e6:.0f} MB in all: HBM-resident, beyond the 256 MiB Infinity Cache)",
           "frames": n_frames, "realtime_1080p60_streams_equiv": n_frames / dt / 60.0, "device_us_per_frame": round(dev_ms * 1e3, 2),
           "moved_bytes_per_frame": moved_scaler + moved_chain, "module_boundary_bytes_per_frame": alg, "fused_minimum_bytes_per_frame": fused_min,
           "hbm_frac_moved_bytes_device": frac(moved_scaler + moved_chain, dev_ms), "hbm_frac_fused_min": frac(fused_min, dev_ms),
           "hbm_frac_moved_bytes_wall": round((moved_scaler + moved_chain) * n_frames / world / dt / 1e9 / HBM_PEAK_GBS, 4),
           "per_kernel_moved_bytes": {"scaler tiles (2 layers)": moved_scaler, "chain tiles": moved_chain, "layers_read_by_the_chain": layers_read, "ticks_per_submission": T}}
    if variants and only is None:
        nf = min(frames, 2048)
        dt2, st2, ms2 = run(no_rest, (), nf, 1)    # every fader inside its travel: nothing is pruned, the chain reads all eight layers
        lr2, mc2 = moved_chain_of(no_rest)
        out["no_rest_fader"] = {"faders": no_rest, "value": st2 * T / dt2, "unit": "frames/s", "device_us_per_frame": round(ms2 * 1e3, 2), "layers_read_by_the_chain": lr2,
                                "moved_bytes_per_frame": moved_scaler + mc2, "hbm_frac_moved_bytes_device": frac(moved_scaler + mc2, ms2), "hbm_frac_fused_min": frac(fused_min, ms2)}
        # BASELINE configs[3]'s "alpha composite" (build-specified): layers 2, 5 (1080p) and 7 (720p, scaled with its coverage) carry a coverage plane
        dt3, st3, ms3 = run(VIDEO_FADERS, ALPHA_LAYERS, nf, 1)
        a1080, a720 = 1920 * 1080, 1280 * 720
        moved_alpha = 2 * a1080 + (a720 + a1080) + a1080          # two planes read by the chain; the 720p plane read + written by the scaler, then read by the chain
        out["alpha"] = {"layers_with_coverage": list(ALPHA_LAYERS), "value": st3 * T / dt3, "unit": "frames/s", "device_us_per_frame": round(ms3 * 1e3, 2),
                        "moved_bytes_per_frame": moved_scaler + moved_chain + moved_alpha, "hbm_frac_moved_bytes_device": frac(moved_scaler + moved_chain + moved_alpha, ms3),
                        "cost_vs_headline_us": round((ms3 - dev_ms) * 1e3, 2), "parity": "bit-exact vs the oracle's rule (tests/test_gpu_video_alpha.py)"}
        # the Q12 colour matrix on the matrix cores (v_mfma_i32_4x4x4_16b_i8, bit-exact) instead of packed f32 FMAs: measured slower, kept opt-in
        os.environ["MX_VIDEO_MFMA_MATRIX"] = "1"
        try:
            dt4, st4, ms4 = run(VIDEO_FADERS, (), nf, 1)
        finally:
            os.environ.pop("MX_VIDEO_MFMA_MATRIX", None)
        out["mfma_matrix"] = {"env": "MX_VIDEO_MFMA_MATRIX=1", "value": st4 * T / dt4, "unit": "frames/s", "device_us_per_frame": round(ms4 * 1e3, 2),
                              "vs_headline_us": round((ms4 - dev_ms) * 1e3, 2), "parity": "bit-exact (integer; tests/test_gpu_video_graph.py)"}
    return out
