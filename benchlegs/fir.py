"""BASELINE.json configs[2] (build-specified): 256 stereo channels @44.1 kHz -> 128-tap FIR reverb -> 160/147
polyphase resampler (16 taps per phase) -> 48 kHz-domain Mixer(256).  f64 accumulation in ascending tap order with separate multiply and add,
one rounding to f32."""
from __future__ import annotations

import time

import numpy as np

from .common import F64_VALU_PEAK_TOPS, HBM_PEAK_GBS, PROFILE_TAG, _load

N_CH, SPT, UP, DOWN, TPP = 256, 735, 160, 147, 16


def resampler_table():
    n = UP * TPP
    m = np.arange(n) - (n - 1) / 2.0
    fc = 0.5 / max(UP, DOWN) * 0.92
    return np.ascontiguousarray((2 * fc * np.sinc(2 * fc * m) * np.kaiser(n, 8.6) * UP).reshape(TPP, UP).T)


def fir_graph(synth, n_ch=N_CH):
    from mixlab_amd.workspace import Workspace
    table = resampler_table()
    ws = Workspace(44100, 60)
    srcs, rs = [], []
    for k in range(n_ch):
        taps = (synth.uniform(20 + k, 128, -1.0, 1.0) * np.exp(-np.arange(128) / 24.0) * 0.35).astype(np.float64)
        s = ws.source_stereo(); f = ws.fir(taps); r = ws.resample(UP, DOWN, table)
        ws.connect(s, 0, f, 0); ws.connect(f, 0, r, 0)
        srcs.append(s); rs.append(r)
    mix = ws.mixer([(0.0, 1.0, k % 2 == 0) for k in range(n_ch)])
    for k, r in enumerate(rs):
        ws.connect(r, 0, mix, k)
    return ws, srcs, mix


def fir_leg(torch, stream, local_rank, T, steps, warmup, flags=0, with_contract=True):
    import synth
    ws, srcs, _mix = fir_graph(synth)
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
    g.close()
    frames_in = T * SPT
    k_ms = {k: v / max(1, n_prof) for k, v in by_kind.items() if v > 0}
    out = {"metric": "fir_resample_stereo_ch_ticks_per_sec", "value": N_CH * T * steps / dt, "unit": "channel-ticks/s",
           "workload": "256 stereo channels @44.1 kHz: 128-tap FIR -> 160/147 polyphase resampler (16 taps/phase) -> Mixer(256) @48 kHz",
           "ticks_per_step": T, "ms_per_step": dt / steps * 1e3, "kernel_ms_per_step": {k: round(v, 5) for k, v in sorted(k_ms.items())},
           "realtime_stereo_channels_equiv": N_CH * T * steps / dt / 60.0}
    # per-kernel roofs: the f64 operations the spec prescribes (per output frame 2 channels x taps x (mul + add)) against the f64 VALU rate, the bytes a
    # kernel has to move against HBM, and the HBM traffic of the committed PMC passes while the kernel sources are the ones they were collected on
    rec = _load("fir_pmc_traffic.json", "fir")
    traffic = rec.get("bytes_per_launch", {}) if rec and rec.get("config", {}).get("ticks_per_step") == T else {}
    ops = {"fir": N_CH * frames_in * 2 * 128 * 2, "resample": N_CH * (T * 800) * 2 * TPP * 2}
    moved = {"fir": N_CH * frames_in * 8 * 2, "resample": N_CH * (frames_in + T * 800) * 8}
    bound = {"fir": "f64 VALU (prescribed mul + add, no FMA by spec)",
             "resample": "on-chip: LDS issue (three 8-byte reads per tap step and lane against four f64 operations) and the latency between a group's barriers"}
    roof = {}
    for k in ("fir", "resample"):
        if k in k_ms:
            sec = k_ms[k] * 1e-3
            roof[k] = {"ms": round(k_ms[k], 5), "f64_ops_per_launch": ops[k], "f64_tops": round(ops[k] / sec / 1e12, 2), "f64_frac": round(ops[k] / sec / 1e12 / F64_VALU_PEAK_TOPS, 3),
                       "moved_bytes_per_launch": moved[k], "hbm_frac": round(moved[k] / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic.get(k), "bound": bound[k]}
    out["roofline"] = {"per_kernel": roof, "f64_peak_tops": F64_VALU_PEAK_TOPS, "hbm_peak_gbs": HBM_PEAK_GBS,
                       "traffic_source": f"{PROFILE_TAG}/fir_pmc_traffic.json" if traffic else None}
    if with_contract:
        # the same leg in the contracted order (MX_FLAG_FP_CONTRACT: acc = fma(h[k], x, acc), half the f64 instructions; <= 1 ULP of the spec)
        from mixlab_amd import abi
        fc = fir_leg(torch, stream, local_rank, T, steps, warmup, flags=abi.FLAG_FP_CONTRACT, with_contract=False)
        out["fp_contract"] = {"flag": "MX_FLAG_FP_CONTRACT", "parity": "<= 1 ULP of the separate multiply-and-add spec; bit-exact vs the oracle's contract mode (tests/test_gpu_fp_contract.py)",
                              "value": fc["value"], "unit": fc["unit"], "ms_per_step": fc["ms_per_step"], "kernel_ms_per_step": fc["kernel_ms_per_step"],
                              "roofline": {k: {kk: v[kk] for kk in ("ms", "f64_tops", "f64_frac", "hbm_frac")} for k, v in fc["roofline"]["per_kernel"].items()},
                              "note": "f64_ops counts the spec's mul and add separately (an fma does two of them): f64_frac can approach 2 x the instruction-rate roof"}
    return out
