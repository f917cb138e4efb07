"""The ONE line of stdout: the contract's keys, `roofline` and `cpu_baseline`, and one number per secondary leg -- small enough for any reader's buffer
(asserted: <= 8 KiB, target <= 4 KiB).  Everything the legs measured goes to bench_full.json beside bench.py."""
from __future__ import annotations

import json

LINE_HARD_CAP = 8192
LINE_TARGET = 4096

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")


def _get(d, *path):
    for p in path:
        if not isinstance(d, dict) or d.get(p) is None:
            return None
        d = d[p]
    return d


def _r(x, nd=4):
    return None if x is None else (round(x, nd) if isinstance(x, float) else x)


def compact(full, full_path):
    line = {k: full[k] for k in CONTRACT}
    line["config"] = {k: full["config"][k] for k in ("workload", "strips", "ticks_per_step", "sample_rate", "gates", "eq_mode", "parallelism", "ticks_policy") if k in full["config"]}
    rf = full.get("roofline")
    if rf:
        line["roofline"] = {"kernel": rf["kernel"].split(" launch group")[0], "bound": rf["bound"], "achieved": rf["achieved"], "peak": rf["peak"], "unit": rf["unit"],
                            "frac": rf["frac"], "traffic": rf["traffic"], "avg_launch_ms": rf["avg_launch_ms"], "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
                            "limiter": rf.get("limiter"), "f64_valu_frac": _get(rf, "f64_valu", "frac"), "window_frac": _get(rf, "window", "frac"),
                            "step_hbm_frac": rf.get("step_hbm_frac")}
    else:
        line["roofline"] = None
    cb = full.get("cpu_baseline")
    line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample")} if cb else None
    hp = full.get("headline_parity")
    line["headline_parity"] = None if hp is None else {"verdict": hp.get("verdict"), "buses": _get(hp, "buses", "verdict"), "strips_checked": hp.get("strips_checked")}
    legs = {
        "video_fps": _r(_get(full, "video", "value"), 1), "video_us_per_frame": _get(full, "video", "device_us_per_frame"),
        "video_hbm_frac": _get(full, "video", "hbm_frac_moved_bytes_device"), "video_hbm_frac_fused_min": _get(full, "video", "hbm_frac_fused_min"),
        "video_cpu_fps": _r(_get(full, "video", "cpu_baseline", "value"), 2),
        "fir_ch_ticks_per_s": _r(_get(full, "fir_resample", "value"), 1), "fir_f64_frac": _get(full, "fir_resample", "roofline", "per_kernel", "fir", "f64_frac"),
        "fir_cpu_ch_ticks_per_s": _r(_get(full, "fir_resample", "cpu_baseline", "value"), 1),
        "fp_contract_value": _r(_get(full, "fp_contract", "value"), 1), "fp_contract_eq_hbm_frac": _get(full, "fp_contract", "roofline", "frac"),
        "one_stream_ms_per_step": _get(full, "one_stream", "ms_per_step"), "rate_44100_value": _r(_get(full, "rate_44100", "value"), 1),
        "realtime_headroom_1024": _get(full, "realtime", "headroom"), "north_star_headroom_10240_plus_video": _get(full, "north_star_realtime", "headroom"),
        "cpu_all_cores_value": _r(_get(full, "cpu_baseline_all_cores", "value"), 1), "cpu_all_cores": _get(full, "cpu_baseline_all_cores", "cores"),
        "model_speedup_8_scaled_ticks": _get(full, "scaling_model", "scale_ticks", "8", "predicted_speedup_vs_1_gpu"),
        "model_speedup_8_fixed_ticks": _get(full, "scaling_model", "fixed_ticks", "8", "predicted_speedup_vs_1_gpu"),
        "exchange_parity": _get(full, "exchange", "parity_check", "all_ranks") or _get(full, "exchange", "parity_check", "verdict"),
        "exchange_ms_per_step": _get(full, "exchange", "exchange_ms_per_step"),
        "other_ticks_policy_value": _r(_get(full, "other_policy", "value"), 1), "other_ticks_policy_ticks": _get(full, "other_policy", "ticks_per_step"),
    }
    line["legs"] = {k: v for k, v in legs.items() if v is not None}
    if full.get("leg_errors"):
        line["leg_errors"] = sorted(full["leg_errors"])         # (names only: the messages are in the full record and on stderr)
    line["full"] = full_path
    return line


def dumps_checked(line):
    """Strict JSON (no NaN / Infinity), one line, within the cap."""
    s = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if "\n" in s or len(s) > LINE_HARD_CAP:
        raise AssertionError(f"bench line is {len(s)} bytes (cap {LINE_HARD_CAP}): move detail to bench_full.json")
    return s


def dumps_within_cap(line):
    """dumps_checked, but a run never ends without its line: should the compact line ever outgrow the cap, the optional parts go (the legs' numbers, then the free-text
    config entries) before the contract's keys do -- they are all in bench_full.json."""
    try:
        return dumps_checked(line)
    except AssertionError:
        slim = dict(line, legs={"dropped": "line over the cap: see `full`"})
        try:
            return dumps_checked(slim)
        except AssertionError:
            slim["config"] = {"workload": str(line["config"].get("workload", ""))[:200]}
            return dumps_checked(slim)
