"""bench.py's one JSON line, as the driver reads it: a small run (64 strips, 64 ticks per step, the secondary legs off) must print the contract's keys with
consistent values -- value = strips x ticks / step time, the roofline block of the dominant launch group, the CPU baseline of the same workload."""
import json
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_bench_line_has_the_contract_keys_and_consistent_numbers():
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--strips", "64", "--ticks-per-step", "64",
           "--no-realtime", "--no-t-sweep", "--no-north-star", "--no-held-leg", "--no-material-leg", "--no-scaling-probe", "--fir-ticks", "0", "--repeats", "0", "--video-frames", "0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"] == "audio_ch_mixed_per_sec" and line["unit"] == "channel-ticks/s"
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["scaling"] in ("weak", "strong") and line["vs_baseline"] is None and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    # value is the whole job's channel-ticks per second of the timed steps
    assert line["value"] == pytest.approx(64 * 64 * 1000.0 / line["ms_per_step"], rel=1e-6)
    rf = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=5e-2, abs=1e-4) and 0.0 < rf["frac"] < 1.0   # (both are printed rounded)
    cb = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == line["unit"]
