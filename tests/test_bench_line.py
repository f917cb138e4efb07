"""The compact stdout line of bench.py (benchlegs/line.py) -- CPU test: the full record of the round-5 default run (profiles/r05/bench_default_line.json, the 20.8 KB
line the driver could not parse) goes through compact() and comes out within the cap, strict JSON, with the contract's keys; a NaN anywhere is refused."""
import json
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _full():
    full = json.loads((ROOT / "profiles" / "r05" / "bench_default_line.json").read_text())
    full["config"]["sample_rate"] = 48000
    full["config"]["ticks_policy"] = "one GPU"
    return full


def test_the_round_5_record_compacts_to_a_line_within_the_target():
    from benchlegs import line
    full = _full()
    assert len(json.dumps(full)) > 16384                       # what broke the driver's parse
    text = line.dumps_checked(line.compact(full, "bench_full.json"))
    assert "\n" not in text and len(text.encode()) <= line.LINE_TARGET
    got = json.loads(text)
    for key in line.CONTRACT + ("config", "roofline", "cpu_baseline", "headline_parity", "legs", "full"):
        assert key in got, key
    for key in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "algorithmic_bytes_per_launch"):
        assert key in got["roofline"], key
    assert set(got["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"}
    assert got["legs"]["video_fps"] > 0 and got["legs"]["fir_f64_frac"] > 0 and got["legs"]["model_speedup_8_fixed_ticks"] > 0


def test_a_nan_or_an_oversized_line_is_refused():
    from benchlegs import line
    full = _full()
    c = line.compact(full, "bench_full.json")
    c["value"] = float("nan")
    with pytest.raises(ValueError):
        line.dumps_checked(c)
    c["value"] = 1.0
    c["config"]["workload"] = "x" * 9000
    with pytest.raises(AssertionError):
        line.dumps_checked(c)
    text = line.dumps_within_cap(c)                           # what bench.py calls: a run never ends without its line -- the optional parts go first
    got = json.loads(text)
    assert len(text) <= line.LINE_HARD_CAP and all(k in got for k in line.CONTRACT + ("roofline", "cpu_baseline", "config")) and got["roofline"] == c["roofline"]


def test_bench_py_stays_a_thin_assembler():
    n = len((ROOT / "bench.py").read_text().splitlines())
    assert n <= 400, f"bench.py has {n} lines: legs belong in benchlegs/"
