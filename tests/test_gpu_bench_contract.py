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


SMALL = ["--steps", "3", "--warmup", "1", "--strips", "64", "--ticks-per-step", "64", "--no-realtime", "--no-t-sweep", "--no-north-star", "--no-held-leg",
         "--no-material-leg", "--no-scaling-probe", "--no-contract-leg", "--fir-ticks", "0", "--repeats", "0", "--video-frames", "0", "--no-cpu-baseline"]


def _line(res):
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["allgather", "slices"])
def test_bench_exchange_path_checks_its_own_parity_single_rank_rccl(mode):
    """The N > 1 path of bench.py at N = 1 (--force-combine: a single-rank RCCL communicator): the line carries the parity evidence the
    first multi-GPU lease will produce with no new code -- the exchange's bus against a host sum, in rank order, of the partial buses
    gathered by a plain all_gather -- and the second tick policy beside the headline's."""
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--force-combine", "--exchange", mode, *SMALL],
                         capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    line = _line(res)
    ex = line["exchange"]
    assert ex["mode"] == mode and ex["rccl_ranks"] == 1
    assert ex["parity_check"]["verdict"] == "bit-exact" and ex["parity_check"]["samples_compared"] == 2 * 64 * 1600
    st = line["scaled_ticks"]
    assert st["ticks_per_step"] == 64 and st["parity"]["verdict"] == "bit-exact" and st["value"] > 0


def test_two_rank_rccl_job_is_bit_exact_when_two_gpus_are_visible():
    """configs[4] over real RCCL peers: skipped on a one-GPU box, turn-key on anything larger -- the driver's own launch line
    (python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2) at a small size, both ordered exchange modes."""
    sys.path.insert(0, str(ROOT))
    from mixlab_amd import abi
    if abi.lib.mx_device_count() < 2:
        pytest.skip("needs two GPUs")
    for k, mode in enumerate(("allgather", "slices")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(29571 + k),
               str(ROOT / "bench.py"), "--gpus", "2", "--exchange", mode, *SMALL]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
        line = _line(res)
        assert line["n_gpus"] == 2 and line["exchange"]["rccl_ranks"] == 2 and line["exchange"]["mode"] == mode
        pc = line["exchange"]["parity_check"]
        assert pc["verdict"] == "bit-exact" and pc["all_ranks"] == "bit-exact", pc
        assert line["scaled_ticks"]["ticks_per_step"] == 128 and line["scaled_ticks"]["parity"]["verdict"] == "bit-exact"
