"""bench.py's ONE stdout line, as the driver reads it: compact (<= 8 KiB; round 5's 20.8 KB line was the first the driver could not parse), strict JSON, the
contract's keys with consistent values -- value = strips x ticks / step time, `roofline` = the dominant launch group's OWN bytes over its OWN duration, the CPU
baseline of the same workload -- and everything else in the file --full-out names."""
import json
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def _strict(text):
    def bad(c):
        raise ValueError(f"non-strict JSON constant {c}")
    return json.loads(text, parse_constant=bad)


def _run(cmd, tmp_path, timeout=900):
    full = tmp_path / "full.json"
    res = subprocess.run([*cmd, "--full-out", str(full)], capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert res.stdout.endswith("\n") and res.stdout.count("\n") == 1, "stdout carries ONE line and nothing else"
    assert len(res.stdout.encode()) <= 8192, f"the line is {len(res.stdout.encode())} bytes"
    assert len(res.stderr.encode()) <= 8192, f"stderr carries {len(res.stderr.encode())} bytes: a reader that keeps a tail of the merged streams must still find the line\n" + res.stderr[-1500:]
    return _strict(res.stdout), _strict(full.read_text())


def _check_contract(line, strips, ticks, steps, warmup):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"] == "audio_ch_mixed_per_sec" and line["unit"] == "channel-ticks/s"
    assert line["n_gpus"] == 1 and line["steps"] == steps and line["warmup"] == warmup and line["higher_is_better"] is True
    assert line["scaling"] in ("weak", "strong") and line["vs_baseline"] is None and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    # value is the whole job's channel-ticks per second of the timed steps
    assert line["value"] == pytest.approx(strips * ticks * 1000.0 / line["ms_per_step"], rel=1e-6)
    rf = line["roofline"]
    for key in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "algorithmic_bytes_per_launch"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=5e-2, abs=1e-4) and 0.0 < rf["frac"] < 1.0   # (both are printed rounded)
    # ONE kernel group: its own algorithmic bytes over its own average launch duration
    assert rf["frac"] * rf["peak"] * 1e9 * rf["avg_launch_ms"] * 1e-3 == pytest.approx(rf["algorithmic_bytes_per_launch"], rel=2e-2)
    per_sample = {"eq_three": 8, "mixer": 4}[rf["kernel"]]
    assert rf["algorithmic_bytes_per_launch"] == pytest.approx(per_sample * strips * ticks * 800, rel=0.01)   # (the mixer adds its two bus outputs)
    assert rf["avg_launch_ms"] <= line["ms_per_step"] * 1.02
    cb = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == line["unit"]


def test_bench_line_has_the_contract_keys_and_consistent_numbers(tmp_path):
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--strips", "64", "--ticks-per-step", "64",
           "--no-realtime", "--no-t-sweep", "--no-north-star", "--no-held-leg", "--no-material-leg", "--no-scaling-probe", "--fir-ticks", "0", "--repeats", "0", "--video-frames", "0"]
    line, full = _run(cmd, tmp_path)
    _check_contract(line, 64, 64, 3, 1)
    assert full["value"] == line["value"] and full["roofline"]["frac"] == line["roofline"]["frac"]
    assert full["headline_parity"]["verdict"] == "bit-exact" and line["headline_parity"]["verdict"] == "bit-exact"


def test_default_command_line_with_every_leg_on_prints_one_compact_strict_line(tmp_path):
    """The DEFAULT command's legs, all of them, at small sizes (the driver's command is `python bench.py --gpus 1 --steps K --warmup W`): one line, within the cap,
    strict JSON, one number per leg; the full record beside it carries the legs themselves."""
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--strips", "128", "--ticks-per-step", "128", "--fir-ticks", "16",
           "--video-frames", "256", "--repeats", "1"]
    line, full = _run(cmd, tmp_path, timeout=1500)
    _check_contract(line, 128, 128, 4, 1)
    for leg in ("video_fps", "video_hbm_frac", "fir_ch_ticks_per_s", "fir_f64_frac", "fp_contract_value", "realtime_headroom_1024", "north_star_headroom_10240_plus_video",
                "model_speedup_8_scaled_ticks", "model_speedup_8_fixed_ticks", "cpu_all_cores_value"):
        assert leg in line["legs"], leg
    for leg in ("video", "fir_resample", "fp_contract", "realtime", "t_sweep", "group_buses", "rate_44100", "material", "scaling_model", "north_star_realtime", "held_gates"):
        assert full.get(leg), leg
    assert full["video"]["cpu_baseline"]["value"] > 0 and full["fir_resample"]["cpu_baseline"]["value"] > 0
    assert "leg_errors" not in full and "leg_errors" not in line


def test_a_failing_secondary_leg_does_not_cost_the_run_its_line(tmp_path, monkeypatch):
    """A secondary leg that raises (injected here) is reported -- `leg_errors` in the line and the record, the traceback on stderr -- and the headline line still goes out."""
    monkeypatch.setenv("MX_BENCH_FAIL_LEG", "fir_resample")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--strips", "64", "--ticks-per-step", "64", "--no-realtime", "--no-t-sweep",
           "--no-north-star", "--no-held-leg", "--no-material-leg", "--no-scaling-probe", "--no-contract-leg", "--no-buses-leg", "--no-rate-leg", "--fir-ticks", "16", "--repeats", "0",
           "--video-frames", "0", "--no-cpu-baseline"]
    line, full = _run(cmd, tmp_path)
    assert line["leg_errors"] == ["fir_resample"] and "injected" in full["leg_errors"]["fir_resample"] and full["fir_resample"] is None
    assert line["value"] > 0 and line["roofline"]["frac"] > 0 and line["headline_parity"]["verdict"] == "bit-exact"


SMALL = ["--steps", "3", "--warmup", "1", "--strips", "64", "--ticks-per-step", "64", "--no-realtime", "--no-t-sweep", "--no-north-star", "--no-held-leg",
         "--no-material-leg", "--no-scaling-probe", "--no-contract-leg", "--fir-ticks", "0", "--repeats", "0", "--video-frames", "0", "--no-cpu-baseline"]


@pytest.mark.parametrize("mode", ["allgather", "slices"])
def test_bench_exchange_path_checks_its_own_parity_single_rank_rccl(mode, tmp_path):
    """The N > 1 path of bench.py at N = 1 (--force-combine: a single-rank RCCL communicator): the record carries the parity evidence the first multi-GPU lease will
    produce with no new code -- the exchange's bus against a host sum, in rank order, of the partial buses gathered by a plain all_gather -- and the second tick
    policy beside the headline's."""
    line, full = _run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--force-combine", "--exchange", mode, *SMALL], tmp_path)
    ex = full["exchange"]
    assert ex["mode"] == mode and ex["rccl_ranks"] == 1
    assert ex["parity_check"]["verdict"] == "bit-exact" and ex["parity_check"]["samples_compared"] == 2 * 64 * 1600
    st = full["other_policy"]
    assert st["ticks_per_step"] == 64 and st["parity"]["verdict"] == "bit-exact" and st["value"] > 0
    assert line["legs"]["exchange_parity"] == "bit-exact" and line["legs"]["other_ticks_policy_ticks"] == 64
    assert line["headline_parity"]["verdict"] == "bit-exact" and line["headline_parity"]["buses"] == "bit-exact"      # (replayed up to the last submission the graph ran: the 3 profiled extra steps of the N > 1 path included)


def test_two_rank_rccl_job_is_bit_exact_when_two_gpus_are_visible(tmp_path):
    """configs[4] over real RCCL peers: skipped on a one-GPU box, turn-key on anything larger -- the driver's own launch line (python -m torch.distributed.run
    --nproc-per-node 2 ... bench.py --gpus 2) at a small size, both ordered exchange modes.  The headline's policy is T x N ticks per step (config.ticks_policy),
    the fixed-T policy runs beside it."""
    sys.path.insert(0, str(ROOT))
    from mixlab_amd import abi
    if abi.lib.mx_device_count() < 2:
        pytest.skip("needs two GPUs")
    for k, mode in enumerate(("allgather", "slices")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(29571 + k),
               str(ROOT / "bench.py"), "--gpus", "2", "--exchange", mode, *SMALL]
        line, full = _run(cmd, tmp_path)
        assert line["n_gpus"] == 2 and full["exchange"]["rccl_ranks"] == 2 and full["exchange"]["mode"] == mode
        pc = full["exchange"]["parity_check"]
        assert pc["verdict"] == "bit-exact" and pc["all_ranks"] == "bit-exact", pc
        assert line["config"]["ticks_per_step"] == 128 and "T x N" in line["config"]["ticks_policy"]
        assert full["other_policy"]["ticks_per_step"] == 64 and full["other_policy"]["parity"]["verdict"] == "bit-exact"
        assert line["headline_parity"]["verdict"] == "bit-exact"


def _multi_rank_on_one_gpu(world, extra, tmp_path, small=None):
    """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` with N REAL ranks on a one-GPU box: every rank on GPU 0 (MX_BENCH_SHARE_GPU), torch.distributed
    on gloo (barriers, the max of the clock, the plain all_gather of the parity check), the library's exchange on the RCCL test double (MX_RCCL_LIB, tests/helpers/fake_rccl.c)."""
    import os
    so = tmp_path / "libfake_rccl.so"
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", str(so), str(ROOT / "tests" / "helpers" / "fake_rccl.c"),
                    "-L/opt/rocm/lib", "-lamdhip64", "-lrt"], check=True)
    full = tmp_path / "full.json"
    env = dict(os.environ, MX_BENCH_SHARE_GPU="1", MX_BENCH_DIST_BACKEND="gloo", MX_RCCL_LIB=str(so))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(29581 + world + 16 * len(extra)),
           str(ROOT / "bench.py"), "--gpus", str(world), *(small or SMALL), *extra, "--full-out", str(full)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= 8192
    return _strict(lines[0]), _strict(full.read_text())


@pytest.mark.parametrize("world,mode,vshard", [(2, "allgather", "replicas"), (4, "slices", "bands"), (8, "auto", "replicas")])
def test_the_drivers_multi_rank_launch_line_on_one_gpu_through_the_rccl_double(world, mode, vshard, tmp_path):
    """Times mean nothing here; what is checked is the N > 1 code of bench.py and of mx_exchange.cpp as the driver will run it: the T x N tick policy as the headline, the
    fixed-T policy beside it, both exchanges bit-exact on every rank against the host sum in rank order, the oracle replay of the last submission, the video leg."""
    small = list(SMALL)
    small[small.index("--video-frames") + 1] = "64"        # the video leg runs here too: replicas per rank, or ONE stream in row bands
    line, rec = _multi_rank_on_one_gpu(world, ["--exchange", mode, "--video-shard", vshard], tmp_path, small)
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["cpu_baseline"] is None
    assert line["config"]["ticks_per_step"] == 64 * world and "T x N" in line["config"]["ticks_policy"]
    assert line["value"] == pytest.approx(64 * 64 * world * 1000.0 / line["ms_per_step"], rel=1e-6)
    ex = rec["exchange"]
    assert ex["mode"] == (mode if mode != "auto" else "slices") and ex["rccl_ranks"] == world
    assert ex["parity_check"]["verdict"] == "bit-exact" and ex["parity_check"]["all_ranks"] == "bit-exact", ex
    assert ex["parity_check"]["samples_compared"] == 2 * 64 * world * 1600
    op = rec["other_policy"]
    assert op["ticks_per_step"] == 64 and op["parity"]["verdict"] == "bit-exact" and op["exchange_mode"] == ex["mode"]
    assert rec["roofline"]["kernel_timing"].startswith("hipEvents on 3 extra steps")
    assert line["headline_parity"]["verdict"] == "bit-exact" and line["headline_parity"]["buses"] == "bit-exact", rec["headline_parity"]
    v = rec["video"]
    assert v["value"] > 0 and v["scaling"] == ("weak" if vshard == "replicas" else "strong") and v["frames"] == (64 * world if vshard == "replicas" else 64)


@pytest.mark.parametrize("world,extra", [(2, ["--exchange", "allreduce"]), (2, ["--fixed-ticks"]), (4, ["--fp-contract"]), (3, ["--strips", "96"]), (2, ["--hold-gates"])],
                         ids=["allreduce", "fixed-ticks", "fp-contract", "three-ranks", "hold-gates"])
def test_flags_of_the_multi_rank_path_that_no_other_test_drives(world, extra, tmp_path):
    """--exchange allreduce (the non-parity collective: its distance from the ordered sum is measured against a second, ordered exchange made on the fly), --fixed-ticks (the
    policies swap places), the contracted order, a rank count that is not a power of two, held gates."""
    line, rec = _multi_rank_on_one_gpu(world, extra, tmp_path)
    assert line["n_gpus"] == world and line["headline_parity"]["verdict"] == "bit-exact" and line["headline_parity"]["buses"] == "bit-exact"
    ex = rec["exchange"]
    if "allreduce" in extra:
        assert ex["mode"] == "allreduce" and "parity_check" not in ex and ex["max_ulp_vs_ordered_sum"] == 0     # (the double's all-reduce IS the rank-ordered sum; real RCCL's is not)
        assert rec["other_policy"]["parity"] is None
    else:
        assert ex["parity_check"]["all_ranks"] == "bit-exact" and rec["other_policy"]["parity"]["verdict"] == "bit-exact"
    fixed = "--fixed-ticks" in extra
    assert line["config"]["ticks_per_step"] == (64 if fixed else 64 * world) and rec["other_policy"]["ticks_per_step"] == (64 * world if fixed else 64)
