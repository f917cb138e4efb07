"""include/mixlab_gpu.h is a C header: compile a plain-C client with gcc (-std=c11 -Wall -Werror -pedantic)
against libmixlab_gpu.so.  On a GPU box the client also runs."""
import pathlib
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


def build(tmp_path):
    exe = tmp_path / "abi_smoke"
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic", "-I", str(ROOT / "include"), str(ROOT / "tests" / "c" / "abi_smoke.c"),
           "-L", str(ROOT / "mixlab_amd"), "-lmixlab_gpu", f"-Wl,-rpath,{ROOT / 'mixlab_amd'}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe), "-lm"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_plain_c_client_compiles_links_and_reports_missing_gpu_cleanly(tmp_path):
    exe = build(tmp_path)
    res = subprocess.run([str(exe)], capture_output=True, text=True)
    assert res.returncode in (0, 2), res.stdout + res.stderr      # 2 = "no GPU", said so on stderr, no crash
    if res.returncode == 2:
        assert "no GPU" in res.stderr


@pytest.mark.gpu
def test_plain_c_client_runs_on_gpu(tmp_path):
    exe = build(tmp_path)
    res = subprocess.run([str(exe)], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "abi_smoke ok" in res.stdout


def build_cpp(tmp_path):
    exe = tmp_path / "host_mirror"
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-I", str(ROOT / "include"), str(ROOT / "tests" / "cpp" / "host_mirror.cpp"),
           "-L", str(ROOT / "mixlab_amd"), "-lmixlab_gpu", f"-Wl,-rpath,{ROOT / 'mixlab_amd'}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_cpp_host_mirror_compiles_links_and_reports_missing_gpu_cleanly(tmp_path):
    # include/mixlab_gpu.hpp: ModuleT / InputRef / OutputRef / Workspace / Engine with the reference's names over the C ABI
    exe = build_cpp(tmp_path)
    res = subprocess.run([str(exe)], capture_output=True, text=True)
    assert res.returncode in (0, 2), res.stdout + res.stderr
    if res.returncode == 2:
        assert "no GPU" in res.stderr


@pytest.mark.gpu
def test_cpp_host_mirror_runs_on_gpu(tmp_path):
    exe = build_cpp(tmp_path)
    res = subprocess.run([str(exe)], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "host_mirror ok" in res.stdout
