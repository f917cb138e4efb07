"""configs[4]'s exchange over the library's RCCL transport with W = 2 and W = 4 REAL peers -- processes sharing the one GPU of the box -- through a test double of RCCL
(tests/helpers/fake_rccl.c, bound with MX_RCCL_LIB): the product's own collective_rccl code (mixlab_amd/csrc/mx_exchange.cpp: the all-gather of the whole partial buses;
the grouped ncclSend / ncclRecv of the time slices with their strides q * 2 * Lp_ and n_flp_, the two all-gathers of the finished slices) runs with more than one rank,
which the single-rank RCCL tests (tests/test_gpu_exchange.py) and the in-process loopback transport (tests/test_gpu_config5_sharded.py: it REPLACES those lines) never
did.  Every rank's combined Master / Cue must be the oracle's run of  W x Mixer(128) -> Mixer(W, unity), bit for bit (src/module/mixer.rs:57-68: channel after channel)."""
import os
import pathlib
import subprocess
import sys
import uuid

import numpy as np
import pytest

from test_gpu_audio_parity import bits
from test_gpu_config5_sharded import SPT, STEPS, T, hierarchical_oracle

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    so = tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so"
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", str(so), str(ROOT / "tests" / "helpers" / "fake_rccl.c"),
                    "-L/opt/rocm/lib", "-lamdhip64", "-lrt"], check=True)
    return so


@pytest.mark.parametrize("world,mode", [(2, "allgather"), (2, "slices"), (4, "allgather"), (4, "slices"), (4, "auto")])
def test_ranks_in_separate_processes_through_collective_rccl_equal_the_hierarchical_oracle(world, mode, fake_rccl, tmp_path):
    want_m, want_c, _noise = hierarchical_oracle(world)
    nccl_id = (b"/fake_rccl_test_" + uuid.uuid4().hex.encode()).ljust(128, b"\0")       # the double's ncclUniqueId is the name of its shared-memory segment
    env = dict(os.environ, MX_RCCL_LIB=str(fake_rccl))
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "helpers" / "fake_rccl_rank.py"), str(r), str(world), mode, nccl_id.hex(), str(tmp_path / f"rank{r}.npz")],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} of {world} ok" in so, f"rank {r}: {so[-1000:]}{se[-3000:]}"
    bus = 2 * 2 * SPT * T * 4
    sliced = mode == "slices" or (mode == "auto" and world >= 4)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert int(z["bytes_received_per_step"][0]) == (2 * (world - 1) * bus // world if sliced else (world - 1) * bus)
        for i in range(STEPS):
            sl = slice(i * T * 2 * SPT, (i + 1) * T * 2 * SPT)
            assert np.array_equal(bits(z[f"m{i}"]), bits(want_m[sl])), f"rank {r} of {world}, {mode}: Master of step {i}"
            assert np.array_equal(bits(z[f"c{i}"]), bits(want_c[sl])), f"rank {r} of {world}, {mode}: Cue of step {i}"
        assert not np.array_equal(bits(z["partial_m"]), bits(want_m[(STEPS - 1) * T * 2 * SPT:]))      # a rank's partial bus alone is NOT the whole bus
