"""mixlab_amd/exchange.py on the GPU with a single-rank RCCL group (what one rank of the 8-GPU job runs, collectives
included): the combined bus of every exchange mode must be the local Mixer's output bit for bit (Mixer(1, unity) of one
partial is that partial), across pipelined steps that reuse the two slots.  The 2-rank layouts and the rank-ordered combine
are covered on CPU by tests/test_distributed_gloo.py.

Runs in a child process: torch brings its own HIP runtime and must be imported BEFORE libmixlab_gpu.so is loaded (as bench.py
does); inside the pytest process the library is already loaded by the time this module is collected."""
import os
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent

WORKER = r"""
import os, sys
import torch, torch.distributed as dist          # first: one HIP runtime in the process
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import numpy as np
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace
from mixlab_amd.exchange import BusExchange

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

def strips(n, sr=48000):
    ws = Workspace(sr, 60)
    mix = ws.mixer([(-1.5 * k, 1.0 - 0.05 * k, k % 3 == 0) for k in range(n)])
    for k in range(n):
        osc = ws.oscillator(110.0 * (k + 1), abi.WAVE_SAW if k % 2 else abi.WAVE_TRIANGLE)
        eq = ws.eq_three(3.0 - k, 0.5 * k, -2.0 + k); pan = ws.stereo_panner()
        ws.connect(osc, 0, eq, 0); ws.connect(eq, 0, pan, 0); ws.connect(eq, 0, pan, 1); ws.connect(pan, 0, mix, k)
    return ws, mix

T, n = 8, 6
for mode in ("allgather", "slices", "allreduce"):
    ws, mix = strips(n)
    stream = torch.cuda.Stream()
    g = ws.build(max_ticks_per_run=T, device=0, stream=stream.cuda_stream)
    ref = ws.build(max_ticks_per_run=T)                    # the same shard, run on its own: what the bus must be
    ex = BusExchange(torch, dist, g, mix, T, 48000, 0, stream, mode=mode)
    assert ex.world == 1 and ex.mode == mode
    local_out = []
    for i in range(4):
        ref.run_ticks(i * T, T)
        local_out.append((ref.read_output(mix, 0, T, True), ref.read_output(mix, 1, T, True)))
    assert not np.array_equal(local_out[0][0], local_out[1][0])        # the steps really differ (oscillators run on absolute time)

    def check(i):
        ex.wait(i)
        m, c = ex.result(i)
        torch.cuda.synchronize()
        wm, wc = local_out[i]
        assert np.array_equal(m.cpu().numpy().view(np.uint32), wm.view(np.uint32)), f"{{mode}}: master of step {{i}}"
        assert np.array_equal(c.cpu().numpy().view(np.uint32), wc.view(np.uint32)), f"{{mode}}: cue of step {{i}}"

    # steps 0 and 1 in flight together (two slots), then step 2 reuses slot 0, step 3 slot 1
    g.run_ticks(0, T); ex.submit(0)
    g.run_ticks(T, T); ex.submit(1)
    check(0)
    g.run_ticks(2 * T, T); ex.submit(2)
    check(1)
    g.run_ticks(3 * T, T); ex.submit(3)
    check(2); check(3)
    assert ex.max_ulp_vs(3, *[torch.from_numpy(a).cuda() for a in local_out[3]]) == 0
    assert ex.bytes_received_per_step() == 0                           # a single rank receives nothing
    ex.close()
    print("ok", mode, flush=True)

ws, mix = strips(2)
stream = torch.cuda.Stream()
g = ws.build(max_ticks_per_run=4, device=0, stream=stream.cuda_stream)
try:
    BusExchange(torch, dist, g, mix, 4, 48000, 0, stream, mode="bogus")
    raise SystemExit("a bogus mode was accepted")
except ValueError:
    print("ok bogus-mode", flush=True)
dist.destroy_process_group()
"""


def test_single_rank_rccl_exchange_returns_the_local_bus_over_pipelined_steps(tmp_path):
    script = tmp_path / "exchange_worker.py"
    script.write_text(WORKER.format(root=str(ROOT)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    for mode in ("allgather", "slices", "allreduce", "bogus-mode"):
        assert f"ok {mode}" in res.stdout, res.stdout[-2000:]
