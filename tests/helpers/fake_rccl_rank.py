"""One rank of a W-process exchange test (tests/test_gpu_exchange_processes.py): this process's shard of the config-5 audio job on GPU 0, its bus exchange over the RCCL
transport of libmixlab_gpu.so -- bound to the test double tests/helpers/fake_rccl.c through MX_RCCL_LIB -- and the combined buses of every step written to an .npz.
usage: fake_rccl_rank.py <rank> <world> <mode> <id-hex> <out.npz>     (MX_RCCL_LIB in the environment)"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import synth  # noqa: E402
from mixlab_amd import shard  # noqa: E402
from mixlab_amd.exchange import BusExchange  # noqa: E402
from mixlab_amd.workspace import Workspace  # noqa: E402
from test_gpu_config5_sharded import PER_RANK, SPT, SR, STEPS, T, add_strips, schedule_gates  # noqa: E402

rank, world, mode, nccl_id, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], bytes.fromhex(sys.argv[4]), sys.argv[5]
total = world * PER_RANK
lo, n = shard.strip_range(rank, world, total)
ws = Workspace(SR, 60)
mix, srcs, trigs = add_strips(ws, lo, n, total)
g = ws.build(max_ticks_per_run=T)
ex = BusExchange(g, mix, T, rank, world, mode=mode, nccl_id=nccl_id)
assert ex.world == world and ex.mode == (mode if mode != "auto" else ("slices" if world >= 4 and T % world == 0 else "allgather"))
noise = [synth.noise(lo + j, STEPS * T * SPT) for j in range(n)]
res = {}
for i in range(STEPS):           # every rank submits the same steps in the same order; a step's result is read one step later (two slots in flight)
    for j, s in enumerate(srcs):
        g.write_source(s, noise[j][i * T * SPT:(i + 1) * T * SPT], T)
    schedule_gates(g, trigs, lo, i * T)
    g.run_ticks(i * T, T)
    ex.submit(i)
    if i >= 1:
        res[f"m{i - 1}"], res[f"c{i - 1}"] = ex.result(i - 1)
res[f"m{STEPS - 1}"], res[f"c{STEPS - 1}"] = ex.result(STEPS - 1)
res["bytes_received_per_step"] = np.array([ex.bytes_received_per_step()])
res["partial_m"] = g.read_output(mix, 0, T, True)
np.savez(out, **res)
ex.close(); g.close()
print(f"rank {rank} of {world} ok")
