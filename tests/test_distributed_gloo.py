"""The N > 1 plan (mixlab_amd/shard.py) on CPU: two gloo processes, each computing the partial bus
of its strip shard, one all-gather, rank-ordered combine -- against the single-process
hierarchical graph  2 x Mixer(n/2) -> Mixer(2, unity)  that defines the sharded semantics.

No GPU here, so the per-rank compute is done by the CPU oracle (the checker doing the checker's
job); what is under test is the sharding, the packed all-gather layout and the combine order that
bench.py uses verbatim on RCCL."""
import os
import pathlib
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent

WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import numpy as np, torch, torch.distributed as dist
import oracle, synth
from mixlab_amd import shard
from mixlab_amd.workspace import Workspace
from mixlab_amd import abi

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
N, T, SPT = 12, 3, 735
first, count = shard.strip_range(rank, world, N)
gains = synth.uniform(11, N, -24.0, 6.0); faders = synth.uniform(12, N, 0.0, 1.0)

def strip_ws(lo, n):
    ws = Workspace(44100, 60)
    mix = ws.mixer([(float(gains[k]), float(faders[k]), k % 3 == 0) for k in range(lo, lo + n)])
    srcs = []
    for j in range(n):
        s = ws.source_mono(); e = ws.eq_three(1.0 * ((lo + j) % 5) - 2.0, 0.0, 3.0); p = ws.stereo_panner()
        ws.connect(s, 0, e, 0); ws.connect(e, 0, p, 0); ws.connect(e, 0, p, 1); ws.connect(p, 0, mix, j)
        srcs.append(s)
    return ws, mix, srcs

ws, mix, srcs = strip_ws(first, count)
og = oracle.OracleGraph(ws)
n_fl = 2 * SPT * T
part_len, offs = shard.packed_layout(world, n_fl)
part = torch.zeros(part_len, dtype=torch.float32)
for t in range(T):
    for j, s in enumerate(srcs):
        og.set_source(s, synth.noise(first + j, T * SPT)[t * SPT:(t + 1) * SPT])
    og.run_tick(t)
    part[t * 2 * SPT:(t + 1) * 2 * SPT] = torch.from_numpy(og.output(mix, 0))
    part[n_fl + t * 2 * SPT: n_fl + (t + 1) * 2 * SPT] = torch.from_numpy(og.output(mix, 1))
chans = shard.combine_channels(world)
if os.environ.get("EXCHANGE") == "slices":
    # ordered reduce-scatter + all-gather: all-to-all of time slices, rank-ordered sum of the own slice, all-gather of the results
    L, soffs = shard.slice_layout(world, n_fl)
    send = torch.empty(world * 2 * L, dtype=torch.float32)
    send.view(world, 2, L).copy_(shard.pack_slices(part, world))
    recv = torch.zeros(world * 2 * L, dtype=torch.float32)
    dist.all_to_all_single(recv, send)
    rv = recv.numpy()
    m_mine, _ = oracle.mixer_run(chans, [rv[soffs[r][0]:soffs[r][0] + L] for r in range(world)], L)
    c_mine, _ = oracle.mixer_run(chans, [rv[soffs[r][1]:soffs[r][1] + L] for r in range(world)], L)
    fin = torch.from_numpy(np.concatenate([m_mine, c_mine]))
    final_all = torch.zeros(world * 2 * L, dtype=torch.float32)
    dist.all_gather_into_tensor(final_all, fin)
    m_t, c_t = shard.unpack_slices(final_all, world)
    master, cue = m_t.numpy().copy(), c_t.numpy().copy()
else:
    gathered = torch.zeros(world * part_len, dtype=torch.float32)
    dist.all_gather_into_tensor(gathered, part)
    g = gathered.numpy()
    master, _ = oracle.mixer_run(chans, [g[offs[r][0]:offs[r][0] + n_fl] for r in range(world)], n_fl)
    cue, _ = oracle.mixer_run(chans, [g[offs[r][1]:offs[r][1] + n_fl] for r in range(world)], n_fl)
np.save(os.environ["OUT_DIR"] + f"/rank{{rank}}.npy", np.stack([master, cue]))
dist.barrier(); dist.destroy_process_group()
"""


def reference_hierarchy(N=12, T=3, SPT=735, world=2):
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle
    import synth
    from mixlab_amd import shard
    from mixlab_amd.workspace import Workspace

    gains = synth.uniform(11, N, -24.0, 6.0); faders = synth.uniform(12, N, 0.0, 1.0)
    ws = Workspace(44100, 60)
    final_m = ws.mixer(shard.combine_channels(world))
    final_c = ws.mixer(shard.combine_channels(world))
    srcs = {}
    for r in range(world):
        lo, n = shard.strip_range(r, world, N)
        sub = ws.mixer([(float(gains[k]), float(faders[k]), k % 3 == 0) for k in range(lo, lo + n)])
        for j in range(n):
            s = ws.source_mono(); e = ws.eq_three(1.0 * ((lo + j) % 5) - 2.0, 0.0, 3.0); p = ws.stereo_panner()
            ws.connect(s, 0, e, 0); ws.connect(e, 0, p, 0); ws.connect(e, 0, p, 1); ws.connect(p, 0, sub, j)
            srcs[lo + j] = s
        ws.connect(sub, 0, final_m, r); ws.connect(sub, 1, final_c, r)
    og = oracle.OracleGraph(ws)
    m, c = [], []
    for t in range(T):
        for k, s in srcs.items():
            og.set_source(s, synth.noise(k, T * SPT)[t * SPT:(t + 1) * SPT])
        og.run_tick(t)
        m.append(og.output(final_m, 0)); c.append(og.output(final_c, 0))
    return np.concatenate(m), np.concatenate(c)


@pytest.mark.parametrize("exchange,port", [("allgather", "29611"), ("slices", "29612")])
def test_two_rank_gloo_exchange_and_ordered_combine_equal_hierarchical_graph(exchange, port):
    with tempfile.TemporaryDirectory() as td:
        script = pathlib.Path(td) / "worker.py"
        script.write_text(WORKER.format(root=str(ROOT)))
        env = dict(os.environ, OUT_DIR=td, MASTER_ADDR="127.0.0.1", EXCHANGE=exchange)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", port, str(script)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        want_m, want_c = reference_hierarchy()
        for r in range(2):
            got = np.load(pathlib.Path(td) / f"rank{r}.npy")
            assert np.array_equal(got[0].view(np.uint32), want_m.view(np.uint32)), f"rank {r}: master differs"
            assert np.array_equal(got[1].view(np.uint32), want_c.view(np.uint32)), f"rank {r}: cue differs"


def test_shard_plan():
    from mixlab_amd import shard
    assert [shard.strip_range(r, 8, 1024) for r in (0, 7)] == [(0, 128), (896, 128)]
    with pytest.raises(ValueError):
        shard.strip_range(0, 3, 1024)
    part, offs = shard.packed_layout(2, 100)
    assert part == 200 and offs == [(0, 100), (200, 300)]
    L, soffs = shard.slice_layout(4, 100)
    assert L == 25 and soffs[1] == (50, 75)
    with pytest.raises(ValueError):
        shard.slice_layout(3, 100)
    import torch
    mc = torch.arange(16, dtype=torch.float32)                 # master 0..7 | cue 8..15, 2 ranks -> slices of 4
    send = torch.empty(16); send.view(2, 2, 4).copy_(shard.pack_slices(mc, 2))
    assert send.tolist() == [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]
    m, c = shard.unpack_slices(send, 2)                        # the same layout comes back from the all-gather
    assert m.tolist() == list(range(8)) and c.tolist() == list(range(8, 16))
