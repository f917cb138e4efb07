"""A live topology edit on the 1024-strip graph (Engine::client_update, src/engine.rs:277-398): time to freeze the edited workspace
(mx_graph_build: off the tick thread, while the old graph keeps running) and to take over the surviving modules' state
(mx_graph_adopt_state: on the tick thread, between two ticks)."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
import synth
import bench
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace

SR, N = 48000, 1024
ws, mix, srcs, trigs = bench.build_strips(abi, Workspace, synth, N, 0, SR, want_trigs=True)
t0 = time.perf_counter(); g = ws.build(max_ticks_per_run=1); t_build = time.perf_counter() - t0
for s in srcs:
    g.write_source(s, np.zeros(SR // 60, np.float32), 1)
for k in range(5):
    g.run_ticks(k, 1)
g.sync()
# the edit: one more module behind the master bus (a new node at the end; every old node keeps its index)
ws2, mix2, srcs2, trigs2 = bench.build_strips(abi, Workspace, synth, N, 0, SR, want_trigs=True)
amp = ws2.amplifier(0.5, 0.0); ws2.connect(mix2, 0, amp, 0)
t0 = time.perf_counter(); g2 = ws2.build(max_ticks_per_run=1); t_build2 = time.perf_counter() - t0
n_old = len(ws.nodes)
mapping = list(range(n_old)) + [-1]
t0 = time.perf_counter(); g2.adopt_state(g, mapping); g2.sync(); t_adopt = time.perf_counter() - t0
for s in srcs2:
    g2.write_source(s, np.zeros(SR // 60, np.float32), 1)
t0 = time.perf_counter(); g2.run_ticks(5, 1); g2.sync(); t_tick = time.perf_counter() - t0
print(f"{N}-strip graph ({len(ws.nodes)} modules): mx_graph_build {t_build * 1e3:.1f} ms (first), {t_build2 * 1e3:.1f} ms (edited workspace, off the tick thread); "
      f"mx_graph_adopt_state {t_adopt * 1e3:.2f} ms (between two ticks; budget 16.7 ms); first tick of the new graph {t_tick * 1e3:.2f} ms")
