"""Host time of one submission (1024 strips x T ticks, gates toggling): mx_graph_schedule_params_batch and mx_graph_run_ticks separately, device idle in between (sync)."""
import sys, time, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import synth
from bench import build_strips, gate_events
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ws, mix, srcs, trigs = build_strips(abi, Workspace, synth, 1024, 0, 48000, want_trigs=True)
g = ws.build(max_ticks_per_run=T)
for j, s in enumerate(srcs):
    g.write_source(s, np.tile(synth.noise(j, min(T, 256) * 800), (T + 255) // 256)[: T * 800], T)
a, b, n_ev = [], [], []
for i in range(40):
    ev = gate_events(abi, trigs, 0, i * T, T)
    g.sync()
    t0 = time.perf_counter()
    if ev: g.schedule_params_batch(ev[0], ev[1])
    t1 = time.perf_counter()
    g.run_ticks(i * T, T)
    t2 = time.perf_counter()
    a.append((t1 - t0) * 1e6); b.append((t2 - t1) * 1e6); n_ev.append(ev[1] if ev else 0)
print(f"T={T}: events per submission ~{int(np.median(n_ev))}; schedule_params_batch median {np.median(a[5:]):.0f} us, run_ticks median {np.median(b[5:]):.0f} us (idle queue)")
