"""Config 2 with GROUP BUSES: 1024 strips -> G Mixers of 1024 / G -> a master Mixer(G), T ticks per submission, gates toggling; the second-stream mode (the tail is the
bank AND the master above it) against one stream.   python tools/q_buses.py [groups] [ticks]"""
import os, sys, time, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import synth
from bench import build_strips, gate_events
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
N, SR, spt = 1024, 48000, 800
for auto in (1, 0):
    if auto: os.environ.pop("MX_OVERLAP_AUTO", None)
    else: os.environ["MX_OVERLAP_AUTO"] = "0"
    ws = Workspace(SR, 60); mixes, srcs, trigs = [], [], []
    for j in range(G):
        ws, m, s, t = build_strips(abi, Workspace, synth, N // G, j * (N // G), SR, ws=ws, total=N, want_trigs=True)
        mixes.append(m); srcs += s; trigs += t
    master = ws.mixer([(0.0, 1.0, False)] * G)
    for j, m in enumerate(mixes):
        ws.connect(m, 0, master, j)
    g = ws.build(max_ticks_per_run=T)
    base = min(T, 256)
    for j, s in enumerate(srcs):
        g.write_source(s, np.tile(synth.noise(j, base * spt), (T + base - 1) // base)[: T * spt], T)
    K = 12
    ev = [gate_events(abi, trigs, 0, i * T, T) for i in range(K + 2)]
    for i in range(2):
        g.schedule_params_batch(ev[i][0], ev[i][1]); g.run_ticks(i * T, T)
    g.sync(); t0 = time.perf_counter()
    for i in range(K):
        g.schedule_params_batch(ev[2 + i][0], ev[2 + i][1]); g.run_ticks((2 + i) * T, T)
    g.sync(); dt = (time.perf_counter() - t0) / K
    print(f"{G} group buses, T = {T}: {'second stream' if g.tail_stream() is not None else 'one stream'}: {dt * 1e3:.3f} ms per run = {N * T / dt / 1e6:.1f} M channel-ticks/s; releases {g.debug_tail_releases()}", flush=True)
    g.close()
