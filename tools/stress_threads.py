"""Two threads on one device: the tick thread runs a video + audio graph and checks every submission against the oracle while a second
thread keeps freezing edited workspaces (mx_graph_build), running them once and destroying them -- what a live edit does off the tick
thread (Engine::client_update, src/engine.rs:277-398) -- plus stand-alone scaler / frame traffic.  Usage: python tools/stress_threads.py [seconds]"""
import sys, pathlib, threading, time, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle, oracle_video as ov, synth
from mixlab_amd import abi, video
from mixlab_amd.workspace import Workspace
from test_gpu_audio_parity import strips

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
SR, SPT, T = 44100, 735, 4
errors, stop = [], threading.Event()
counts = {"ticks": 0, "builds": 0}


def builder():
    rng = np.random.default_rng(1)
    try:
        while not stop.is_set():
            n = int(rng.integers(1, 40))
            ws, mix, srcs, trigs = strips(n, SR)
            sv = ws.source_video(); vm = ws.video_mixer(a=0, b=None, fader=0.5); ws.connect(sv, 0, vm, 0)
            g = ws.build(max_ticks_per_run=2)
            d = video.DFrame(64, 48); video.graph_set_video_source(g, sv, d, repeat=True)
            for s in srcs:
                g.write_source(s, np.zeros(2 * SPT, np.float32), 2)
            g.run_ticks(0, 2); g.sync()
            sc = video.Scaler(int(rng.integers(8, 100)) * 2, int(rng.integers(8, 100)) * 2); sc.scale(d)
            g.close() if hasattr(g, "close") else None
            counts["builds"] += 1
    except Exception as e:
        errors.append(("builder", e, traceback.format_exc()))


def ticker():
    try:
        ws, mix, srcs, trigs = strips(12, SR)
        og = oracle.OracleGraph(ws)          # the audio part (the oracle's graph runner has no video kinds); the video nodes come after, ids unchanged
        sa, sb = ws.source_video(), ws.source_video()
        vm = ws.video_mixer(a=0, b=1, fader=0.3); ws.connect(sa, 0, vm, 0); ws.connect(sb, 0, vm, 1)
        g = ws.build(max_ticks_per_run=T)
        omx = ov.OracleVideoMixer(a=0, b=1, fader=0.3)
        la, lb = ov.HostFrame(160, 90).fill(1, seed=1), ov.HostFrame(96, 54).fill(2, seed=2)
        da, db = video.DFrame(160, 90).upload(*la.visible()), video.DFrame(96, 54).upload(*lb.visible())
        video.graph_set_video_source(g, sa, da, repeat=True); video.graph_set_video_source(g, sb, db, repeat=True)
        run = 0
        while not stop.is_set():
            noise = [synth.noise((run * 12 + k) % 50000, T * SPT) for k in range(12)]
            for k, s in enumerate(srcs):
                g.write_source(s, noise[k], T)
            g.run_ticks(run * T, T)
            got = g.read_output(mix, 0, T, True)
            for kk in range(T):
                for k, s in enumerate(srcs):
                    og.set_source(s, noise[k][kk * SPT:(kk + 1) * SPT])
                og.run_tick(run * T + kk)
                sl = slice(kk * 2 * SPT, (kk + 1) * 2 * SPT)
                if not np.array_equal(np.asarray(got[sl]).view(np.uint32), np.asarray(og.output(mix, 0), np.float32).view(np.uint32)):
                    raise AssertionError(f"run {run} tick {kk}: master differs while another thread builds graphs")
                want = omx.run_tick((run * T + kk) * SPT, [(la, (1, 60), (0, 1)), (lb, (1, 60), (0, 1)), None, None])
            prog = video.graph_video_output(g, vm, 0)
            for a, b in zip(prog.download(), want.visible()):
                if not np.array_equal(a, b):
                    raise AssertionError(f"run {run}: program frame differs")
            run += 1; counts["ticks"] += T
    except Exception as e:
        errors.append(("ticker", e, traceback.format_exc()))


ts = [threading.Thread(target=builder), threading.Thread(target=builder), threading.Thread(target=ticker)]
for t in ts:
    t.start()
time.sleep(secs); stop.set()
for t in ts:
    t.join()
for who, e, tb in errors:
    print(who, tb)
print(f"{counts['ticks']} ticks checked against the oracle beside {counts['builds']} graph builds / runs / destroys on two other threads; {len(errors)} errors")
sys.exit(1 if errors else 0)
