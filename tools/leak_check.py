"""Create / use / destroy every handle type in a loop and watch device-free memory (hipMemGetInfo) and the process RSS.
Usage: python tools/leak_check.py [iterations]"""
import ctypes as C, os, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import psutil
from mixlab_amd import abi, ingest, video
from mixlab_amd.workspace import Workspace
from test_gpu_audio_parity import strips

hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value / 2**20
proc = psutil.Process(os.getpid())
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
SR, SPT = 48000, 800
rng = np.random.default_rng(0)


def one(k):
    ws, mix, srcs, trigs = strips(8, SR)
    sv = ws.source_video(); vm = ws.video_mixer(a=0, b=None, fader=0.5); ws.connect(sv, 0, vm, 0)
    rg = ws.video_to_rgba(None); ws.connect(vm, 0, rg, 0); mon = ws.monitor(64, 48); ws.connect(vm, 0, mon, 0)
    g = ws.build(max_ticks_per_run=4)
    st = ingest.FrameStager(2)
    d = st.upload([np.zeros((90, 160), np.uint8), np.zeros((45, 80), np.uint8), np.zeros((45, 80), np.uint8)], 160, 90)
    st.fence_graph(g)
    ms = ingest.MediaSource(SR, 60); ms.set_media(True); ms.send(d, 0, (1, 30)); ms.feed(g, sv, 0, 4)
    si = ingest.StreamInput(SR); si.write_audio(1, 0, np.zeros(4000, np.int16))
    for s in srcs:
        g.write_source(s, np.zeros(4 * SPT, np.float32), 4)
    g.run_ticks(0, 4)
    ingest.graph_read_monitor_video(g, mon, 0, 4); video.graph_rgba_output(g, rg)
    sc = video.Scaler(96, 54); sc.scale(d)
    m = abi.Module(abi.KIND_EQ_THREE, abi.EqThreeParams(1.0, 2.0, 3.0)); out = np.empty(SPT, np.float32)
    m.run_tick(0, [(abi.MX_MONO, np.zeros(SPT, np.float32))], [(abi.MX_MONO, out)])
    vmx = video.VideoMixer(a=0, b=1, fader=0.5); vmx.run_tick(0, [(d, (1, 30), (0, 1)), None, None, None])
    ring = abi.PcmRing(); ring.push(np.zeros(100, np.int16))
    for o in (g, st, ms, si, sc, m, vmx, ring):
        o.close() if hasattr(o, "close") else None


for k in range(20):
    one(k)
f0, r0 = free_mb(), proc.memory_info().rss / 2**20
for k in range(iters):
    one(k)
f1, r1 = free_mb(), proc.memory_info().rss / 2**20
print(f"{iters} iterations: device free {f0:.0f} -> {f1:.0f} MiB ({f0 - f1:+.1f} MiB used), RSS {r0:.0f} -> {r1:.0f} MiB ({r1 - r0:+.1f})")
sys.exit(1 if (f0 - f1) > 64 or (r1 - r0) > 200 else 0)
