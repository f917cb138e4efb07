"""The Monitor / StreamOutput hand-off per tick: DynamicScaler 1080p -> 560x350 / 1120x700 on the device (widened bicubic), the read-back
of the small frame, and the i16 mix.  us per tick, one GPU."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
from mixlab_amd import video, ingest
from mixlab_amd.workspace import Workspace

SR, SPT, T = 48000, 800, 64
rng = np.random.default_rng(0)
for mon in ((560, 350), (1120, 700)):
    ws = Workspace(SR, 60)
    sa, sb = ws.source_video(), ws.source_video()
    mx = ws.video_mixer(a=0, b=1, fader=0.5)
    ws.connect(sa, 0, mx, 0); ws.connect(sb, 0, mx, 1)
    src = ws.source_stereo(); amp = ws.amplifier(1.0, 0.0); ws.connect(src, 0, amp, 0)
    m = ws.monitor(*mon); ws.connect(mx, 0, m, 0); ws.connect(amp, 0, m, 1)
    g = ws.build(max_ticks_per_run=T)
    frames = []
    for k in range(4):
        d = video.DFrame(1920, 1080)
        d.upload(rng.integers(0, 256, (1080, 1920), dtype=np.uint8), rng.integers(0, 256, (540, 960), dtype=np.uint8), rng.integers(0, 256, (540, 960), dtype=np.uint8))
        frames.append(d)
    video.graph_set_video_source_ring(g, sa, frames[:2]); video.graph_set_video_source_ring(g, sb, frames[2:])
    g.write_source(src, rng.standard_normal(T * 2 * SPT).astype(np.float32) * 0.3, T)
    for rep in range(3):
        g.run_ticks(rep * T, T); g.sync()
    t0 = time.perf_counter()
    R = 10
    for rep in range(R):
        g.run_ticks((3 + rep) * T, T)
    g.sync()
    dt = (time.perf_counter() - t0) / (R * T)
    t1 = time.perf_counter()
    pcm = ingest.graph_read_monitor_audio_i16(g, m, T, SPT)
    small = []
    for k in range(T):
        ts, vid = ingest.graph_read_monitor_tick(g, m, k)
        small.append(vid[0].download())
    dr = (time.perf_counter() - t1) / T
    lay = ingest.graph_monitor_layout(g, m)
    for name, buf in (("pageable", np.zeros(T * lay.frame_bytes, np.uint8)), ("page-locked", ingest.PinnedBuffer(T * lay.frame_bytes))):
        arr = buf if isinstance(buf, np.ndarray) else buf.a
        ingest.graph_read_monitor_video(g, m, 0, T, out=arr)
        t2 = time.perf_counter()
        for _ in range(5):
            ingest.graph_read_monitor_video(g, m, 0, T, out=arr)
        db = (time.perf_counter() - t2) / (5 * T)
        print(f"monitor {mon[0]}x{mon[1]}: packed read-back of the {T} kept frames into {name} memory: {db * 1e6:6.1f} us per frame ({lay.frame_bytes / db / 1e9:.1f} GB/s)")
    print(f"monitor {mon[0]}x{mon[1]}: {dt * 1e6:7.1f} us per tick on the device (cross-fade of two 1080p layers + DynamicScaler), "
          f"{dr * 1e6:7.1f} us per tick to read the frame and the PCM back (synchronous, python)")
