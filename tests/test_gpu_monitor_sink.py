"""The Monitor / StreamOutput hand-off as a graph node: every tick of a multi-tick submission keeps its program
frame -- through the DynamicScaler to the encoder's picture, on the device -- and the mix is served as the encoder's i16 PCM.  Checked
against the oracle composed as the reference composes it: VideoMixer -> Monitor::run_tick timestamps (monitor.rs:113-139) -> codec thread
(monitor.rs:226-236): frame_ts = ts + tick_offset, DynamicScaler::scale (encode.rs:287-295,338-397), f32 -> i16 (encode.rs:183-195)."""
import ctypes as C
from fractions import Fraction as F

import numpy as np
import pytest

import oracle
import oracle_video as ov
import synth
from mixlab_amd import abi, ingest, video
from mixlab_amd.workspace import Workspace

pytestmark = pytest.mark.gpu
SR, SPT = 44100, 735
oracle.lib.orc_f32_to_i16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]


def to_i16(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.size, np.int16)
    oracle.lib.orc_f32_to_i16(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), x.size)
    return out


def upload(hf):
    return video.DFrame(hf.w, hf.h).upload(*hf.visible())


@pytest.mark.parametrize("prog,mon", [((1280, 720), (560, 350)), ((320, 180), (560, 350)), ((560, 350), (560, 350)), ((1920, 1080), (1120, 700)), ((350, 560), (560, 350))],
                         ids=["720p-to-monitor", "upscaled", "same-size-passes-through", "1080p-to-stream-output", "pillarbox"])
def test_monitor_keeps_every_tick_of_a_submission(prog, mon):
    T, RUNS, FIRST = 6, 3, 40           # the node first runs on tick 40: its epoch
    ws = Workspace(SR, 60)
    sa, sb = ws.source_video(), ws.source_video()
    mx = ws.video_mixer(a=0, b=1, fader=0.35)
    ws.connect(sa, 0, mx, 0); ws.connect(sb, 0, mx, 1)
    src = ws.source_stereo(); amp = ws.amplifier(1.7, 0.0); ws.connect(src, 0, amp, 0)   # loud enough to clip
    m = ws.monitor(*mon)
    ws.connect(mx, 0, m, 0); ws.connect(amp, 0, m, 1)
    g = ws.build(max_ticks_per_run=T)
    omx = ov.OracleVideoMixer(a=0, b=1, fader=0.35)
    rng = np.random.default_rng(prog[0] + mon[0])
    b_layer = ov.HostFrame(prog[0] // 2 * 2, prog[1] // 2 * 2).fill(7, seed=1)
    pics, keep = {}, []
    for run in range(RUNS):
        t0 = FIRST + run * T
        audio = synth.noise(100 + run, T * 2 * SPT)
        g.write_source(src, audio, T)
        # layer A: a new frame on some ticks only, with offsets inside the tick; layer B: one long-lived still on the very first tick
        plan = {}
        for k in range(T):
            if rng.random() < 0.6:
                hf = ov.HostFrame(*prog).fill(int(rng.integers(0, 50)), seed=run * T + k)
                off = F(int(rng.integers(0, 700)), SR)
                plan[k] = (hf, F(1, 30), off)
                d = upload(hf); keep.append(d)
                ingest.graph_queue_video_source(g, sa, t0 + k, d, dur=(1, 30), off=off)
        if run == 0:
            db = upload(b_layer); keep.append(db)
            ingest.graph_queue_video_source(g, sb, t0, db, dur=(10, 1), off=(0, 1))
        g.run_ticks(t0, T)
        # ---- oracle ----
        want_audio = to_i16(oracle.amplifier_run(1.7, 0.0, audio, None))
        assert np.array_equal(ingest.graph_read_monitor_audio_i16(g, m, T, SPT), want_audio)
        assert (np.abs(want_audio.astype(np.int32)) == 32767).any()          # the clamp was exercised
        for k in range(T):
            tick = t0 + k
            a_in = None
            if k in plan:
                hf, dur, off = plan[k]
                a_in = (hf, (dur.numerator, dur.denominator), (off.numerator, off.denominator))
            b_in = (b_layer, (10, 1), (0, 1)) if (run == 0 and k == 0) else None
            program = omx.run_tick(tick * SPT, [a_in, b_in, None, None])
            ts, vid = ingest.graph_read_monitor_tick(g, m, k)
            assert ts == F(tick * SPT, SR) - F(FIRST * SPT, SR)
            assert (vid is None) == (program is None)
            if program is None:
                continue
            frame, frame_ts, dur = vid
            assert frame_ts == ts and dur == F(1, 60)                       # a VideoMixer program frame: offset 0, 1/60 s (video_mixer.rs:241-247)
            if (program.w, program.h) == mon:
                want = program
            else:
                want = ov.HostFrame(*mon); ov.blank(want); ov.dynamic_scale(program, want)
            assert (frame.width, frame.height) == mon
            for p, (x, y) in enumerate(zip(frame.download(), want.visible())):
                assert np.array_equal(x, y), f"run {run} tick {k}: plane {p} differs"
            pics[(run, k)] = frame
        # the packed read-back: every kept picture of the submission in one copy
        packed = ingest.graph_read_monitor_video(g, m, 0, T)
        for k in range(T):
            ts, vid = ingest.graph_read_monitor_tick(g, m, k)
            assert (packed[k] is None) == (vid is None)
            if vid is not None:
                for x, y in zip(packed[k], vid[0].download()):
                    assert np.array_equal(x, y)
        part = ingest.graph_read_monitor_video(g, m, 2, 3)
        for k in range(3):
            assert (part[k] is None) == (packed[2 + k] is None)
            if part[k] is not None:
                assert all(np.array_equal(x, y) for x, y in zip(part[k], packed[2 + k]))
    assert len(pics) >= 8


def test_monitor_passes_source_offsets_and_handles_disconnected_inputs():
    """Fed straight from a source, the frame keeps its own tick offset and duration hint: frame_ts = ts + tick_offset (monitor.rs:229).
    Without an audio connection the mix is the zero buffer; without video every tick is None."""
    ws = Workspace(SR, 60)
    sv = ws.source_video()
    m = ws.monitor(64, 48)
    ws.connect(sv, 0, m, 0)
    m2 = ws.monitor(64, 48)             # nothing connected at all
    g = ws.build(max_ticks_per_run=4)
    hf = ov.HostFrame(64, 48).fill(3, seed=3)
    d = upload(hf)
    ingest.graph_queue_video_source(g, sv, 2, d, dur=(1001, 30000), off=(-5, 441))
    g.run_ticks(0, 4)
    for k in range(4):
        ts, vid = ingest.graph_read_monitor_tick(g, m, k)
        assert ts == F(k, 60)
        if k != 2:
            assert vid is None
        else:
            frame, frame_ts, dur = vid
            assert frame_ts == F(2, 60) + F(-5, 441) and dur == F(1001, 30000)
            assert frame.handle == d.handle                                  # same picture settings: the scaler returns its input (encode.rs:342-345)
        assert ingest.graph_read_monitor_tick(g, m2, k) == (F(k, 60), None)
    assert not ingest.graph_read_monitor_audio_i16(g, m, 4, SPT).any()
    with pytest.raises(abi.MxError):
        ingest.graph_read_monitor_tick(g, m, 4)                              # beyond the run
    with pytest.raises(abi.MxError):
        ingest.graph_read_monitor_tick(g, sv, 0)                             # not a monitor


def test_monitor_with_a_queue_depth_drops_ticks_like_try_send_on_a_full_channel():
    """mx_monitor_params_ex.queue_depth = 2: Monitor::run_tick hands the tick to its codec thread with try_send on a channel of TWO and drops it
    when the thread lags (monitor.rs:163-177).  Model: a queue of capacity 2; a tick enters if there is room, else it is lost; the consumer takes
    ticks between submissions (mx_graph_monitor_consume).  Dropped ticks keep their timestamp, carry no picture and scale nothing."""
    T = 5
    ws = Workspace(SR, 60)
    sv = ws.source_video()
    m = ws.monitor(64, 48, queue_depth=2)
    keep_all = ws.monitor(64, 48)
    ws.connect(sv, 0, m, 0); ws.connect(sv, 0, keep_all, 0)
    g = ws.build(max_ticks_per_run=T)
    frames = [upload(ov.HostFrame(32, 24).fill(k, seed=k)) for k in range(4 * T)]
    queue, consumed_plan = 0, [0, 1, 5, 2]            # how many ticks the consumer takes after each submission
    for run in range(4):
        t0 = run * T
        for k in range(T):
            ingest.graph_queue_video_source(g, sv, t0 + k, frames[t0 + k], dur=(1, 60), off=(0, 1))
        g.run_ticks(t0, T)
        for k in range(T):
            ts, vid, dropped = ingest.graph_read_monitor_tick(g, m, k, with_dropped=True)
            ts_all, vid_all, dropped_all = ingest.graph_read_monitor_tick(g, keep_all, k, with_dropped=True)
            want_drop = queue >= 2
            if not want_drop:
                queue += 1
            assert dropped == want_drop, f"run {run} tick {k}"
            assert ts == ts_all == F(t0 + k, 60) and not dropped_all and vid_all is not None
            if want_drop:
                assert vid is None
            else:
                assert vid is not None and vid[1] == ts
                for x, y in zip(vid[0].download(), vid_all[0].download()):
                    assert np.array_equal(x, y)                                  # a kept tick's picture is what the keep-everything node holds
        n = consumed_plan[run]
        ingest.graph_monitor_consume(g, m, n)
        queue -= min(queue, n)
    # the packed read-back skips dropped ticks like ticks without a picture
    packed = ingest.graph_read_monitor_video(g, m, 0, T)
    for k in range(T):
        _ts, vid, _d = ingest.graph_read_monitor_tick(g, m, k, with_dropped=True)
        assert (packed[k] is None) == (vid is None)
    with pytest.raises(abi.MxError):
        ingest.graph_monitor_consume(g, sv, 1)                                   # not a monitor
