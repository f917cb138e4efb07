"""Timed ingest through the C-ABI: MediaSource / StreamInput pacing against the oracle's restatement on the
same seeded scenarios, the page-locked frame staging ring, and both modules feeding video nodes of a graph -- bit-exact."""
from fractions import Fraction as F

import numpy as np
import pytest

import ingest_model as im
import oracle
import oracle_video as ov
from mixlab_amd import abi, ingest, video
from mixlab_amd.workspace import Workspace
from test_ingest_oracle import play_media, play_stream

pytestmark = pytest.mark.gpu
SR, SPT = 44100, 735


class FrameBook:
    """frame ids of a scenario <-> small device frames"""

    def __init__(self):
        self.by_id, self.by_handle = {}, {}

    def get(self, fid):
        if fid not in self.by_id:
            f = video.DFrame(16, 16)
            self.by_id[fid] = f; self.by_handle[f.handle] = fid
        return self.by_id[fid]

    def name(self, tv):
        if tv is None:
            return None
        fr, dur, off = tv
        return self.by_handle[fr.handle], dur, off


class AbiMedia:
    def __init__(self, book):
        self.m, self.book = ingest.MediaSource(SR, 60), book

    def set_media(self, present):
        self.m.set_media(present)

    def send(self, fid, pts, dur):
        try:
            return 1 if self.m.send(self.book.get(fid), pts, dur) else 0
        except abi.MxError as e:
            assert e.code == abi.MX_ERR_INVALID
            return -1

    def run_tick(self, t):
        return self.book.name(self.m.run_tick(t))


@pytest.mark.parametrize("seed", range(8))
def test_media_source_pacing_equals_oracle(seed):
    acts = im.media_scenario(seed)
    got = play_media(AbiMedia(FrameBook()), acts)
    want = play_media(oracle.OMediaSource(SR, 60), acts)
    assert got == want


class AbiStream:
    def __init__(self, book):
        self.s, self.book = ingest.StreamInput(SR), book

    def listen(self, on):
        self.s.listen(on)

    def write_audio(self, sid, ts, samples):
        return self.s.write_audio(sid, ts, samples)

    def write_video(self, sid, ts, fid, dur):
        return self.s.write_video(sid, ts, self.book.get(fid), dur)

    def run_tick(self, t, n_out):
        a, v, z = self.s.run_tick(t, n_out)
        return a, self.book.name(v), z


@pytest.mark.parametrize("seed", range(8))
def test_stream_input_pacing_and_reblocking_equal_oracle(seed):
    acts = im.stream_scenario(seed)
    got = play_stream(AbiStream(FrameBook()), acts)
    want = play_stream(oracle.OStreamInput(SR), acts)
    assert got == want


def test_ingest_argument_errors():
    s = ingest.StreamInput(SR)
    with pytest.raises(abi.MxError) as e:
        s.write_audio(0, 0, np.zeros(4, np.int16))          # SourceId is non-zero
    assert e.value.code == abi.MX_ERR_INVALID
    s.listen(False)
    assert s.write_audio(1, 0, np.zeros(4, np.int16)) is False   # nobody listens: Err(())
    with pytest.raises(abi.MxError):
        s.run_tick(0, 3)                                      # interleaved stereo: even
    m = ingest.MediaSource()
    with pytest.raises(abi.MxError):
        m.send(video.DFrame(16, 16), 0, F(1, 30))             # no media open


# ------------------------------------------------------------------------------------------------
# staging ring
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fmt", [video.PIXFMT_YUV420P, video.PIXFMT_YUV422P, video.PIXFMT_YUV444P, video.PIXFMT_NV12], ids=["420p", "422p", "444p", "nv12"])
def test_frame_stager_round_trip_and_reuse(fmt):
    """20 frames of three sizes through a ring of 3 slots; only the last two frames of each size stay referenced, so the device frames
    themselves are recycled too.  Host strides differ from the device's; every download equals what was uploaded."""
    rng = np.random.default_rng(fmt)
    st = ingest.FrameStager(slots=3)
    st.fence(None)                      # name a consumer stream: pooled frames may be recycled
    keep = []
    for k in range(20):
        w, h = [(322, 182), (64, 48), (1280, 720)][k % 3]
        hf = ov.HostFrame(w, h, fmt=fmt).fill(k, seed=5)
        planes = list(hf.visible())     # nv12: (Y, interleaved UV)
        padded = []
        for a in planes:                # a host stride wider than the row
            b = rng.integers(0, 256, (a.shape[0], a.shape[1] + int(rng.integers(0, 40))), dtype=np.uint8)
            b[:, : a.shape[1]] = a
            padded.append(b)
        # _host_frame takes stride = array row length; the visible width comes from (w, h)
        d = st.upload(padded, w, h, fmt)
        keep.append((d, planes))
        keep = keep[-6:]
        st.sync()
        for dd, want in keep:
            for p, (a, b) in enumerate(zip(dd.download(), want)):
                assert np.array_equal(a, b), f"frame {k}: plane {p} differs"


@pytest.mark.parametrize("fmt", [video.PIXFMT_YUV420P10, video.PIXFMT_P010], ids=["yuv420p10", "p010"])
def test_frame_stager_takes_ten_bit_pictures(fmt):
    """What a 10-bit decoder hands over -- 16-bit words, strides in bytes -- through the staging ring: the device frame holds the words as they came, and a
    scaler makes of it what the oracle makes of the 8-bit frame it stands for (include/mixlab_gpu.h mx_pixfmt)."""
    rng = np.random.default_rng(fmt)
    st = ingest.FrameStager(slots=2)
    st.fence(None)
    for k in range(5):
        w, h = [(322, 182), (1280, 720)][k % 2]
        y = rng.integers(0, 1024, (h, w), dtype=np.uint16); u = rng.integers(0, 1024, (h // 2, w // 2), dtype=np.uint16); v = rng.integers(0, 1024, (h // 2, w // 2), dtype=np.uint16)
        if fmt == video.PIXFMT_P010:
            uv = np.empty((h // 2, w), np.uint16); uv[:, 0::2] = u; uv[:, 1::2] = v
            planes = [y << 6, uv << 6]
        else:
            planes = [y, u, v]
        wide = []
        for a in planes:                # a host stride wider than the row
            b = rng.integers(0, 65536, (a.shape[0], a.shape[1] + 2 * int(rng.integers(0, 20))), dtype=np.uint16)
            b[:, : a.shape[1]] = a
            wide.append(b)
        d = st.upload(wide, w, h, fmt)
        st.sync()
        for got, want in zip(d.download(), planes):
            assert np.array_equal(got, want)
        out = video.DFrame(640, 360)
        video.scale(d, out)
        ref = ov.HostFrame(640, 360); ov.blank(ref); ov.dynamic_scale(ov.deep_to_8(planes, w, h, fmt), ref)
        for p, (a, b) in enumerate(zip(out.download(), ref.visible())):
            assert np.array_equal(a, b), f"frame {k}: plane {p}"


def test_frame_stager_acquire_commit_is_copy_free_and_bounded():
    """A decoder writing straight into the slots: three pictures held at once (reference pictures), committed out of order; a fourth
    acquire while all three slots are held is MX_ERR_FULL."""
    st = ingest.FrameStager(slots=3)
    pics, held = {}, []
    for k in range(3):
        hf = ov.HostFrame(200, 120).fill(k, seed=9)
        ticket, views = st.acquire(200, 120)
        assert views[0].shape[1] % 64 == 0 and views[1].shape[1] % 64 == 0          # rows laid out like the device frame
        assert not views[0][:, 200:].any() and (views[1][:, 100:] == 0x80).all()   # padding is blank already
        for v, src in zip(views, hf.visible()):
            v[:, : src.shape[1]] = src
        pics[ticket] = hf; held.append(ticket)
    with pytest.raises(abi.MxError) as e:
        st.acquire(200, 120)
    assert e.value.code == abi.MX_ERR_FULL
    out = {t: st.commit(t) for t in (held[1], held[2], held[0])}
    with pytest.raises(abi.MxError):
        st.commit(held[0])                 # not held any more
    st.sync()
    for t, d in out.items():
        for a, b in zip(d.download(), pics[t].visible()):
            assert np.array_equal(a, b)


# ------------------------------------------------------------------------------------------------
# both modules feeding a graph
# ------------------------------------------------------------------------------------------------
def test_media_source_feeds_a_video_mixer_in_batches():
    """A 24 fps medium of 320x180 frames, staged through the pinned ring, paced by MediaSource into a 60 ticks/s graph that runs 5 ticks per
    submission; layer B is a still of another size.  After every submission the program frame equals the oracle's: MediaSource oracle ->
    reference VideoMixer, ticked."""
    T = 5
    ws = Workspace(SR, 60)
    sa, sb = ws.source_video(), ws.source_video()
    mx = ws.video_mixer(a=0, b=1, fader=0.4)
    ws.connect(sa, 0, mx, 0); ws.connect(sb, 0, mx, 1)
    g = ws.build(max_ticks_per_run=T)
    st = ingest.FrameStager(slots=4)
    still = ov.HostFrame(212, 120).fill(99, seed=8)
    dstill = video.DFrame(212, 120).upload(*still.visible())
    video.graph_set_video_source(g, sb, dstill, dur=(1, 1), off=(0, 1), repeat=False)   # one frame, active for a second

    ms, oms = ingest.MediaSource(SR, 60), oracle.OMediaSource(SR, 60)
    ms.set_media(True); oms.set_media(True)
    omx = ov.OracleVideoMixer(a=0, b=1, fader=0.4)
    host, dev = {}, {}
    nxt = 0                                # next frame of the medium the decode thread will offer
    seen_frames = 0
    for run in range(16):
        # the decode thread runs ahead as far as the channel lets it
        while True:
            if nxt not in host:
                host[nxt] = ov.HostFrame(320, 180).fill(nxt, seed=21)
                dev[nxt] = st.upload(list(host[nxt].visible()), 320, 180)
            ok = ms.send(dev[nxt], F(nxt, 24), F(1, 24))
            assert oms.send(nxt + 1, F(nxt, 24), F(1, 24)) == (1 if ok else 0)
            if not ok:
                break
            nxt += 1
        st.fence_graph(g)
        ms.feed(g, sa, run * T, T)
        g.run_ticks(run * T, T)
        want = None
        for k in range(T):
            tick = run * T + k
            tv = oms.run_tick(tick * SPT)
            a_in = None
            if tv is not None:
                seen_frames += 1
                a_in = (host[tv[0] - 1], (tv[1].numerator, tv[1].denominator), (tv[2].numerator, tv[2].denominator))
            b_in = (still, (1, 1), (0, 1)) if tick == 0 else None
            want = omx.run_tick(tick * SPT, [a_in, b_in, None, None])
        got = video.graph_video_output(g, mx, 0)
        assert (got is None) == (want is None)
        if want is not None:
            assert (got.width, got.height) == (want.w, want.h)
            for p, (a, b) in enumerate(zip(got.download(), want.visible())):
                assert np.array_equal(a, b), f"run {run}: plane {p} differs"
    assert seen_frames >= 20      # two frames per submission is all a channel of two lets through


def test_stream_input_feeds_audio_and_video_nodes():
    T = 4
    ws = Workspace(SR, 60)
    sa = ws.source_stereo(); amp = ws.amplifier(1.0, 0.0); ws.connect(sa, 0, amp, 0)
    sv = ws.source_video(); mx = ws.video_mixer(a=0, b=None, fader=0.7); ws.connect(sv, 0, mx, 0)
    g = ws.build(max_ticks_per_run=T)
    acts = im.stream_scenario(3, n_ticks=48)
    si, osi = ingest.StreamInput(SR), oracle.OStreamInput(SR)
    book, pics = {}, {}

    def frame_of(fid):
        if fid not in book:
            pics[fid] = ov.HostFrame(64, 48).fill(fid, seed=2)
            book[fid] = video.DFrame(64, 48).upload(*pics[fid].visible())
        return book[fid]

    omx = ov.OracleVideoMixer(a=0, b=None, fader=0.7)
    delivered = 0
    for run in range(len(acts) // T):
        for tick in range(run * T, run * T + T):      # everything the network threads wrote before this submission
            for act in acts[tick]:
                if act[0] == "listen":
                    si.listen(act[1]); osi.listen(act[1])
                elif act[0] == "audio":
                    assert si.write_audio(act[1], act[2], act[3]) == osi.write_audio(act[1], act[2], act[3])
                else:
                    assert si.write_video(act[1], act[2], frame_of(act[3]), act[4]) == osi.write_video(act[1], act[2], act[3], act[4])
        zeroed = si.feed(g, sa, sv, run * T, T)
        g.run_ticks(run * T, T)
        want_audio, want_zero, want_pic = [], 0, None
        for k in range(T):
            a, v, z = osi.run_tick((run * T + k) * SPT, 2 * SPT)
            want_audio.append(a.astype(np.float32) / np.float32(32768.0)); want_zero += z
            vin = None
            if v is not None:
                delivered += 1
                vin = (pics[v[0]], (v[1].numerator, v[1].denominator), (v[2].numerator, v[2].denominator))
            want_pic = omx.run_tick((run * T + k) * SPT, [vin, None, None, None])
        assert zeroed == want_zero
        got = g.read_output(amp, 0, T, True)
        assert np.array_equal(got.view(np.uint32).ravel(), np.concatenate(want_audio).view(np.uint32))
        gp = video.graph_video_output(g, mx, 0)
        assert (gp is None) == (want_pic is None)
        if want_pic is not None:
            for a, b in zip(gp.download(), want_pic.visible()):
                assert np.array_equal(a, b)
    assert delivered >= 5


def test_queue_video_source_validates_order():
    ws = Workspace(SR, 60)
    sv = ws.source_video(); mx = ws.video_mixer(a=0, b=None, fader=0.0); ws.connect(sv, 0, mx, 0)
    g = ws.build(max_ticks_per_run=4)
    f = video.DFrame(16, 16)
    ingest.graph_queue_video_source(g, sv, 2, f)
    with pytest.raises(abi.MxError) as e:
        ingest.graph_queue_video_source(g, sv, 2, f)        # one frame per tick, ascending
    assert e.value.code == abi.MX_ERR_INVALID
    g.run_ticks(0, 2)
    assert video.graph_video_output(g, mx, 0) is None        # ticks 0 and 1: None
    g.run_ticks(2, 1)
    assert video.graph_video_output(g, mx, 0) is not None


def test_decode_thread_and_engine_thread_run_concurrently():
    """The shape of the real thing: a decode thread writes pictures into staging slots, commits them and sends them down the channel of two
    (retrying while it is full) while the engine thread fences, feeds and runs submissions.  Timing decides WHICH tick a frame leaves on;
    whatever it decides, the frames a Monitor node kept are the medium's pictures, complete, in order, none twice."""
    import threading
    import time
    N, T = 40, 4
    ws = Workspace(SR, 60)
    sv = ws.source_video()
    mon = ws.monitor(96, 64)                 # the medium's own size: the node keeps the frames themselves
    ws.connect(sv, 0, mon, 0)
    g = ws.build(max_ticks_per_run=T)
    st = ingest.FrameStager(slots=3)
    st.fence_graph(g)
    ms = ingest.MediaSource(SR, 60)
    ms.set_media(True)
    pics = [ov.HostFrame(96, 64).fill(k, seed=77) for k in range(N)]
    errors = []

    def decode():
        try:
            for k in range(N):
                ticket, views = st.acquire(96, 64)
                for v, src in zip(views, pics[k].visible()):
                    v[:, : src.shape[1]] = src
                d = st.commit(ticket)
                while not ms.send(d, F(k, 50), F(1, 50)):
                    time.sleep(0.0002)
        except Exception as e:          # surfaced by the main thread
            errors.append(e)

    th = threading.Thread(target=decode)
    th.start()
    got, tick = [], 0
    deadline = time.time() + 60
    while len(got) < N and time.time() < deadline:
        st.fence_graph(g)
        ms.feed(g, sv, tick, T)
        g.run_ticks(tick, T)
        for k, planes in enumerate(ingest.graph_read_monitor_video(g, mon, 0, T)):
            if planes is not None:
                got.append(planes)
        tick += T
    th.join()
    assert not errors, errors
    assert len(got) == N
    for k, planes in enumerate(got):
        for a, b in zip(planes, pics[k].visible()):
            assert np.array_equal(a, b), f"frame {k} differs"
