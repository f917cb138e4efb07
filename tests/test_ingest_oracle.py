"""CPU: the C oracle of the two ingest modules' pacing (oracle/mixlab_oracle_ingest.c) against an independent plain-Python restatement
(tests/ingest_model.py) on seeded scenarios, plus hand-worked cases read off the reference text."""
from fractions import Fraction as F

import numpy as np
import pytest

import ingest_model as im
import oracle


def play_media(src, acts, sr=44100, spt=735):
    """drive a MediaSource-like object; a frame the channel refused is offered again on later ticks, in order (the decode thread blocks)"""
    out, backlog = [], []
    for tick, a in enumerate(acts):
        for act in a:
            if act[0] == "set_media":
                src.set_media(act[1]); backlog = []
            else:
                backlog.append(act)
        while backlog:
            rc = src.send(*backlog[0][1:])
            if rc == 0:
                break
            backlog.pop(0)      # sent, or no receiver (-1: the reference's thread ends, the frame is lost)
        out.append(src.run_tick(tick * spt))
    return out


@pytest.mark.parametrize("seed", range(12))
def test_media_source_oracle_equals_python_model(seed):
    acts = im.media_scenario(seed)
    got = play_media(oracle.OMediaSource(), acts)
    want = play_media(im.PyMediaSource(), acts)
    assert got == want
    assert sum(x is not None for x in want) > 20


def test_media_source_hand_worked():
    """24 fps into 60 ticks/s at 44.1 kHz: epoch = the tick the first frame is received on; a frame leaves on the first tick whose END
    lies beyond its pts (media_source.rs:113-114), one frame per tick at most, offsets exact."""
    m = oracle.OMediaSource()
    assert m.run_tick(0) is None and m.send(1, 0, F(1, 24)) == -1          # no media yet
    m.set_media(True)
    assert [m.send(k + 1, F(k, 24), F(1, 24)) for k in range(3)] == [1, 1, 0]   # sync_channel(2)
    assert m.run_tick(5 * 735) == (1, F(1, 24), F(0))                       # epoch = 5/60
    assert m.send(3, F(2, 24), F(1, 24)) == 1
    assert m.run_tick(6 * 735) is None                                      # frame 2 received; pts 5/60 + 1/24 = 7.5/60 not before 7/60
    assert m.run_tick(7 * 735) == (2, F(1, 24), F(1, 120))                  # 7.5/60 - 7/60; frame 3 received on this tick
    assert m.run_tick(8 * 735) is None
    assert m.run_tick(20 * 735) == (3, F(1, 24), F(10, 60) - F(20, 60))     # late: negative offset, pts - start
    m.set_media(True)                                                       # a new medium: new epoch
    m.send(9, F(1, 2), F(1, 30))
    assert m.run_tick(100 * 735) is None and m.run_tick(129 * 735) is None
    assert m.run_tick(130 * 735) == (9, F(1, 30), F(0))


def play_stream(src, acts, spt=735):
    out = []
    for tick, a in enumerate(acts):
        for act in a:
            if act[0] == "listen":
                src.listen(act[1])
            elif act[0] == "audio":
                src.write_audio(act[1], act[2], act[3])
            else:
                src.write_video(act[1], act[2], act[3], act[4])
        s, v, z = src.run_tick(tick * spt, 2 * spt)
        out.append((s.tobytes(), v, z))
    return out


@pytest.mark.parametrize("seed", range(12))
def test_stream_input_oracle_equals_python_model(seed):
    acts = im.stream_scenario(seed)
    got = play_stream(oracle.OStreamInput(), acts)
    want = play_stream(im.PyStreamInput(), acts)
    assert got == want
    assert sum(x[1] is not None for x in want) > 20 and any(x[2] for x in want)


def test_stream_input_hand_worked():
    s = oracle.OStreamInput()
    n = 2 * 735
    # video before any audio: no source timing yet => offset 0, delivered at once (stream_input.rs:127-133)
    s.write_video(1, F(7), 11, F(1, 30))
    a, v, z = s.run_tick(0, n)
    assert v == (11, F(1, 30), F(0)) and z == n and not a.any()
    # first audio frame of source 1 at its time 7 s on engine tick 1: epoch = 1/60 - 7
    s.write_audio(1, F(7), np.arange(2 * 1000, dtype=np.int16))
    s.write_video(1, F(7) + F(1, 30), 12, F(1, 30))                 # due 1/30 s after the tick start: beyond the 1/60 s tick => held
    a, v, z = s.run_tick(735, n)
    assert np.array_equal(a, np.arange(n, dtype=np.int16)) and z == 0 and v is None
    a, v, z = s.run_tick(2 * 735, n)                                # the remainder of the frame, then silence
    assert np.array_equal(a[:530], np.arange(n, 2000, dtype=np.int16)) and z == n - 530 and not a[530:].any()
    assert v == (12, F(1, 30), F(1, 60))                            # now exactly one tick ahead: not > tick_duration, so it leaves


def test_stream_input_offset_equal_to_tick_is_delivered():
    """`if tick_offset > tick_duration` puts the frame back (stream_input.rs:135): an offset of exactly one tick leaves."""
    s = oracle.OStreamInput()
    s.write_audio(5, F(0), np.zeros(2 * 735, np.int16))
    s.write_video(5, F(2, 60), 1, F(1, 30))
    assert s.run_tick(0, 2 * 735)[1] is None                        # 2/60 > 1/60
    s.write_audio(5, F(1, 60), np.zeros(2 * 735, np.int16))
    assert s.run_tick(735, 2 * 735)[1] == (1, F(1, 30), F(1, 60))   # exactly one tick ahead: delivered


# ------------------------------------------------------------------------------------------------
# the C-ABI's host state machines need no device: StreamInput's audio side and its source timing on CPU
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_stream_input_abi_audio_side_equals_oracle_without_a_gpu(seed):
    from mixlab_amd import ingest
    acts = [[a for a in tick if a[0] != "video"] for tick in im.stream_scenario(seed)]   # device frames need a GPU; the rest does not

    class Abi:
        def __init__(self):
            self.s = ingest.StreamInput(44100)

        def listen(self, on):
            self.s.listen(on)

        def write_audio(self, sid, ts, samples):
            return self.s.write_audio(sid, ts, samples)

        def run_tick(self, t, n_out):
            return self.s.run_tick(t, n_out)

    got = play_stream(Abi(), acts)
    want = play_stream(oracle.OStreamInput(44100), acts)
    assert got == want
    assert any(x[2] for x in want)


def test_media_source_abi_without_frames_needs_no_gpu():
    from mixlab_amd import ingest
    m = ingest.MediaSource(48000, 60)
    assert m.run_tick(0) is None
    m.set_media(True)
    assert [m.run_tick(k * 800) for k in range(5)] == [None] * 5
    m.set_media(False)
    assert m.run_tick(4000) is None
