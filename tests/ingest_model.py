"""A second, independent restatement of the two ingest modules' pacing rules in plain Python (fractions + deque), written from the
reference text -- test infrastructure: the C oracle (oracle/mixlab_oracle_ingest.c) is checked against it on CPU, and the random
scenarios the GPU tests drive through the C-ABI are generated here.
  MediaSource::run_tick  src/module/media_source.rs:93-126        StreamInput::run_tick  src/module/stream_input.rs:72-147"""
from collections import deque
from fractions import Fraction as F

import numpy as np


class PyMediaSource:
    def __init__(self, sr=44100, tps=60):
        self.sr, self.tps, self.media = sr, tps, None

    def set_media(self, present=True):
        self.media = {"chan": deque(), "epoch": None, "buf": deque()} if present else None

    def send(self, frame, pts, dur):
        if self.media is None:
            return -1
        if len(self.media["chan"]) == 2:
            return 0
        self.media["chan"].append((frame, F(pts), F(dur)))
        return 1

    def run_tick(self, t):
        start = F(t, self.sr)
        end = start + F(1, self.tps)
        m = self.media
        if m is None:
            return None
        if m["chan"]:
            frame, pts, dur = m["chan"].popleft()
            if m["epoch"] is None:
                m["epoch"] = start
            m["buf"].append((frame, pts + m["epoch"], dur))
        if m["buf"] and m["buf"][0][1] < end:
            frame, pts, dur = m["buf"].popleft()
            return frame, dur, pts - start
        return None


class PyStreamInput:
    RING = 65536

    def __init__(self, sr=44100):
        self.sr = sr
        self.arx, self.vrx = deque(), deque()
        self.listening = True
        self.audio_frame = self.video_frame = self.source = None

    def listen(self, listening=True):
        self.arx, self.vrx, self.listening = deque(), deque(), listening

    def write_audio(self, sid, ts, samples):
        if not self.listening or len(self.arx) >= self.RING:
            return False
        self.arx.append([sid, F(ts), list(np.asarray(samples, np.int16))])
        return True

    def write_video(self, sid, ts, frame, dur):
        if not self.listening or len(self.vrx) >= self.RING:
            return False
        self.vrx.append((sid, F(ts), frame, F(dur)))
        return True

    def run_tick(self, t, n_out):
        now, tick_dur = F(t, self.sr), F(n_out // 2, self.sr)
        vf, self.video_frame = self.video_frame, None
        if vf is None and self.vrx:
            vf = self.vrx.popleft()
        existing = self.source[0] if self.source else None
        out, zeroed = [], 0
        while len(out) < n_out:
            af, self.audio_frame = self.audio_frame, None
            if af is None and self.arx:
                af = self.arx.popleft()
            if af is None:
                zeroed = n_out - len(out)
                out += [0] * zeroed
                break
            if existing != af[0]:
                self.source = (af[0], now - af[1])
            n = min(n_out - len(out), len(af[2]))
            out += af[2][:n]
            if n < len(af[2]):
                af[2] = af[2][n:]
                self.audio_frame = af
        video = None
        if vf is not None:
            off = F(0)
            if self.source is not None:
                d = vf[1] + self.source[1] - now
                if d >= 0:
                    off = d
            if off > tick_dur:
                self.video_frame = vf
            else:
                video = (vf[2], vf[3], off)
        return np.array(out, np.int16), video, zeroed


def media_scenario(seed, n_ticks=400, sr=44100):
    """-> list of per-tick action lists: ('send', frame_id, pts, dur) | ('set_media', present), played BEFORE that tick's run_tick"""
    rng = np.random.default_rng(seed)
    fps = [F(24), F(25), F(30000, 1001), F(60), F(120), F(15)][seed % 6]
    dur = 1 / fps
    acts, fid, k = [], 1, 0   # k = frame index within the current medium
    present = False
    for tick in range(n_ticks):
        a = []
        if tick == 0 or rng.random() < 0.01:
            present = not (present and rng.random() < 0.3)
            a.append(("set_media", present)); k = 0
        # the decode thread tries to stay a little ahead of presentation; sometimes it stalls, sometimes it bursts
        tries = int(rng.integers(0, 4)) if rng.random() < 0.8 else 0
        for _ in range(tries):
            jitter = F(int(rng.integers(-3, 4)), 1000) if rng.random() < 0.2 else F(0)
            a.append(("send", fid, k * dur + jitter if k * dur + jitter >= 0 else F(0), dur))
            fid += 1; k += 1   # the model decides whether it was accepted; the driver re-sends the same frame when it was not
        acts.append(a)
    return acts


def stream_scenario(seed, n_ticks=300, sr=44100, spt=735):
    """per-tick action lists: ('audio', sid, ts, samples) | ('video', sid, ts, frame_id, dur) | ('listen', bool)"""
    rng = np.random.default_rng(1000 + seed)
    acts, fid = [], 1
    sid, a_pos, v_k = 1, 0, 0           # samples (per channel) written so far by this source; video frame index
    vfps = [F(30), F(25), F(60), F(24)][seed % 4]
    base = F(int(rng.integers(0, 5000)), 1000)   # the source's own clock starts anywhere
    for tick in range(n_ticks):
        a = []
        if rng.random() < 0.01:
            a.append(("listen", rng.random() < 0.8))
        if rng.random() < 0.02:          # a new connection: another source id, its clock starts over
            sid += 1; a_pos = 0; v_k = 0; base = F(int(rng.integers(0, 5000)), 1000)
        want = spt * (tick + 1) + int(rng.integers(-2000, 2000))
        while a_pos < want and rng.random() < 0.9:
            n = int(rng.choice([1024, 1152, 576, 441, 2048, 1, 735]))
            s = rng.integers(-32768, 32768, 2 * n, dtype=np.int64).astype(np.int16)
            a.append(("audio", sid, base + F(a_pos, sr), s)); a_pos += n
        while F(v_k) / vfps < F(a_pos, sr) and rng.random() < 0.9:
            a.append(("video", sid, base + F(v_k) / vfps, fid, 1 / vfps)); fid += 1; v_k += 1
        acts.append(a)
    return acts
