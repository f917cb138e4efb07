"""Host-side graph description: the frozen part of the reference's Workspace
(src/engine/workspace.rs:13-19 -- modules + connections), in the form both the C ABI
(include/mixlab_gpu.h: mx_node / mx_edge) and the test oracle consume.

Defaults follow protocol/src/lib.rs (EnvelopeParams::default :318-327, MixerChannelParams::default
:342-347 -- gain 0 dB, fader 0.0, cue false).
"""
from __future__ import annotations

from . import abi


class Workspace:
    def __init__(self, sample_rate: int = 44100, ticks_per_second: int = 60):
        self.sample_rate = sample_rate
        self.ticks_per_second = ticks_per_second
        self.nodes: list[tuple[int, object]] = []
        self._conn: dict[tuple[int, int], tuple[int, int]] = {}   # InputId -> OutputId (HashMap, src/engine/workspace.rs:17)

    @property
    def spt(self) -> int:
        return self.sample_rate // self.ticks_per_second  # src/engine.rs:55

    def add(self, kind: int, params=None) -> int:
        self.nodes.append((kind, params))
        return len(self.nodes) - 1

    def connect(self, src: int, src_port: int, dst: int, dst_port: int) -> None:
        """InputId(dst, dst_port) -> OutputId(src, src_port); a later connect to the same input replaces it
        (HashMap insert, src/engine/workspace.rs:110)."""
        self._conn[(dst, dst_port)] = (src, src_port)

    @property
    def edges(self) -> list[tuple[int, int, int, int]]:
        """(src, src_port, dst, dst_port) per connection."""
        return [(s, sp, d, dp) for (d, dp), (s, sp) in self._conn.items()]

    # ---- module constructors (ModuleT::create) ----
    def oscillator(self, freq: float, waveform: int) -> int:
        return self.add(abi.KIND_OSCILLATOR, abi.OscillatorParams(freq, waveform, 0))

    def mixer(self, channels) -> int:
        """channels: iterable of (gain_db, fader, cue)."""
        return self.add(abi.KIND_MIXER, [abi.MixerChannelParams(g, f, 1 if c else 0) for (g, f, c) in channels])

    def eq_three(self, lo_db: float, mid_db: float, hi_db: float) -> int:
        return self.add(abi.KIND_EQ_THREE, abi.EqThreeParams(lo_db, mid_db, hi_db))

    def envelope(self, attack_ms=25.0, decay_ms=500.0, sustain=0.8, release_ms=200.0) -> int:
        return self.add(abi.KIND_ENVELOPE, abi.EnvelopeParams(attack_ms, decay_ms, sustain, release_ms))

    def amplifier(self, amplitude: float, mod_depth: float) -> int:
        return self.add(abi.KIND_AMPLIFIER, abi.AmplifierParams(amplitude, mod_depth))

    def fm_sine(self, freq_lo: float, freq_hi: float) -> int:
        return self.add(abi.KIND_FM_SINE, abi.FmSineParams(freq_lo, freq_hi))

    def trigger(self, gate_open: bool) -> int:
        return self.add(abi.KIND_TRIGGER, abi.TriggerParams(1 if gate_open else 0))

    def stereo_panner(self) -> int:
        return self.add(abi.KIND_STEREO_PANNER, None)

    def stereo_splitter(self) -> int:
        return self.add(abi.KIND_STEREO_SPLITTER, None)

    def plotter(self) -> int:
        return self.add(abi.KIND_PLOTTER, None)

    def source_mono(self) -> int:
        return self.add(abi.KIND_SOURCE_MONO, None)

    def source_stereo(self) -> int:
        return self.add(abi.KIND_SOURCE_STEREO, None)

    def build(self, max_ticks_per_run: int = 1, flags: int = 0, device: int = -1, stream=None) -> "abi.Graph":
        return abi.Graph(self.nodes, self.edges, self.sample_rate, self.ticks_per_second, max_ticks_per_run, flags, device, stream)

    # ---- build-specified audio extras (no reference module) ----
    def fir(self, taps) -> int:
        import struct
        taps = [float(t) for t in taps]
        return self.add(abi.KIND_FIR, struct.pack("<II", len(taps), 0) + struct.pack(f"<{len(taps)}d", *taps))

    def resample(self, up: int, down: int, taps) -> int:
        """taps: up x taps_per_phase polyphase table (row-major)."""
        import struct
        import numpy as np
        t = np.ascontiguousarray(taps, dtype=np.float64)
        assert t.ndim == 2 and t.shape[0] == up
        return self.add(abi.KIND_RESAMPLE, struct.pack("<IIII", up, down, t.shape[1], 0) + t.tobytes())

    # ---- video nodes ----
    def video_mixer(self, a=None, b=None, fader=1.0) -> int:
        """VideoMixerParams (protocol/src/lib.rs:405-420); default fader 1.0 = start at A."""
        from .video import VideoMixerParams
        return self.add(abi.KIND_VIDEO_MIXER, VideoMixerParams(-1 if a is None else a, -1 if b is None else b, fader))

    def source_video(self) -> int:
        return self.add(abi.KIND_SOURCE_VIDEO, None)

    def monitor(self, width=560, height=350, queue_depth=None) -> int:
        """Monitor (560 x 350, monitor.rs:21-22) / StreamOutput (1120 x 700, stream_output.rs:23-24) hand-off: in Video, Stereo.
        queue_depth (mx_monitor_params_ex): ticks are dropped while that many wait for the consumer (try_send on a channel of two,
        monitor.rs:163-177); None = keep every tick"""
        import struct
        if queue_depth is None:
            return self.add(abi.KIND_MONITOR, struct.pack("<II", width, height))
        return self.add(abi.KIND_MONITOR, struct.pack("<IIII", width, height, queue_depth, 0))

    def video_to_rgba(self, matrix_q12=None) -> int:
        from .video import to_rgba_params
        return self.add(abi.KIND_VIDEO_TO_RGBA, to_rgba_params(matrix_q12))
