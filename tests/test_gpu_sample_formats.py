"""the sample formats either side of the path, converted on the device.
Sinks: f32 -> i16 (src/video/encode.rs:183-195).  Ingest: i16 -> f32 (src/module/stream_input.rs:167-173)."""
import ctypes as C

import numpy as np
import pytest

import oracle
import synth
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace
from test_gpu_audio_parity import SPT, assert_bit_exact

pytestmark = pytest.mark.gpu

oracle.lib.orc_f32_to_i16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
oracle.lib.orc_i16_to_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]


def test_sink_f32_to_i16_bit_exact_including_clip_and_nan():
    ws = Workspace(44100, 60)
    s = ws.source_stereo(); a = ws.amplifier(1.0, 0.0)
    ws.connect(s, 0, a, 0)
    g = ws.build(max_ticks_per_run=2)
    x = (synth.noise(1, 2 * 2 * SPT) * np.float32(1.6)).astype(np.float32)       # clips on both sides
    x[:12] = [1.0, -1.0, 0.0, -0.0, 0.9999999, -0.9999999, 1.5, -7.0, np.nan, 3.0517578e-05, -3.0517578e-05, 2.9e-05]
    g.write_source(s, x, 2); g.run_ticks(0, 2)
    got = g.read_output_i16(a, 0, 2, True)
    want = np.empty(x.size, np.int16)
    oracle.lib.orc_f32_to_i16(x.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), x.size)
    assert np.array_equal(got, want)
    assert got[0] == 32767 and got[1] == -32767 and got[6] == 32767 and got[7] == -32767 and got[8] == 0


def test_ingest_i16_to_f32_bit_exact():
    ws = Workspace(44100, 60)
    s = ws.source_mono(); e = ws.eq_three(0.0, 0.0, 0.0)
    ws.connect(s, 0, e, 0)
    g = ws.build(flags=abi.FLAG_EQ_EXACT | abi.FLAG_NO_FUSE)
    pcm = (synth.splitmix64(5, SPT) >> np.uint64(48)).astype(np.uint16).view(np.int16)
    pcm[:4] = [-32768, 32767, 0, -1]
    g.write_source_i16(s, pcm, 1); g.run_ticks(0, 1)
    want = np.empty(SPT, np.float32)
    oracle.lib.orc_i16_to_f32(pcm.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), SPT)
    assert_bit_exact(g.read_output(s, 0, 1, False), want, "i16 -> f32")
    assert want[0] == -1.0 and want[1] == np.float32(32767 / 32768)


def test_sink_i16_of_a_mono_stored_fused_strip():
    # a fused strip's stereo result is stored as one float per frame: the i16 read-out must still be interleaved stereo
    from test_gpu_audio_parity import strips
    ws, mix, srcs, trigs = strips(2)
    g = ws.build(max_ticks_per_run=1)
    for k, s in enumerate(srcs):
        g.write_source(s, synth.noise(k, SPT), 1)
    g.run_ticks(0, 1)
    amp = mix + 6
    f = g.read_output(amp, 0, 1, True)
    want = np.empty(f.size, np.int16)
    oracle.lib.orc_f32_to_i16(f.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), f.size)
    assert np.array_equal(g.read_output_i16(amp, 0, 1, True), want)
