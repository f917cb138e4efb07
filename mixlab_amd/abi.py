"""ctypes binding of include/mixlab_gpu.h (plumbing for tests and bench.py -- the product is the .so).

There is no CPU fallback: if libmixlab_gpu.so is missing this module raises at import.
"""
from __future__ import annotations

import ctypes as C
import pathlib

import numpy as np

_PKG = pathlib.Path(__file__).resolve().parent
LIB_PATH = _PKG / "libmixlab_gpu.so"

if not LIB_PATH.exists():
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -m mixlab_amd.build` (hipcc, gfx950). "
        "mixlab_amd has no CPU fallback."
    )

lib = C.CDLL(str(LIB_PATH))

# ---- constants (include/mixlab_gpu.h) ----
MX_OK, MX_ERR_INVALID, MX_ERR_TYPE, MX_ERR_DEVICE, MX_ERR_NOMEM, MX_ERR_INTERNAL, MX_ERR_FULL = 0, -1, -2, -3, -4, -5, -6
MX_DISCONNECTED, MX_MONO, MX_STEREO, MX_VIDEO = 0, 1, 2, 3
(KIND_AMPLIFIER, KIND_ENVELOPE, KIND_EQ_THREE, KIND_FM_SINE, KIND_MIXER, KIND_OSCILLATOR, KIND_PLOTTER,
 KIND_STEREO_PANNER, KIND_STEREO_SPLITTER, KIND_TRIGGER, KIND_VIDEO_MIXER, KIND_SOURCE_MONO,
 KIND_SOURCE_STEREO, KIND_SOURCE_VIDEO, KIND_VIDEO_TO_RGBA, KIND_FIR, KIND_RESAMPLE, KIND_MONITOR, KIND_COUNT) = range(19)
KIND_NAMES = ["amplifier", "envelope", "eq_three", "fm_sine", "mixer", "oscillator", "plotter", "stereo_panner",
              "stereo_splitter", "trigger", "video_mixer", "source_mono", "source_stereo", "source_video", "video_to_rgba", "fir", "resample", "monitor"]
WAVE_ON, WAVE_OFF, WAVE_SINE, WAVE_SQUARE, WAVE_TRIANGLE, WAVE_SAW = range(6)
FLAG_EQ_EXACT = 1   # the default (kept as a no-op name)
FLAG_NO_FUSE = 2
FLAG_EQ_FAST = 4    # time-parallel EqThree: <= 1 ULP, not bit-exact
FLAG_FP_CONTRACT = 16   # the contracted order (mul+add fused): <= 1 ULP of the exact order, equal to the oracle's contract mode
FLAG_OVERLAP_TAIL = 8   # the last Mixer bank runs on a second stream beside the next run's earlier groups


class MixerChannelParams(C.Structure):
    _fields_ = [("gain_db", C.c_double), ("fader", C.c_double), ("cue", C.c_uint8), ("_pad", C.c_uint8 * 7)]


class EqThreeParams(C.Structure):
    _fields_ = [("gain_lo_db", C.c_double), ("gain_mid_db", C.c_double), ("gain_hi_db", C.c_double)]


class EnvelopeParams(C.Structure):
    _fields_ = [("attack_ms", C.c_double), ("decay_ms", C.c_double), ("sustain_amplitude", C.c_double), ("release_ms", C.c_double)]


class AmplifierParams(C.Structure):
    _fields_ = [("amplitude", C.c_double), ("mod_depth", C.c_double)]


class OscillatorParams(C.Structure):
    _fields_ = [("freq", C.c_double), ("waveform", C.c_uint32), ("_pad", C.c_uint32)]


class FmSineParams(C.Structure):
    _fields_ = [("freq_lo", C.c_double), ("freq_hi", C.c_double)]


class TriggerParams(C.Structure):
    _fields_ = [("gate_open", C.c_uint32)]


class Node(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("params_len", C.c_uint32), ("params", C.c_void_p)]


class Edge(C.Structure):
    _fields_ = [("src_node", C.c_uint32), ("src_port", C.c_uint32), ("dst_node", C.c_uint32), ("dst_port", C.c_uint32)]


class GraphOpts(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("ticks_per_second", C.c_uint32), ("max_ticks_per_run", C.c_uint32),
                ("flags", C.c_uint32), ("device", C.c_int32), ("_pad", C.c_int32), ("stream", C.c_void_p)]


class Frame(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("data", C.c_void_p * 3), ("stride", C.c_int32 * 3),
                ("dur_num", C.c_int64), ("dur_den", C.c_int64), ("off_num", C.c_int64), ("off_den", C.c_int64)]


class Input(C.Structure):
    _fields_ = [("kind", C.c_int), ("samples", C.c_void_p), ("len", C.c_size_t), ("video", C.c_void_p)]


class Output(C.Structure):
    _fields_ = [("kind", C.c_int), ("samples", C.c_void_p), ("len", C.c_size_t), ("video", C.c_void_p), ("video_present", C.c_int)]


def _proto(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_proto("mx_last_error", C.c_char_p)
_proto("mx_abi_version", C.c_uint32)
_proto("mx_device_count", C.c_int)
_proto("mx_graph_build", C.c_int, C.POINTER(Node), C.c_size_t, C.POINTER(Edge), C.c_size_t, C.POINTER(GraphOpts), C.POINTER(C.c_void_p))
_proto("mx_graph_destroy", None, C.c_void_p)
_proto("mx_graph_samples_per_tick", C.c_int, C.c_void_p, C.POINTER(C.c_size_t))
_proto("mx_graph_run_order", C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_size_t))
_proto("mx_graph_update_params", C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t)
class ParamEvent(C.Structure):
    """mx_param_event: ModuleT::update of `node` at the boundary before tick `tick_in_run` of the next run."""
    _fields_ = [("node", C.c_uint32), ("tick_in_run", C.c_uint32), ("params", C.c_void_p), ("params_len", C.c_size_t)]


_proto("mx_graph_schedule_params", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t)
_proto("mx_graph_schedule_params_batch", C.c_int, C.c_void_p, C.POINTER(ParamEvent), C.c_size_t)
_proto("mx_graph_eq_spec_stats", C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
_proto("mx_graph_debug_eq_records", C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))
_proto("mx_graph_debug_tail_releases", C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
_proto("mx_graph_eq_repair_stats", C.c_int, C.c_void_p, C.POINTER(C.c_uint64))
_proto("mx_graph_write_source", C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t)
_proto("mx_graph_bind_source_device", C.c_int, C.c_void_p, C.c_uint32, C.c_void_p)
_proto("mx_graph_run_ticks", C.c_int, C.c_void_p, C.c_uint64, C.c_uint32)
_proto("mx_graph_sync", C.c_int, C.c_void_p)
_proto("mx_graph_read_output", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t)
_proto("mx_graph_read_output_window", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_size_t)
_proto("mx_graph_read_output_i16", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t)
_proto("mx_graph_write_source_i16", C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t)
_proto("mx_graph_output_device_ptr", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))
_proto("mx_graph_tail_stream", C.c_int, C.c_void_p, C.POINTER(C.c_void_p))
_proto("mx_graph_stream", C.c_int, C.c_void_p, C.POINTER(C.c_void_p))
_proto("mx_graph_read_plotter", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int))
_proto("mx_graph_profile_run", C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float))
_proto("mx_graph_profile_enable", C.c_int, C.c_void_p, C.c_int)
_proto("mx_graph_profile_collect", C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint32))


class PerformanceInfo(C.Structure):
    """mx_performance_info (PerformanceInfo, protocol/src/lib.rs:32-59)."""
    _fields_ = [("realtime", C.c_int32), ("lag", C.c_int32), ("tick_rate", C.c_uint32), ("n_modules", C.c_uint32),
                ("tick_budget_us", C.c_uint64), ("engine_us", C.c_uint64)]


_proto("mx_graph_performance_info", C.c_int, C.c_void_p, C.POINTER(PerformanceInfo), C.POINTER(C.c_uint64), C.c_size_t)
_proto("mx_graph_adopt_state", C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_size_t)
_proto("mx_pcm_ring_create", C.c_int, C.POINTER(C.c_void_p))
_proto("mx_pcm_ring_destroy", None, C.c_void_p)
_proto("mx_pcm_ring_push_i16", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
_proto("mx_pcm_ring_queued", C.c_int, C.c_void_p, C.POINTER(C.c_size_t))
_proto("mx_pcm_ring_feed", C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t))
_proto("mx_module_create", C.c_int, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p))
_proto("mx_module_create_ex", C.c_int, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(GraphOpts), C.POINTER(C.c_void_p))
_proto("mx_module_update", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
_proto("mx_module_run_tick", C.c_int, C.c_void_p, C.c_uint64, C.POINTER(Input), C.c_size_t, C.POINTER(Output), C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t))
_proto("mx_module_destroy", None, C.c_void_p)


class MxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mixlab_gpu error {code}: {msg}")
        self.code = code


def check(rc: int) -> None:
    if rc != MX_OK:
        raise MxError(rc, (lib.mx_last_error() or b"").decode("utf-8", "replace"))


def params_bytes(p) -> bytes:
    """Serialise a params object (ctypes struct, list of MixerChannelParams, bytes or None)."""
    if p is None:
        return b""
    if isinstance(p, (bytes, bytearray)):
        return bytes(p)
    if isinstance(p, (list, tuple)):
        return b"".join(bytes(x) for x in p)
    return bytes(p)


class Graph:
    """Thin RAII wrapper of mx_graph_* (a frozen Workspace, src/engine/workspace.rs:13-19)."""

    def __init__(self, nodes, edges, sample_rate=44100, ticks_per_second=60, max_ticks_per_run=1, flags=0,
                 device=-1, stream=None):
        self._h = C.c_void_p()
        blobs = [params_bytes(p) for (_k, p) in nodes]
        self._keep = []
        n_arr = (Node * max(1, len(nodes)))()
        for i, ((kind, _p), blob) in enumerate(zip(nodes, blobs)):
            buf = C.create_string_buffer(blob, len(blob)) if blob else None
            self._keep.append(buf)
            n_arr[i] = Node(kind, len(blob), C.cast(buf, C.c_void_p) if buf else None)
        e_arr = (Edge * max(1, len(edges)))()
        for i, e in enumerate(edges):
            e_arr[i] = Edge(*e)
        opts = GraphOpts(sample_rate, ticks_per_second, max_ticks_per_run, flags, device, 0, stream)
        check(lib.mx_graph_build(n_arr, len(nodes), e_arr, len(edges), C.byref(opts), C.byref(self._h)))
        spt = C.c_size_t()
        check(lib.mx_graph_samples_per_tick(self._h, C.byref(spt)))
        self.spt = spt.value
        self.max_ticks = max_ticks_per_run
        self.n_nodes = len(nodes)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.mx_graph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_order(self):
        n = C.c_size_t()
        arr = (C.c_uint32 * max(1, self.n_nodes))()
        check(lib.mx_graph_run_order(self._h, arr, self.n_nodes, C.byref(n)))
        return list(arr[: n.value])

    def update_params(self, node, params):
        blob = params_bytes(params)
        check(lib.mx_graph_update_params(self._h, node, blob, len(blob)))

    def schedule_params(self, node, tick_in_run: int, params):
        """ModuleT::update at the boundary before tick `tick_in_run` of the next run (client_update between ticks)."""
        blob = params_bytes(params)
        check(lib.mx_graph_schedule_params(self._h, node, tick_in_run, blob, len(blob)))

    def schedule_params_batch(self, events: "C.Array", n: int | None = None):
        """events: a ctypes array of ParamEvent whose `params` pointers the caller keeps alive for the call."""
        check(lib.mx_graph_schedule_params_batch(self._h, events, len(events) if n is None else n))

    def eq_repair_stats(self) -> dict:
        """mx_graph_eq_repair_stats: what the proof / repair pass of the speculative EqThree did (counters since the graph was built)"""
        v = (C.c_uint64 * 8)()
        check(lib.mx_graph_eq_repair_stats(self._h, v))
        keys = ("chunks_run", "chunks_repaired", "settled_by_comparison", "walk_steps_16", "fill_steps_16", "island_rounds", "in_order_walks", "nan_fills")
        return {k: int(x) for k, x in zip(keys, v)}

    def debug_tail_releases(self):
        """-> (gated, at_once): how the held-back Mixer banks of the second-stream mode went out (mx_graph_debug_tail_releases)"""
        a, b = C.c_uint64(), C.c_uint64()
        check(lib.mx_graph_debug_tail_releases(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def debug_eq_records(self):
        """-> (device pointer, bytes) of the first EqThree group's chunk records of the last speculative launch (mx_graph_debug_eq_records)"""
        p, n = C.c_void_p(), C.c_size_t()
        check(lib.mx_graph_debug_eq_records(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def eq_spec_stats(self):
        """-> (chunks run, chunks repaired) of the speculative exact EqThree path since the graph was built."""
        a, b = C.c_uint64(), C.c_uint64()
        check(lib.mx_graph_eq_spec_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def write_source(self, node, samples: np.ndarray, n_ticks: int):
        a = np.ascontiguousarray(samples, dtype=np.float32)
        check(lib.mx_graph_write_source(self._h, node, a.ctypes.data_as(C.c_void_p), n_ticks))

    def bind_source_device(self, node, device_ptr: int):
        check(lib.mx_graph_bind_source_device(self._h, node, C.c_void_p(device_ptr)))

    def run_ticks(self, first_tick: int, n_ticks: int = 1):
        check(lib.mx_graph_run_ticks(self._h, first_tick, n_ticks))

    def sync(self):
        check(lib.mx_graph_sync(self._h))

    def read_output(self, node, port, n_ticks: int, stereo: bool, rate=(1, 1)) -> np.ndarray:
        """rate = (up, down) of the port's sample-rate domain (a Resample node's output is not at the graph's rate)."""
        out = np.empty(n_ticks * (self.spt * rate[0] // rate[1]) * (2 if stereo else 1), dtype=np.float32)
        check(lib.mx_graph_read_output(self._h, node, port, out.ctypes.data_as(C.c_void_p), n_ticks))
        return out

    def read_output_window(self, node, port, first_tick: int, n_ticks: int, stereo: bool) -> np.ndarray:
        """ticks [first_tick, first_tick + n_ticks) of the last run (ports in the graph's own rate domain)"""
        out = np.empty(n_ticks * self.spt * (2 if stereo else 1), dtype=np.float32)
        check(lib.mx_graph_read_output_window(self._h, node, port, out.ctypes.data_as(C.c_void_p), first_tick, n_ticks))
        return out

    def read_output_i16(self, node, port, n_ticks: int, stereo: bool, rate=(1, 1)) -> np.ndarray:
        out = np.empty(n_ticks * (self.spt * rate[0] // rate[1]) * (2 if stereo else 1), dtype=np.int16)
        check(lib.mx_graph_read_output_i16(self._h, node, port, out.ctypes.data_as(C.c_void_p), n_ticks))
        return out

    def write_source_i16(self, node, samples: np.ndarray, n_ticks: int):
        a = np.ascontiguousarray(samples, dtype=np.int16)
        check(lib.mx_graph_write_source_i16(self._h, node, a.ctypes.data_as(C.c_void_p), n_ticks))

    def output_device_ptr(self, node, port):
        p = C.c_void_p()
        n = C.c_size_t()
        check(lib.mx_graph_output_device_ptr(self._h, node, port, C.byref(p), C.byref(n)))
        return p.value, n.value

    def stream(self):
        """the hipStream_t the graph launches on"""
        p = C.c_void_p()
        check(lib.mx_graph_stream(self._h, C.byref(p)))
        return p.value

    def tail_stream(self):
        """MX_FLAG_OVERLAP_TAIL: the stream the last launch group runs on (None when the mode is off)."""
        p = C.c_void_p()
        check(lib.mx_graph_tail_stream(self._h, C.byref(p)))
        return p.value

    def read_plotter(self, node, tick_in_run):
        l = np.empty(self.spt, dtype=np.float32)
        r = np.empty(self.spt, dtype=np.float32)
        fired = C.c_int()
        check(lib.mx_graph_read_plotter(self._h, node, tick_in_run, l.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), C.byref(fired)))
        return (l, r) if fired.value else None

    def profile_run(self, first_tick, n_ticks):
        by_kind = (C.c_float * KIND_COUNT)()
        total = C.c_float()
        check(lib.mx_graph_profile_run(self._h, first_tick, n_ticks, by_kind, C.byref(total)))
        return {KIND_NAMES[k]: by_kind[k] for k in range(KIND_COUNT) if by_kind[k] > 0}, total.value


    def profile_enable(self, on: bool):
        check(lib.mx_graph_profile_enable(self._h, 1 if on else 0))

    def profile_collect(self):
        """-> ({kind_name: total ms}, total ms, n_runs) accumulated since profile_enable(True)."""
        by_kind = (C.c_float * KIND_COUNT)()
        total = C.c_float()
        n = C.c_uint32()
        check(lib.mx_graph_profile_collect(self._h, by_kind, C.byref(total), C.byref(n)))
        return {KIND_NAMES[k]: by_kind[k] for k in range(KIND_COUNT) if by_kind[k] > 0}, total.value, n.value


    def performance_info(self, n_nodes: int):
        """-> (PerformanceInfo, [module us per tick]) for the most recent profiled run (src/engine/timing.rs:46-60)."""
        info = PerformanceInfo()
        us = (C.c_uint64 * max(1, n_nodes))()
        check(lib.mx_graph_performance_info(self._h, C.byref(info), us, n_nodes))
        return info, list(us[:n_nodes])

    def adopt_state(self, old: "Graph", old_node_of_new):
        """Topology edit (src/engine.rs:277-398): take over the state of surviving modules; `old` must not run again."""
        arr = (C.c_int32 * max(1, len(old_node_of_new)))(*old_node_of_new)
        check(lib.mx_graph_adopt_state(self._h, old._h, arr, len(old_node_of_new)))


class PcmRing:
    """mx_pcm_ring_*: decoded i16 frames of any length re-blocked to ticks (src/module/stream_input.rs:92-124)."""

    def __init__(self):
        self._h = C.c_void_p()
        check(lib.mx_pcm_ring_create(C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.mx_pcm_ring_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def push(self, samples: np.ndarray):
        a = np.ascontiguousarray(samples, dtype=np.int16)
        check(lib.mx_pcm_ring_push_i16(self._h, a.ctypes.data_as(C.c_void_p), a.size))

    def queued(self) -> int:
        n = C.c_size_t()
        check(lib.mx_pcm_ring_queued(self._h, C.byref(n)))
        return n.value

    def feed(self, graph: "Graph", node: int, n_ticks: int) -> int:
        z = C.c_size_t()
        check(lib.mx_pcm_ring_feed(self._h, graph._h, node, n_ticks, C.byref(z)))
        return z.value


class Module:
    """mx_module_*: one ModuleT instance with host buffers (src/module/mod.rs:7-19)."""

    def __init__(self, kind, params=None, sample_rate=44100, ticks_per_second=60, flags=0):
        self._h = C.c_void_p()
        blob = params_bytes(params)
        opts = GraphOpts(sample_rate, ticks_per_second, 1, flags, -1, 0, None)
        check(lib.mx_module_create_ex(kind, blob, len(blob), C.byref(opts), C.byref(self._h)))
        self.kind = kind

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.mx_module_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, params):
        blob = params_bytes(params)
        check(lib.mx_module_update(self._h, blob, len(blob)))

    def run_tick(self, t: int, inputs, outputs, indication: np.ndarray | None = None):
        """inputs: list of (line_kind, np.float32 array | None); outputs: list of (line_kind, np.float32 array)."""
        ins = (Input * max(1, len(inputs)))()
        keep = []
        for i, (lk, arr) in enumerate(inputs):
            if lk == MX_DISCONNECTED or arr is None:
                ins[i] = Input(MX_DISCONNECTED, None, 0, None)
            else:
                a = np.ascontiguousarray(arr, dtype=np.float32)
                keep.append(a)
                ins[i] = Input(lk, a.ctypes.data_as(C.c_void_p), a.size, None)
        outs = (Output * max(1, len(outputs)))()
        for i, (lk, arr) in enumerate(outputs):
            assert arr.dtype == np.float32 and arr.flags.c_contiguous
            outs[i] = Output(lk, arr.ctypes.data_as(C.c_void_p), arr.size, None, 0)
        ind_len = C.c_size_t(indication.nbytes if indication is not None else 0)   # in: capacity, out: bytes written
        ind_ptr = indication.ctypes.data_as(C.c_void_p) if indication is not None else None
        check(lib.mx_module_run_tick(self._h, t, ins, len(inputs), outs, len(outputs), ind_ptr, C.byref(ind_len)))
        return ind_len.value
