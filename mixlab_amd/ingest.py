"""ctypes binding of the timed-ingest part of include/mixlab_gpu.h: MediaSource / StreamInput pacing and the frame staging ring
(plumbing for tests; the host mirror of src/module/media_source.rs and src/module/stream_input.rs)."""
from __future__ import annotations

import ctypes as C
from fractions import Fraction

import numpy as np

from . import abi
from .abi import check, lib
from .video import DFrame, VideoInput, _host_frame, PIXFMT_YUV420P, PIXFMT_NV12

MX_ERR_FULL = -6
NO_VIDEO_NODE = 0xFFFFFFFF


def _proto(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)


_I64 = C.c_int64
_proto("mx_graph_queue_video_source", C.c_int, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, _I64, _I64, _I64, _I64)
_proto("mx_media_source_create", C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p))
_proto("mx_media_source_destroy", None, C.c_void_p)
_proto("mx_media_source_set_media", C.c_int, C.c_void_p, C.c_int)
_proto("mx_media_source_send", C.c_int, C.c_void_p, C.c_void_p, _I64, _I64, _I64, _I64)
_proto("mx_media_source_run_tick", C.c_int, C.c_void_p, C.c_uint64, C.POINTER(VideoInput))
_proto("mx_media_source_feed", C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32)
_proto("mx_stream_input_create", C.c_int, C.c_uint32, C.POINTER(C.c_void_p))
_proto("mx_stream_input_destroy", None, C.c_void_p)
_proto("mx_stream_input_listen", C.c_int, C.c_void_p, C.c_int)
_proto("mx_stream_input_write_audio", C.c_int, C.c_void_p, C.c_uint64, _I64, _I64, C.c_void_p, C.c_size_t)
_proto("mx_stream_input_write_video", C.c_int, C.c_void_p, C.c_uint64, _I64, _I64, C.c_void_p, _I64, _I64)
_proto("mx_stream_input_run_tick", C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t, C.POINTER(VideoInput), C.POINTER(C.c_size_t))
_proto("mx_stream_input_feed", C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_size_t))
_proto("mx_frame_stager_create", C.c_int, C.c_uint32, C.POINTER(C.c_void_p))
_proto("mx_frame_stager_destroy", None, C.c_void_p)
_proto("mx_frame_stager_upload", C.c_int, C.c_void_p, C.POINTER(abi.Frame), C.c_int, C.POINTER(C.c_void_p))
_proto("mx_frame_stager_acquire", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(abi.Frame), C.POINTER(C.c_uint32))
_proto("mx_frame_stager_commit", C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p))
_proto("mx_frame_stager_fence", C.c_int, C.c_void_p, C.c_void_p)
_proto("mx_frame_stager_fence_graph", C.c_int, C.c_void_p, C.c_void_p)
_proto("mx_frame_stager_sync", C.c_int, C.c_void_p)


class MonitorTick(C.Structure):
    _fields_ = [("video_present", C.c_int32), ("ts_num", _I64), ("ts_den", _I64), ("frame_ts_num", _I64), ("frame_ts_den", _I64),
                ("dur_num", _I64), ("dur_den", _I64), ("dropped", C.c_int32), ("_pad", C.c_int32)]


_proto("mx_graph_monitor_consume", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32)
_proto("mx_graph_read_monitor_tick", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(MonitorTick), C.POINTER(C.c_void_p))
_proto("mx_graph_read_monitor_audio_i16", C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32)


def graph_monitor_consume(g, node, n_ticks):
    """queue_depth > 0: the consumer took n_ticks ticks off the node's queue"""
    check(lib.mx_graph_monitor_consume(g._h, node, n_ticks))


def graph_read_monitor_tick(g, node, tick_in_run, with_dropped=False):
    """-> (ts, None) or (ts, (DFrame, frame_ts, dur)) with exact Fractions; with_dropped: (ts, ..., dropped) -- the tick found the queue full"""
    info, h = MonitorTick(), C.c_void_p()
    check(lib.mx_graph_read_monitor_tick(g._h, node, tick_in_run, C.byref(info), C.byref(h)))
    ts = Fraction(info.ts_num, info.ts_den)
    if with_dropped:
        vid = None if not info.video_present else (DFrame(handle=h.value), Fraction(info.frame_ts_num, info.frame_ts_den), Fraction(info.dur_num, info.dur_den))
        return ts, vid, bool(info.dropped)
    if not info.video_present:
        return ts, None
    return ts, (DFrame(handle=h.value), Fraction(info.frame_ts_num, info.frame_ts_den), Fraction(info.dur_num, info.dur_den))


class MonitorLayout(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("frame_bytes", C.c_size_t), ("plane_offset", C.c_size_t * 3), ("stride", C.c_int32 * 3)]


_proto("mx_graph_monitor_layout", C.c_int, C.c_void_p, C.c_uint32, C.POINTER(MonitorLayout))
_proto("mx_graph_read_monitor_video", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p)


_proto("mx_host_alloc", C.c_int, C.c_size_t, C.POINTER(C.c_void_p))
_proto("mx_host_free", None, C.c_void_p)


class PinnedBuffer:
    """page-locked host bytes as a numpy array (.a)"""

    def __init__(self, nbytes):
        self._p = C.c_void_p()
        check(lib.mx_host_alloc(nbytes, C.byref(self._p)))
        self.a = np.frombuffer((C.c_uint8 * nbytes).from_address(self._p.value), np.uint8)

    def __del__(self):
        if getattr(self, "_p", None) is not None and self._p.value:
            self.a = None
            lib.mx_host_free(self._p)
            self._p = C.c_void_p()


def graph_monitor_layout(g, node) -> MonitorLayout:
    l = MonitorLayout()
    check(lib.mx_graph_monitor_layout(g._h, node, C.byref(l)))
    return l


def graph_read_monitor_video(g, node, first_tick, n_ticks, out=None):
    """-> list of None or [Y, U, V] visible planes (views into one packed read-back), one entry per tick; `out`: uint8 array of
    n_ticks * frame_bytes to read into (a PinnedBuffer's for speed)"""
    l = graph_monitor_layout(g, node)
    buf = np.empty(n_ticks * l.frame_bytes, np.uint8) if out is None else out
    assert buf.size >= n_ticks * l.frame_bytes
    present = np.zeros(n_ticks, np.uint8)
    check(lib.mx_graph_read_monitor_video(g._h, node, first_tick, n_ticks, buf.ctypes.data_as(C.c_void_p), present.ctypes.data_as(C.c_void_p)))
    out = []
    for k in range(n_ticks):
        if not present[k]:
            out.append(None); continue
        fr = buf[k * l.frame_bytes:(k + 1) * l.frame_bytes]
        planes = []
        for p in range(3):
            rows, w = (l.height, l.width) if p == 0 else (l.height >> 1, l.width >> 1)
            planes.append(fr[l.plane_offset[p]: l.plane_offset[p] + rows * l.stride[p]].reshape(rows, l.stride[p])[:, :w])
        out.append(planes)
    return out


def graph_read_monitor_audio_i16(g, node, n_ticks, spt):
    out = np.empty(n_ticks * 2 * spt, np.int16)
    check(lib.mx_graph_read_monitor_audio_i16(g._h, node, out.ctypes.data_as(C.c_void_p), n_ticks))
    return out


def _q(x) -> tuple[int, int]:
    f = Fraction(x) if not isinstance(x, tuple) else Fraction(x[0], x[1])
    return f.numerator, f.denominator


def _tick_video(v: VideoInput):
    """-> None or (DFrame owning the returned reference, duration_hint, tick_offset) with exact Fractions"""
    if not v.frame:
        return None
    return DFrame(handle=v.frame), Fraction(v.dur_num, v.dur_den), Fraction(v.off_num, v.off_den)


def graph_queue_video_source(g, node, tick, frame, dur=(1, 60), off=(0, 1)):
    d, o = _q(dur), _q(off)
    check(lib.mx_graph_queue_video_source(g._h, node, tick, frame.handle, d[0], d[1], o[0], o[1]))


class _Handle:
    _destroy = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            getattr(lib, self._destroy)(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MediaSource(_Handle):
    """MediaSource::run_tick (src/module/media_source.rs:93-126)."""
    _destroy = "mx_media_source_destroy"

    def __init__(self, sample_rate=44100, ticks_per_second=60):
        self._h = C.c_void_p()
        check(lib.mx_media_source_create(sample_rate, ticks_per_second, C.byref(self._h)))

    def set_media(self, present=True):
        check(lib.mx_media_source_set_media(self._h, 1 if present else 0))

    def send(self, frame: DFrame, pts, dur) -> bool:
        """False = the channel holds two frames already (the reference's decode thread blocks there)"""
        p, d = _q(pts), _q(dur)
        rc = lib.mx_media_source_send(self._h, frame.handle, p[0], p[1], d[0], d[1])
        if rc == MX_ERR_FULL:
            return False
        check(rc)
        return True

    def run_tick(self, t: int):
        v = VideoInput()
        check(lib.mx_media_source_run_tick(self._h, t, C.byref(v)))
        return _tick_video(v)

    def feed(self, graph, node, first_tick, n_ticks):
        check(lib.mx_media_source_feed(self._h, graph._h, node, first_tick, n_ticks))


class StreamInput(_Handle):
    """StreamInput::run_tick (src/module/stream_input.rs:72-147)."""
    _destroy = "mx_stream_input_destroy"

    def __init__(self, sample_rate=44100):
        self._h = C.c_void_p()
        check(lib.mx_stream_input_create(sample_rate, C.byref(self._h)))

    def listen(self, listening=True):
        check(lib.mx_stream_input_listen(self._h, 1 if listening else 0))

    def write_audio(self, source_id, source_time, samples) -> bool:
        a = np.ascontiguousarray(samples, dtype=np.int16)
        t = _q(source_time)
        rc = lib.mx_stream_input_write_audio(self._h, source_id, t[0], t[1], a.ctypes.data_as(C.c_void_p), a.size)
        if rc == MX_ERR_FULL:
            return False
        check(rc)
        return True

    def write_video(self, source_id, source_time, frame: DFrame, dur) -> bool:
        t, d = _q(source_time), _q(dur)
        rc = lib.mx_stream_input_write_video(self._h, source_id, t[0], t[1], frame.handle, d[0], d[1])
        if rc == MX_ERR_FULL:
            return False
        check(rc)
        return True

    def run_tick(self, t: int, n_out: int):
        """-> (i16 samples [n_out], None or (frame, dur, off), samples zero-filled)"""
        out = np.empty(n_out, np.int16)
        v, z = VideoInput(), C.c_size_t()
        check(lib.mx_stream_input_run_tick(self._h, t, out.ctypes.data_as(C.c_void_p), n_out, C.byref(v), C.byref(z)))
        return out, _tick_video(v), z.value

    def feed(self, graph, audio_node, video_node, first_tick, n_ticks) -> int:
        z = C.c_size_t()
        check(lib.mx_stream_input_feed(self._h, graph._h, audio_node, NO_VIDEO_NODE if video_node is None else video_node,
                                       first_tick, n_ticks, C.byref(z)))
        return z.value


class FrameStager(_Handle):
    """Page-locked H2D staging ring for decoded frames."""
    _destroy = "mx_frame_stager_destroy"

    def __init__(self, slots=4):
        self._h = C.c_void_p()
        check(lib.mx_frame_stager_create(slots, C.byref(self._h)))

    def upload(self, planes, width, height, fmt=PIXFMT_YUV420P) -> DFrame:
        # 10-bit formats: uint16 planes (little-endian words) travel as rows of bytes
        ps = [np.ascontiguousarray(a, dtype="<u2").view(np.uint8) if np.asarray(a).dtype.itemsize == 2 else np.ascontiguousarray(a, dtype=np.uint8) for a in planes]
        assert len(ps) == (2 if fmt in (PIXFMT_NV12, 13, 20) else 3)
        hf = _host_frame(ps, width, height)
        h = C.c_void_p()
        check(lib.mx_frame_stager_upload(self._h, C.byref(hf), fmt, C.byref(h)))
        return DFrame(handle=h.value)

    def acquire(self, width, height, fmt=PIXFMT_YUV420P):
        """-> (ticket, [numpy views of the slot's planes, rows x stride bytes]): write the picture into them, then commit(ticket)"""
        hf, ticket = abi.Frame(), C.c_uint32()
        check(lib.mx_frame_stager_acquire(self._h, width, height, fmt, C.byref(hf), C.byref(ticket)))
        cw, ch = (0 if fmt in (2, 8, 12, 16, 19) else (2 if fmt in (6, 7) else 1)), (1 if fmt in (0, 3, 8, 10, 13, 14, 17, 20) else (2 if fmt == 6 else 0))
        views = []
        for p in range(2 if fmt in (PIXFMT_NV12, 13, 20) else 3):
            rows = height if p == 0 else height >> ch
            buf = (C.c_uint8 * (rows * hf.stride[p])).from_address(hf.data[p])
            views.append(np.frombuffer(buf, np.uint8).reshape(rows, hf.stride[p]))
        return ticket.value, views

    def commit(self, ticket) -> DFrame:
        h = C.c_void_p()
        check(lib.mx_frame_stager_commit(self._h, ticket, C.byref(h)))
        return DFrame(handle=h.value)

    def fence(self, stream=None):
        check(lib.mx_frame_stager_fence(self._h, stream))

    def fence_graph(self, graph):
        check(lib.mx_frame_stager_fence_graph(self._h, graph._h))

    def sync(self):
        check(lib.mx_frame_stager_sync(self._h))
