"""ctypes binding of the pixel half of include/mixlab_gpu.h (plumbing for tests and bench.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi
from .abi import check, lib


class VideoMixerParams(C.Structure):
    _fields_ = [("a", C.c_int32), ("b", C.c_int32), ("fader", C.c_double)]   # -1 = None (protocol/src/lib.rs:405-410)


class VideoInput(C.Structure):
    _fields_ = [("frame", C.c_void_p), ("dur_num", C.c_int64), ("dur_den", C.c_int64), ("off_num", C.c_int64), ("off_den", C.c_int64)]


def _proto(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)


_proto("mx_dframe_create", C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p))
_proto("mx_dframe_create_fmt", C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(C.c_void_p))
_proto("mx_dframe_format", C.c_int, C.c_void_p, C.POINTER(C.c_int))
_proto("mx_dframe_retain", C.c_int, C.c_void_p)
_proto("mx_dframe_release", None, C.c_void_p)
_proto("mx_dframe_upload", C.c_int, C.c_void_p, C.POINTER(abi.Frame), C.c_void_p)
_proto("mx_dframe_download", C.c_int, C.c_void_p, C.POINTER(abi.Frame), C.c_void_p)
_proto("mx_dframe_planes", C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p * 3), C.POINTER(C.c_int32 * 3))
_proto("mx_dframe_upload_alpha", C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)
_proto("mx_dframe_download_alpha", C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)
_proto("mx_dframe_alpha_plane", C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32))
_proto("mx_video_blank", C.c_int, C.c_void_p, C.c_void_p)
_proto("mx_video_crossfade", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p)
_proto("mx_video_scale", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
_proto("mx_video_scale_band", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p)
_proto("mx_video_scaler_create", C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p))
_proto("mx_video_scaler_scale", C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p))
_proto("mx_video_scaler_destroy", None, C.c_void_p)
_proto("mx_video_scale_geometry", C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
       C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))
_proto("mx_video_to_rgba", C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p)
_proto("mx_video_sync", C.c_int, C.c_void_p)
_proto("mx_stream_retired", C.c_int, C.c_void_p)
_proto("mx_video_mixer_create", C.c_int, C.POINTER(VideoMixerParams), C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p))
_proto("mx_video_mixer_update", C.c_int, C.c_void_p, C.POINTER(VideoMixerParams))
_proto("mx_video_mixer_run_tick", C.c_int, C.c_void_p, C.c_uint64, C.POINTER(VideoInput),
       C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p))
_proto("mx_video_mixer_sync", C.c_int, C.c_void_p)
_proto("mx_video_mixer_destroy", None, C.c_void_p)
_proto("mx_device_alloc", C.c_int, C.c_size_t, C.POINTER(C.c_void_p))
_proto("mx_device_free", None, C.c_void_p)
_proto("mx_device_download", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


_proto("mx_graph_set_video_source", C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int)
_proto("mx_graph_set_video_source_ring", C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.c_size_t, C.c_int64, C.c_int64, C.c_int64, C.c_int64)
_proto("mx_graph_video_output", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p))
_proto("mx_graph_rgba_output", C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))


class VideoToRgbaParams(C.Structure):
    _fields_ = [("use_matrix", C.c_int32), ("matrix_q12", C.c_int32 * 12)]


def to_rgba_params(matrix_q12=None) -> VideoToRgbaParams:
    p = VideoToRgbaParams()
    p.use_matrix = 1 if matrix_q12 is not None else 0
    for k in range(12):
        p.matrix_q12[k] = int(matrix_q12[k]) if matrix_q12 is not None else 0
    return p


# ---- video nodes of an abi.Graph ----
def graph_set_video_source(g, node, frame, dur=(1, 60), off=(0, 1), repeat=False):
    check(lib.mx_graph_set_video_source(g._h, node, frame.handle if frame is not None else None, dur[0], dur[1], off[0], off[1], 1 if repeat else 0))


_proto("mx_graph_set_video_source_band", C.c_int, C.c_void_p, *([C.c_uint32] * 9))


def graph_set_video_source_band(g, node, in_w, in_full_h, src_row0, slice_rows, full_w, full_h, row0, band_rows):
    """The source's frames are halo slices of a smaller layer; it delivers this rank's row band of the layer's letterboxed scale."""
    check(lib.mx_graph_set_video_source_band(g._h, node, in_w, in_full_h, src_row0, slice_rows, full_w, full_h, row0, band_rows))


def graph_set_video_source_ring(g, node, frames, dur=(1, 60), off=(0, 1)):
    """A new frame on every tick, cycling through `frames` (a decoder's stream)."""
    arr = (C.c_void_p * max(1, len(frames)))(*[f.handle for f in frames])
    check(lib.mx_graph_set_video_source_ring(g._h, node, arr, len(frames), dur[0], dur[1], off[0], off[1]))


def graph_video_output(g, node, port=0):
    h = C.c_void_p()
    check(lib.mx_graph_video_output(g._h, node, port, C.byref(h)))
    return DFrame(handle=h.value) if h.value else None


def graph_rgba_output(g, node, stream=None):
    """-> HxWx4 uint8 array of the last tick's RGBA frame, or None."""
    p, stride, w, h = C.c_void_p(), C.c_int32(), C.c_uint32(), C.c_uint32()
    check(lib.mx_graph_rgba_output(g._h, node, C.byref(p), C.byref(stride), C.byref(w), C.byref(h)))
    if not w.value:
        return None
    g.sync()
    out = np.empty(stride.value * h.value, np.uint8)
    check(lib.mx_device_download(out.ctypes.data_as(C.c_void_p), p, out.size, stream))
    return out.reshape(h.value, stride.value)[:, : w.value * 4].reshape(h.value, w.value, 4)


def _host_frame(planes, w, h):
    """abi.Frame over three contiguous uint8 plane arrays (rows = plane height, stride = array row length)."""
    f = abi.Frame()
    f.width, f.height = w, h
    for p in range(len(planes)):
        a = planes[p]
        assert a.dtype == np.uint8 and a.flags.c_contiguous and a.ndim == 2
        f.data[p] = a.ctypes.data
        f.stride[p] = a.shape[1]
    return f


PIXFMT_YUV420P, PIXFMT_YUV422P, PIXFMT_YUV444P, PIXFMT_NV12 = 0, 1, 2, 3   # mx_pixfmt (nv12: plane 1 = interleaved U,V, no plane 2)
PIXFMT_RGB24, PIXFMT_BGRA = 4, 5                                            # packed RGB, one plane: scaler inputs only
PIXFMT_YUV410P, PIXFMT_YUV411P, PIXFMT_YUV440P, PIXFMT_GRAY8 = 6, 7, 8, 9    # chroma 1/4 x 1/4, 1/4 x 1, 1 x 1/2; one luma plane (stands for yuv444p with U = V = 0x80)


PIXFMT_YUV420P10, PIXFMT_YUV422P10, PIXFMT_YUV444P10, PIXFMT_P010 = 10, 11, 12, 13   # 10-bit samples in 16-bit LE words (p010: semi-planar, value in the high bits): scaler inputs only
PIXFMT_YUV420P12, PIXFMT_YUV422P12, PIXFMT_YUV444P12 = 14, 15, 16                    # 12 bits, low-aligned
PIXFMT_YUV420P16, PIXFMT_YUV422P16, PIXFMT_YUV444P16, PIXFMT_P016 = 17, 18, 19, 20   # all 16 bits (p016: semi-planar)
# deep format -> (8-bit format of the layout, bits, shift of the value inside a word)
DEEP = {10: (0, 10, 0), 11: (1, 10, 0), 12: (2, 10, 0), 13: (0, 10, 6), 14: (0, 12, 0), 15: (1, 12, 0), 16: (2, 12, 0), 17: (0, 16, 0), 18: (1, 16, 0), 19: (2, 16, 0), 20: (0, 16, 0)}
PIXFMT_YUYV422, PIXFMT_UYVY422 = 21, 22                                              # packed 4:2:2, one plane of 2 bytes per pixel: scaler inputs only
_DEEP = tuple(DEEP)
PIXFMT_BGR24, PIXFMT_RGBA, PIXFMT_ARGB, PIXFMT_ABGR = 23, 24, 25, 26                  # the other byte orders of packed RGB
_PACKED_BPP = {PIXFMT_RGB24: 3, PIXFMT_BGRA: 4, PIXFMT_GRAY8: 1, PIXFMT_YUYV422: 2, PIXFMT_UYVY422: 2, PIXFMT_BGR24: 3, PIXFMT_RGBA: 4, PIXFMT_ARGB: 4, PIXFMT_ABGR: 4}
_SEMI = (PIXFMT_NV12, PIXFMT_P010, PIXFMT_P016)
PIXFMT_YUVA420P = 27   # yuv420p + a coverage plane (per-pixel alpha, build-specified): upload() takes the three YUV planes, upload_alpha() the fourth


class DFrame:
    """Device-resident planar YUV frame (one reference owned by this object); yuv420p unless `fmt` says otherwise."""

    def __init__(self, width=None, height=None, stream=None, handle=None, fmt=PIXFMT_YUV420P):
        self.stream = stream
        if handle is not None:
            self._h = C.c_void_p(handle)
        else:
            self._h = C.c_void_p()
            check(lib.mx_dframe_create_fmt(width, height, fmt, stream, C.byref(self._h)))
        f = C.c_int()
        check(lib.mx_dframe_format(self._h, C.byref(f)))
        self.fmt = f.value
        lay = DEEP[self.fmt][0] if self.fmt in DEEP else (PIXFMT_YUV420P if self.fmt == PIXFMT_YUVA420P else self.fmt)
        self.cw = 0 if lay in (PIXFMT_YUV444P, PIXFMT_YUV440P) else (2 if lay in (PIXFMT_YUV410P, PIXFMT_YUV411P) else 1)
        self.ch = 1 if lay in (PIXFMT_YUV420P, PIXFMT_NV12, PIXFMT_YUV440P) else (2 if lay == PIXFMT_YUV410P else 0)
        w, h = C.c_uint32(), C.c_uint32()
        self._data = (C.c_void_p * 3)()
        self._stride = (C.c_int32 * 3)()
        check(lib.mx_dframe_planes(self._h, C.byref(w), C.byref(h), C.byref(self._data), C.byref(self._stride)))
        self.width, self.height = w.value, h.value

    @property
    def handle(self):
        return self._h.value

    def release(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.mx_dframe_release(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def upload(self, y, u, v=None):
        """planar formats: (y, u, v); nv12: (y, uv) with uv the interleaved chroma rows of `width` bytes"""
        semi = self.fmt in _SEMI
        if self.fmt in _DEEP:   # 16-bit little-endian words, handed over as rows of bytes
            planes = [np.ascontiguousarray(a, dtype="<u2").view(np.uint8) for a in ((y, u) if semi else (y, u, v))]
        else:
            planes = [np.ascontiguousarray(a, dtype=np.uint8) for a in ((y, u) if semi else (y, u, v))]
        hf = _host_frame(planes, self.width, self.height)
        check(lib.mx_dframe_upload(self._h, C.byref(hf), self.stream))
        return self

    def upload_packed(self, pix):
        """packed RGB formats: pix (height, width, 3) rgb24 / (height, width, 4) bgra"""
        a = np.ascontiguousarray(pix, dtype=np.uint8)
        if self.fmt == PIXFMT_GRAY8:
            a = a.reshape(self.height, self.width, 1)
        if self.fmt in (PIXFMT_YUYV422, PIXFMT_UYVY422):
            a = a.reshape(self.height, self.width, 2)
        assert self.fmt in _PACKED_BPP and a.shape == (self.height, self.width, _PACKED_BPP[self.fmt])
        plane = a.reshape(self.height, -1)
        hf = _host_frame([plane], self.width, self.height)
        check(lib.mx_dframe_upload(self._h, C.byref(hf), self.stream))
        return self

    def download(self):
        if self.fmt in _PACKED_BPP:
            bpp = _PACKED_BPP[self.fmt]
            plane = np.empty((self.height, self.width * bpp), np.uint8)
            hf = _host_frame([plane], self.width, self.height)
            check(lib.mx_dframe_download(self._h, C.byref(hf), self.stream))
            return [plane.reshape(self.height, self.width, bpp)]
        bps = 2 if self.fmt in _DEEP else 1
        if self.fmt in _SEMI:
            planes = [np.empty((self.height, self.width * bps), np.uint8), np.empty((self.height >> 1, self.width * bps), np.uint8)]
        else:
            planes = [np.empty((self.height >> (self.ch if p else 0), (self.width >> (self.cw if p else 0)) * bps), np.uint8) for p in range(3)]
        hf = _host_frame(planes, self.width, self.height)
        check(lib.mx_dframe_download(self._h, C.byref(hf), self.stream))
        return [a.view("<u2") for a in planes] if bps == 2 else planes

    def has_alpha(self) -> bool:
        p, st = C.c_void_p(), C.c_int32()
        check(lib.mx_dframe_alpha_plane(self._h, C.byref(p), C.byref(st)))
        return bool(p.value)

    def upload_alpha(self, alpha):
        """yuva420p: the coverage plane, (height, width) uint8, 255 = opaque"""
        a = np.ascontiguousarray(alpha, dtype=np.uint8)
        assert a.shape == (self.height, self.width)
        check(lib.mx_dframe_upload_alpha(self._h, a.ctypes.data_as(C.c_void_p), self.width, self.stream))
        return self

    def download_alpha(self):
        a = np.empty((self.height, self.width), np.uint8)
        check(lib.mx_dframe_download_alpha(self._h, a.ctypes.data_as(C.c_void_p), self.width, self.stream))
        return a

    def device_planes(self):
        return [self._data[p] for p in range(3)], [self._stride[p] for p in range(3)]


def stream_retired(stream):
    """a caller-owned stream is going away: the library drops what it keeps for it (mx_stream_retired)"""
    check(lib.mx_stream_retired(stream))


def blank(f: DFrame, stream=None):
    check(lib.mx_video_blank(f._h, stream))


def crossfade(out: DFrame, a: DFrame | None, b: DFrame | None, fader: float, stream=None):
    check(lib.mx_video_crossfade(out._h, a._h if a else None, b._h if b else None, fader, stream))


def scale(src: DFrame, dst: DFrame, stream=None):
    check(lib.mx_video_scale(src._h, dst._h, stream))


def scale_band(slice_: DFrame, in_full_h: int, src_row0: int, out_band: DFrame, full_w: int, full_h: int, row0: int, stream=None):
    """Rows [row0, row0 + out_band.height) of the letterboxed (full_w x full_h) scale, from a slice of the source (mx_video_scale_band)."""
    check(lib.mx_video_scale_band(slice_._h, in_full_h, src_row0, out_band._h, full_w, full_h, row0, stream))


class Scaler:
    """mx_video_scaler_*: DynamicScaler (src/video/encode.rs:338-397) -- context kept across calls."""

    def __init__(self, out_w, out_h, stream=None):
        self._h = C.c_void_p()
        check(lib.mx_video_scaler_create(out_w, out_h, stream, C.byref(self._h)))

    def scale(self, src: "DFrame") -> "DFrame":
        h = C.c_void_p()
        check(lib.mx_video_scaler_scale(self._h, src._h, C.byref(h)))
        return DFrame(handle=h.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.mx_video_scaler_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


def scale_geometry(in_w, in_h, out_w, out_h):
    v = [C.c_uint32() for _ in range(4)]
    check(lib.mx_video_scale_geometry(in_w, in_h, out_w, out_h, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def sync(stream=None):
    check(lib.mx_video_sync(stream))


class DeviceBuffer:
    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = nbytes
        check(lib.mx_device_alloc(nbytes, C.byref(self.ptr)))

    def download(self, stream=None) -> np.ndarray:
        out = np.empty(self.nbytes, np.uint8)
        check(lib.mx_device_download(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes, stream))
        return out

    def __del__(self):
        if getattr(self, "ptr", None) is not None and self.ptr.value:
            lib.mx_device_free(self.ptr)
            self.ptr = C.c_void_p()


def to_rgba(f: DFrame, matrix_q12=None, stream=None, out: DeviceBuffer | None = None, download=True):
    stride = ((f.width * 4 + 15) // 16) * 16
    buf = out or DeviceBuffer(stride * f.height)
    m = (C.c_int32 * 12)(*matrix_q12) if matrix_q12 is not None else None
    check(lib.mx_video_to_rgba(f._h, buf.ptr, stride, m, stream))
    if not download:
        return buf
    return buf.download(stream).reshape(f.height, stride)[:, : f.width * 4].reshape(f.height, f.width, 4)


class VideoMixer:
    """mx_video_mixer_*: VideoMixer::run_tick on device-resident frames (src/module/video_mixer.rs)."""

    def __init__(self, a=None, b=None, fader=1.0, sample_rate=44100, stream=None):
        self._h = C.c_void_p()
        p = VideoMixerParams(-1 if a is None else a, -1 if b is None else b, fader)
        check(lib.mx_video_mixer_create(C.byref(p), sample_rate, stream, C.byref(self._h)))

    def update(self, a=None, b=None, fader=1.0):
        p = VideoMixerParams(-1 if a is None else a, -1 if b is None else b, fader)
        check(lib.mx_video_mixer_update(self._h, C.byref(p)))

    def run_tick(self, t, inputs):
        """inputs: 4 entries, each None or (DFrame, (dur_num, dur_den), (off_num, off_den)).
        Returns (program, a, b) as DFrame or None (each owns one reference)."""
        arr = (VideoInput * 4)()
        for i in range(4):
            e = inputs[i] if i < len(inputs) else None
            if e is None:
                arr[i] = VideoInput(None, 0, 1, 0, 1)
            else:
                fr, dur, off = e
                arr[i] = VideoInput(fr.handle, dur[0], dur[1], off[0], off[1])
        o, a, b = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib.mx_video_mixer_run_tick(self._h, t, arr, C.byref(o), C.byref(a), C.byref(b)))
        return tuple(DFrame(handle=x.value) if x.value else None for x in (o, a, b))

    def sync(self):
        check(lib.mx_video_mixer_sync(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.mx_video_mixer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
