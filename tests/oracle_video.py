"""ctypes wrapper of the oracle's pixel path -- TEST INFRASTRUCTURE (the checker), never the product."""
from __future__ import annotations

import ctypes as C

import numpy as np

from oracle import OFrame, Rational, ScaleGeometry, lib


class VMixer(C.Structure):
    _fields_ = [("a", C.c_int32), ("b", C.c_int32), ("fader", C.c_double), ("sample_rate", C.c_uint32),
                ("has_stored", C.c_int * 4), ("stored", OFrame * 4), ("active_until", Rational * 4),
                ("has_scaler", C.c_int * 4), ("scaler_w", C.c_uint32 * 4), ("scaler_h", C.c_uint32 * 4)]


class VInput(C.Structure):
    _fields_ = [("frame", C.POINTER(OFrame)), ("duration_hint", Rational), ("tick_offset", Rational)]


lib.orc_video_mixer_init.argtypes = [C.POINTER(VMixer), C.c_int32, C.c_int32, C.c_double, C.c_uint32]
lib.orc_video_mixer_free.argtypes = [C.POINTER(VMixer)]
lib.orc_video_mixer_run_tick.argtypes = [C.POINTER(VMixer), C.c_uint64, C.POINTER(VInput), C.POINTER(OFrame), C.POINTER(C.c_int)]
lib.orc_video_crossfade.argtypes = [C.POINTER(OFrame), C.POINTER(OFrame), C.POINTER(OFrame), C.c_uint8]
lib.orc_frame_blank.argtypes = [C.POINTER(OFrame)]
lib.orc_dynamic_scale.argtypes = [C.POINTER(OFrame), C.POINTER(OFrame)]
lib.orc_scaler_geometry.argtypes = [C.c_uint32] * 4 + [C.POINTER(ScaleGeometry)]
lib.orc_unify_picture_settings.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)] * 2
lib.orc_yuv420_to_rgba.argtypes = [C.POINTER(OFrame), C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
lib.orc_deep_to_8.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.c_int, C.POINTER(OFrame)]
lib.orc_yuyv_to_422p.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(OFrame)]
lib.orc_packed_rgb_to_yuv444.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(OFrame)]
lib.orc_bicubic_tap_count.argtypes = [C.c_uint32, C.c_uint32]
lib.orc_bicubic_tap_count.restype = C.c_uint32


def align(v, a):
    return (v + a - 1) // a * a


class HostFrame:
    """YUV frame (fmt 0 yuv420p -- the default and everything a VideoMixer produces --, 1 yuv422p, 2 yuv444p, 3 nv12: plane 1 = interleaved
    U,V, no plane 2) in host memory with 64-byte-aligned strides (padding zero-initialised)."""

    def __init__(self, w, h, fmt=0):
        self.w, self.h, self.fmt = w, h, fmt
        self.cw, self.ch = (0 if fmt in (2, 8) else (2 if fmt in (6, 7) else 1)), (1 if fmt in (0, 3, 8) else (2 if fmt == 6 else 0))
        if fmt == 3:
            self.planes = [np.zeros((h, align(w, 64)), np.uint8), np.zeros((h >> 1, align(w, 64)), np.uint8)]
        else:
            self.planes = [np.zeros((h >> (self.ch if p else 0), align(w >> (self.cw if p else 0), 64)), np.uint8) for p in range(3)]
        self.c = OFrame()
        self.c.width, self.c.height, self.c.fmt = w, h, fmt
        for p in range(len(self.planes)):
            self.c.data[p] = self.planes[p].ctypes.data
            self.c.stride[p] = self.planes[p].shape[1]

    def set_alpha(self, alpha=None):
        """attach a coverage plane (build-specified per-pixel alpha): (h, w) uint8, default opaque"""
        self.alpha = np.full((self.h, align(self.w, 64)), 255, np.uint8)
        if alpha is not None:
            self.alpha[:, : self.w] = alpha
        self.c.alpha = self.alpha.ctypes.data
        self.c.alpha_stride = self.alpha.shape[1]
        return self

    def visible_alpha(self):
        return self.alpha[:, : self.w]

    def visible(self):
        if self.fmt == 3:
            return [self.planes[0][:, : self.w], self.planes[1][:, : self.w]]
        return [self.planes[p][:, : self.w >> (self.cw if p else 0)] for p in range(3)]

    def fill(self, layer, seed=0):
        """SURVEY.md section 8d config 4 pattern: Y(x,y) = (x + 2y + 31*layer + LCG noise) mod 256, U/V similar at chroma resolution."""
        import synth
        if self.fmt == 3:
            y, u, v = synth.yuv_pattern(self.w, self.h, layer, seed, 0)
            self.planes[0][:, : self.w] = y
            self.planes[1][:, 0: self.w: 2] = u
            self.planes[1][:, 1: self.w: 2] = v
            return self
        for p, a in enumerate(synth.yuv_pattern(self.w, self.h, layer, seed, self.fmt)):
            self.planes[p][:, : a.shape[1]] = a
        return self


def blank(f: HostFrame):
    lib.orc_frame_blank(C.byref(f.c))


def crossfade(out: HostFrame, a: HostFrame | None, b: HostFrame | None, fader: float):
    fade = lib.orc_crossfade_factor(fader)
    lib.orc_video_crossfade(C.byref(out.c), C.byref(a.c) if a else None, C.byref(b.c) if b else None, fade)


def dynamic_scale(src: HostFrame, dst: HostFrame):
    lib.orc_dynamic_scale(C.byref(src.c), C.byref(dst.c))


def packed_rgb_to_yuv444(pix: np.ndarray, fmt: int) -> HostFrame:
    """pix: (h, w, 3) rgb24 / bgr24 (fmt 4 / 23) or (h, w, 4) bgra / rgba / argb / abgr (5 / 24 / 25 / 26) uint8 -> the yuv444p frame a scaler input of that format stands for"""
    a = np.ascontiguousarray(pix, dtype=np.uint8)
    h, w, bpp = a.shape
    out = HostFrame(w, h, 2)
    if bpp == 4:
        out.set_alpha()       # the A byte is the pixel's coverage: it travels as the planar frame's coverage plane
    lib.orc_packed_rgb_to_yuv444(a.ctypes.data_as(C.c_void_p), w * bpp, w, h, fmt, C.byref(out.c))
    return out


def deep_to_8(planes, w: int, h: int, fmt: int) -> HostFrame:
    """planes: uint16 arrays (little-endian words) of a frame deeper than 8 bits -- planar formats (10 .. 12, 14 .. 19): (y, u, v); p010 / p016 (13, 20): (y, uv)
    with uv rows of `w` words -- -> the 8-bit frame of the layout (fmt 0 / 1 / 2) a scaler input of that format stands for"""
    arrs = [np.ascontiguousarray(a, dtype="<u2") for a in planes]
    out = HostFrame(w, h, {10: 0, 11: 1, 12: 2, 13: 0, 14: 0, 15: 1, 16: 2, 17: 0, 18: 1, 19: 2, 20: 0}[fmt])
    ptrs = (C.c_void_p * 3)(*[a.ctypes.data for a in arrs], *([None] * (3 - len(arrs))))
    strides = (C.c_int32 * 3)(*[a.strides[0] for a in arrs], *([0] * (3 - len(arrs))))
    lib.orc_deep_to_8(ptrs, strides, w, h, fmt, C.byref(out.c))
    return out


def yuyv_to_422p(pix: np.ndarray, fmt: int) -> HostFrame:
    """pix: (h, 2 * w) uint8 rows of packed 4:2:2 (fmt 21 yuyv422 / 22 uyvy422) -> the yuv422p frame with the same samples"""
    a = np.ascontiguousarray(pix, dtype=np.uint8)
    h, w2 = a.shape
    out = HostFrame(w2 // 2, h, 1)
    lib.orc_yuyv_to_422p(a.ctypes.data_as(C.c_void_p), w2, w2 // 2, h, fmt, C.byref(out.c))
    return out


def scaler_geometry(iw, ih, ow, oh):
    g = ScaleGeometry()
    lib.orc_scaler_geometry(iw, ih, ow, oh, C.byref(g))
    return g.scaled_w, g.scaled_h, g.letterbox_x, g.letterbox_y


def unify(aw, ah, bw, bh):
    w, h = C.c_uint32(), C.c_uint32()
    lib.orc_unify_picture_settings(aw, ah, bw, bh, C.byref(w), C.byref(h))
    return w.value, h.value


def to_rgba(f: HostFrame, matrix_q12=None):
    out = np.zeros((f.h, f.w * 4), np.uint8)
    m = (C.c_int32 * 12)(*matrix_q12) if matrix_q12 is not None else None
    lib.orc_yuv420_to_rgba(C.byref(f.c), out.ctypes.data_as(C.c_void_p), f.w * 4, m)
    return out.reshape(f.h, f.w, 4)


class OracleVideoMixer:
    def __init__(self, a=None, b=None, fader=1.0, sample_rate=44100):
        self.m = VMixer()
        lib.orc_video_mixer_init(C.byref(self.m), -1 if a is None else a, -1 if b is None else b, fader, sample_rate)

    def update(self, a=None, b=None, fader=1.0):
        self.m.a = -1 if a is None else a
        self.m.b = -1 if b is None else b
        self.m.fader = fader

    def run_tick(self, t, inputs, max_w=4096, max_h=2304):
        arr = (VInput * 4)()
        for i in range(4):
            e = inputs[i] if i < len(inputs) else None
            if e is not None:
                fr, dur, off = e
                arr[i].frame = C.pointer(fr.c)
                arr[i].duration_hint = lib.orc_rational_new(dur[0], dur[1])
                arr[i].tick_offset = lib.orc_rational_new(off[0], off[1])
        # the unified size is only known after the call: give the oracle room, then re-view at the real size
        # (strides depend on the width, so run twice: first to learn the size)
        present = C.c_int()
        # learn target size without mutating state: replicate the fold here (host logic under test lives in the C code;
        # this is only buffer sizing)
        sizes = []
        for i in range(4):
            e = inputs[i] if i < len(inputs) else None
            if e is not None:
                sizes.append((e[0].w, e[0].h))
            elif self.m.has_stored[i] and not self._expired(i, t):
                sizes.append((self.m.stored[i].width, self.m.stored[i].height))
        if not sizes:
            out = HostFrame(2, 2)
            lib.orc_video_mixer_run_tick(C.byref(self.m), t, arr, C.byref(out.c), C.byref(present))
            assert not present.value
            return None
        tw, th = sizes[0]
        for (w, h) in sizes[1:]:
            tw, th = unify(tw, th, w, h)
        out = HostFrame(tw, th)
        lib.orc_video_mixer_run_tick(C.byref(self.m), t, arr, C.byref(out.c), C.byref(present))
        assert present.value and out.c.width == tw and out.c.height == th
        return out

    def _expired(self, i, t):
        now = lib.orc_rational_new(t, self.m.sample_rate)
        return lib.orc_rational_cmp(now, self.m.active_until[i]) >= 0

    def __del__(self):
        lib.orc_video_mixer_free(C.byref(self.m))
