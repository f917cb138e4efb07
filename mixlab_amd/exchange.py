"""ctypes wrapper of the library's bus exchange, mx_exchange_* (include/mixlab_gpu.h "multi-GPU"):
N ranks each hold the partial Master / Cue buses of their strip shard; every rank ends with the whole bus, the f32 sum of
the partials in rank order 0 .. N-1 -- the reference-expressible graph  N x Mixer(strips / N) -> Mixer(N, unity)
(mixlab_amd/shard.py).  The collectives (RCCL over xGMI: ncclAllGather / grouped ncclSend + ncclRecv / ncclAllReduce), the
rank-ordered combine and the pipelining against the next step's compute all live in libmixlab_gpu.so
(mixlab_amd/csrc/mx_exchange.cpp); this file only declares the entry points.  No torch here.

Two transports: an RCCL communicator built from the job's ncclUniqueId (`unique_id()` on rank 0, distributed by the host),
or a `LoopbackGroup` of W exchanges inside one process (W virtual ranks on one GPU: device-to-device copies in place of the
collectives) -- how the single-GPU tests run configs[4] with world > 1.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .abi import _proto, check, lib

MODES = ("auto", "allgather", "slices", "allreduce")          # index = MX_EXCHANGE_*
ID_BYTES = 128


class ExchangeInfo(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("rank", C.c_uint32), ("world", C.c_uint32), ("loopback", C.c_uint32),
                ("floats_per_bus", C.c_uint64), ("bytes_received_per_step", C.c_uint64)]


_proto("mx_exchange_unique_id", C.c_int, C.c_void_p)
_proto("mx_loopback_group_create", C.c_int, C.c_uint32, C.POINTER(C.c_void_p))
_proto("mx_loopback_group_destroy", None, C.c_void_p)
_proto("mx_exchange_create", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p))
_proto("mx_exchange_destroy", None, C.c_void_p)
_proto("mx_exchange_submit", C.c_int, C.c_void_p, C.c_uint64)
_proto("mx_exchange_wait", C.c_int, C.c_void_p, C.c_uint64, C.c_void_p)
_proto("mx_exchange_result", C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))
_proto("mx_exchange_release", C.c_int, C.c_void_p, C.c_uint64, C.c_void_p)
_proto("mx_exchange_read_result", C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p)
_proto("mx_exchange_elapsed_ms", C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_float))
_proto("mx_exchange_sync", C.c_int, C.c_void_p)
_proto("mx_exchange_get_info", C.c_int, C.c_void_p, C.POINTER(ExchangeInfo))


class _rccl_banner_to_stderr:
    """RCCL prints a version banner ("RCCL version : ...", five lines) on the PROCESS's stdout when its first communicator comes up.  A program whose stdout is its
    result -- bench.py prints one JSON line -- must not carry it: while RCCL initialises, file descriptor 1 points at stderr (C stdio flushed on both sides)."""

    def __enter__(self):
        import os, sys
        self._os = os
        sys.stdout.flush()
        self._libc = C.CDLL(None)
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        self._libc.fflush(None)
        self._os.dup2(self._saved, 1)
        self._os.close(self._saved)
        return False


def unique_id() -> bytes:
    """ncclGetUniqueId: made on rank 0, handed to every rank's BusExchange."""
    buf = C.create_string_buffer(ID_BYTES)
    with _rccl_banner_to_stderr():
        check(lib.mx_exchange_unique_id(buf))
    return buf.raw


class LoopbackGroup:
    """mx_loopback_group: the in-process transport for `world` exchanges of one process."""

    def __init__(self, world: int):
        self._h = C.c_void_p()
        check(lib.mx_loopback_group_create(world, C.byref(self._h)))
        self.world = world

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.mx_loopback_group_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


class BusExchange:
    """mx_exchange: combine the partial (Master, Cue) buses of `graph`'s Mixer `mix` across the ranks of the job.

    submit(i)        after graph.run_ticks(...) of step i: pack + exchange + combine, asynchronously
    wait(i, stream)  make a stream (default: the graph's) wait for step i's combined bus
    result(i)        (master, cue) of step i on the host (synchronous)
    """

    def __init__(self, graph, mix: int, n_ticks: int, rank: int, world: int, mode: str = "auto", nccl_id: bytes | None = None,
                 loopback: LoopbackGroup | None = None):
        if mode not in MODES:
            raise ValueError(f"exchange mode must be one of {MODES}")
        self._h = C.c_void_p()
        self._graph, self._grp = graph, loopback            # the library requires both to outlive the exchange
        idbuf = C.create_string_buffer(nccl_id, ID_BYTES) if nccl_id is not None else None
        with _rccl_banner_to_stderr():
            check(lib.mx_exchange_create(graph._h, mix, n_ticks, rank, world, idbuf, loopback._h if loopback is not None else None,
                                         MODES.index(mode), C.byref(self._h)))
        info = ExchangeInfo()
        check(lib.mx_exchange_get_info(self._h, C.byref(info)))
        self.mode, self.rank, self.world = MODES[info.mode], info.rank, info.world
        self.n_fl = info.floats_per_bus
        self._bytes = info.bytes_received_per_step

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.mx_exchange_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def bytes_received_per_step(self) -> int:
        """bytes a rank receives per step (what the xGMI links carry towards it)"""
        return self._bytes

    def submit(self, i: int) -> None:
        check(lib.mx_exchange_submit(self._h, i))

    def wait(self, i: int, stream: int | None = None) -> None:
        check(lib.mx_exchange_wait(self._h, i, stream))

    def release(self, i: int, stream: int | None = None) -> None:
        check(lib.mx_exchange_release(self._h, i, stream))

    def sync(self) -> None:
        check(lib.mx_exchange_sync(self._h))

    def device_result(self, i: int):
        """-> (master device pointer, cue device pointer, floats per bus)"""
        m, c, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        check(lib.mx_exchange_result(self._h, i, C.byref(m), C.byref(c), C.byref(n)))
        return m.value, c.value, n.value

    def result(self, i: int):
        m, c = np.empty(self.n_fl, dtype=np.float32), np.empty(self.n_fl, dtype=np.float32)
        check(lib.mx_exchange_read_result(self._h, i, m.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p)))
        return m, c

    def elapsed_ms(self, i: int) -> float:
        ms = C.c_float()
        check(lib.mx_exchange_elapsed_ms(self._h, i, C.byref(ms)))
        return ms.value

    def max_ulp_vs(self, i: int, other_master: np.ndarray, other_cue: np.ndarray) -> int:
        """Largest distance in f32 ULPs between step i's combined bus and another (e.g. the ordered) result."""
        def key(x):
            v = np.ascontiguousarray(x, dtype=np.float32).view(np.int32).astype(np.int64)
            return np.where(v < 0, -0x80000000 - v, v)
        m, c = self.result(i)
        return int(max(np.abs(key(m) - key(other_master)).max(), np.abs(key(c) - key(other_cue)).max()))
