"""The one exchange step of the sharded audio job: N ranks each hold the partial Master / Cue buses of
their strip shard; every rank ends with the whole bus.  RCCL over xGMI through torch.distributed (backend "nccl").

The sharded graph is DEFINED as the reference-expressible hierarchy  N x Mixer(strips / N) -> Mixer(N, unity)  (mixlab_amd/shard.py):
every output sample is the f32 sum of the N partials in rank order 0 .. N-1 (src/module/mixer.rs:57-68 applied to the partial
buses), computed here with the ordinary Mixer kernel on each rank.  Three exchanges:

  "allgather"  one all-gather of the whole [master | cue] partials, then the rank-ordered sum: (N - 1) bus lengths received.
  "slices"     the ORDERED form of reduce-scatter + all-gather: the step's time axis is cut into N slices, an all-to-all hands
               rank j slice j of every partial, rank j adds them in rank order, an all-gather distributes the finished slices:
               2 (N - 1) / N bus lengths received (1/4 of the above at N = 8).  Bit-identical to "allgather".
  "allreduce"  ncclAllReduce(sum): what the north-star names.  NOT the sum order of any graph the reference can express
               (ring order differs per chunk): offered as an explicitly non-parity mode; BusExchange.max_ulp_vs() measures its
               deviation from the ordered result.

The exchange is pipelined against the next step's compute: the partial buses are packed device-to-device into one of two
slots on the compute stream; a second stream waits for that, runs the collectives and the combine and records `done`; the
compute stream only waits for `done` of the slot it is about to reuse.  Steady-state step = max(compute, exchange).

PyTorch is plumbing here (streams, events, torch.distributed); the summing is the library's Mixer kernel.
"""
from __future__ import annotations

from . import shard
from .workspace import Workspace

MODES = ("auto", "slices", "allgather", "allreduce")


class _DevArray:
    """zero-copy torch view of a device buffer owned by libmixlab_gpu"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def dev_view(torch, ptr: int, n: int):
    return torch.as_tensor(_DevArray(ptr, n), device="cuda")


class BusExchange:
    """Combine the partial (Master, Cue) buses of `graph`'s Mixer `mix` across the ranks of the default process group.

    submit(i)        after graph.run_ticks(...) of step i on the compute stream: pack + exchange + combine, asynchronously
    wait(i)          make the CURRENT torch stream wait for step i's combined bus
    result(i)        (master, cue) torch views of step i's combined bus (valid after wait(i) / a synchronise)
    """

    def __init__(self, torch, dist, graph, mix: int, n_ticks: int, sample_rate: int, device: int, compute_stream, mode: str = "auto"):
        if mode not in MODES:
            raise ValueError(f"exchange mode must be one of {MODES}")
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.T = n_ticks
        self.stream = compute_stream
        self.comm = torch.cuda.Stream()
        world, T = self.world, n_ticks
        m_ptr, fpt = graph.output_device_ptr(mix, 0)
        c_ptr, _ = graph.output_device_ptr(mix, 1)
        self.n_fl = n_fl = fpt * T
        self.m_view, self.c_view = dev_view(torch, m_ptr, n_fl), dev_view(torch, c_ptr, n_fl)
        # Master and Cue are neighbours in the graph's slab: one device-to-device copy packs both
        self.mc_view = dev_view(torch, m_ptr, 2 * n_fl) if c_ptr == m_ptr + 4 * n_fl else None
        if mode == "auto":
            mode = "slices" if (world >= 4 and T % world == 0) else "allgather"
        if mode == "slices" and T % world:
            raise ValueError(f"{T} ticks per step do not divide into {world} time slices")
        self.mode = mode
        self.slots = []
        for _ in range(2):
            sl = {"packed": torch.cuda.Event(), "done": torch.cuda.Event(), "used": False}
            if mode == "allreduce":
                sl["buf"] = torch.empty(2 * n_fl, dtype=torch.float32, device="cuda")
                self.slots.append(sl)
                continue
            cws = Workspace(sample_rate, 60)
            fm = cws.mixer(shard.combine_channels(world))   # unity gains: the f32 sum of the partials in rank order
            fc = cws.mixer(shard.combine_channels(world))
            src_m = [cws.source_stereo() for _ in range(world)]
            src_c = [cws.source_stereo() for _ in range(world)]
            for r in range(world):
                cws.connect(src_m[r], 0, fm, r)
                cws.connect(src_c[r], 0, fc, r)
            if mode == "slices":
                L, offs = shard.slice_layout(world, n_fl)
                t_slice = T // world
                sl.update(send=torch.empty(world * 2 * L, dtype=torch.float32, device="cuda"),
                          recv=torch.empty(world * 2 * L, dtype=torch.float32, device="cuda"),
                          fin=torch.empty(2 * L, dtype=torch.float32, device="cuda"),
                          final_all=torch.empty(world * 2 * L, dtype=torch.float32, device="cuda"), L=L, t_slice=t_slice)
                cg = cws.build(max_ticks_per_run=t_slice, device=device, stream=self.comm.cuda_stream)
                for r in range(world):
                    cg.bind_source_device(src_m[r], sl["recv"].data_ptr() + offs[r][0] * 4)
                    cg.bind_source_device(src_c[r], sl["recv"].data_ptr() + offs[r][1] * 4)
                fm_ptr, _ = cg.output_device_ptr(fm, 0)
                fc_ptr, _ = cg.output_device_ptr(fc, 0)
                sl.update(cg=cg, fm_view=dev_view(torch, fm_ptr, L), fc_view=dev_view(torch, fc_ptr, L))
            else:
                part_len, offs = shard.packed_layout(world, n_fl)
                sl.update(part=torch.empty(part_len, dtype=torch.float32, device="cuda"),
                          gathered=torch.empty(world * part_len, dtype=torch.float32, device="cuda"))
                cg = cws.build(max_ticks_per_run=T, device=device, stream=self.comm.cuda_stream)
                for r in range(world):
                    cg.bind_source_device(src_m[r], sl["gathered"].data_ptr() + offs[r][0] * 4)
                    cg.bind_source_device(src_c[r], sl["gathered"].data_ptr() + offs[r][1] * 4)
                fm_ptr, _ = cg.output_device_ptr(fm, 0)
                fc_ptr, _ = cg.output_device_ptr(fc, 0)
                sl.update(cg=cg, fm_view=dev_view(torch, fm_ptr, n_fl), fc_view=dev_view(torch, fc_ptr, n_fl))
            self.slots.append(sl)

    # bytes a rank receives per step (what the xGMI links carry towards it)
    def bytes_received_per_step(self) -> int:
        bus = 2 * self.n_fl * 4
        w = self.world
        if self.mode == "allgather":
            return (w - 1) * bus
        return 2 * (w - 1) * bus // w            # slices, and a ring all-reduce

    def submit(self, i: int) -> None:
        torch, dist, world = self.torch, self.dist, self.world
        sl = self.slots[i % 2]
        stream = self.stream
        if sl["used"]:
            stream.wait_event(sl["done"])          # the exchange that last used this slot has finished
        with torch.cuda.stream(stream):
            if self.mode == "slices":              # device-to-device pack: [dest][master slice | cue slice]
                sv = sl["send"].view(world, 2, sl["L"])
                sv[:, 0, :].copy_(self.m_view.view(world, sl["L"]))
                sv[:, 1, :].copy_(self.c_view.view(world, sl["L"]))
            else:
                dst = sl["buf"] if self.mode == "allreduce" else sl["part"]
                if self.mc_view is not None:
                    dst.copy_(self.mc_view)        # (master, cue) in one copy
                else:
                    dst[: self.n_fl].copy_(self.m_view)
                    dst[self.n_fl:].copy_(self.c_view)
            sl["packed"].record(stream)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(sl["packed"])
            if self.mode == "slices":
                dist.all_to_all_single(sl["recv"], sl["send"])            # slice j of every rank's partial buses -> rank j
                sl["cg"].run_ticks(0, sl["t_slice"])                      # rank-ordered f32 sum of my slice: Mixer(N, unity)
                sl["fin"][: sl["L"]].copy_(sl["fm_view"]); sl["fin"][sl["L"]:].copy_(sl["fc_view"])
                dist.all_gather_into_tensor(sl["final_all"], sl["fin"])   # every rank ends with the whole Master and Cue
            elif self.mode == "allgather":
                dist.all_gather_into_tensor(sl["gathered"], sl["part"])   # ONE all-gather per step
                sl["cg"].run_ticks(0, self.T)                              # rank-ordered f32 sum: Mixer(N, unity)
            else:
                dist.all_reduce(sl["buf"], op=dist.ReduceOp.SUM)           # non-parity: the ring decides the order
            sl["done"].record(self.comm)
        sl["used"] = True

    def wait(self, i: int) -> None:
        self.torch.cuda.current_stream().wait_event(self.slots[i % 2]["done"])

    def result(self, i: int):
        sl = self.slots[i % 2]
        if self.mode == "slices":
            return shard.unpack_slices(sl["final_all"], self.world)
        if self.mode == "allgather":
            return sl["fm_view"], sl["fc_view"]
        return sl["buf"][: self.n_fl], sl["buf"][self.n_fl:]

    def max_ulp_vs(self, i: int, other_master, other_cue) -> int:
        """Largest distance in f32 ULPs between step i's combined bus and another (e.g. the ordered) result."""
        torch = self.torch

        def key(x):
            v = x.contiguous().view(torch.int32).to(torch.int64)
            return torch.where(v < 0, -0x80000000 - v, v)
        m, c = self.result(i)
        return int(max((key(m) - key(other_master)).abs().max().item(), (key(c) - key(other_cue)).abs().max().item()))

    def close(self) -> None:
        for sl in self.slots:
            cg = sl.get("cg")
            if cg is not None:
                cg.close()
        self.slots = []
