"""Multi-GPU plan for the tick graph.  One process per GPU; the only exchange
step of the audio path is the mix bus.

Strips (source -> EQ -> envelope -> amplifier) are independent, so they are partitioned
contiguously over the ranks.  f32 bus summation is order-sensitive, so the sharded graph is
DEFINED as the reference-expressible hierarchy

    world x Mixer(strips / world)  ->  Mixer(world, unity gains)

Each rank computes its sub-mixer in channel order; the partial Master and Cue buses of all ranks
are exchanged with ONE all-gather per step (RCCL over xGMI; messages are 2 x T x 2 x SPT f32 per
rank -- latency-bound, far below the per-link bandwidth regime), and every rank adds the partials
in rank order 0..world-1 with the ordinary Mixer kernel (unity gain = 0 dB, fader 1.0, so each
term is `(x as f64 * 1.0) as f32 = x`).  An all-reduce would be faster but its ring order differs
per chunk, which is not the order of any graph the reference can express.

The same result with 1/4 of the traffic at 8 ranks (what bench.py uses when the step divides): the ORDERED form of
reduce-scatter + all-gather.  The run's time axis is cut into `world` slices; an all-to-all hands rank j slice j of every
rank's partial buses, rank j adds them in rank order 0..world-1 (same Mixer kernel) and an all-gather distributes the
finished slices.  Every output sample is still the rank-ordered f32 sum of the same partials -- bit for bit the
hierarchical graph above -- but a rank receives 2 (world-1)/world bus lengths instead of (world-1).

Video does not exchange anything: ranks run independent VideoMixer instances (or row bands).
"""
from __future__ import annotations


def strip_range(rank: int, world: int, total: int) -> tuple[int, int]:
    """[first, first + count) of the strips owned by `rank` (contiguous, equal shares)."""
    if total % world:
        raise ValueError(f"{total} strips do not divide over {world} ranks")
    per = total // world
    return rank * per, per


def packed_layout(world: int, floats_per_bus: int):
    """Layout of the all-gather buffers.  Each rank contributes [master | cue] (2 * floats_per_bus
    f32); the gathered buffer is rank-major.  Returns (part_len, [(master_off, cue_off)] per rank)."""
    part = 2 * floats_per_bus
    return part, [(r * part, r * part + floats_per_bus) for r in range(world)]


def combine_channels(world: int):
    """MixerParams of the final Mixer(world): unity gain, fader 1.0, no cue -- the rank-ordered f32 sum."""
    return [(0.0, 1.0, False)] * world


def slice_layout(world: int, floats_per_bus: int):
    """Slice-wise exchange: each bus is cut into `world` equal time slices of L floats.  Send / receive / final buffers are
    all [world][2][L] f32 (rank-major, then master | cue).  Returns (L, [(master_off, cue_off)] per peer inside such a buffer)."""
    if floats_per_bus % world:
        raise ValueError(f"a bus of {floats_per_bus} floats does not divide into {world} slices")
    L = floats_per_bus // world
    return L, [(r * 2 * L, r * 2 * L + L) for r in range(world)]


def pack_slices(mc, world: int):
    """[master | cue] (2 * n f32, a torch tensor) -> the all-to-all send buffer [dest][master slice | cue slice]."""
    n = mc.numel() // 2
    L = n // world
    return mc.view(2, world, L).transpose(0, 1)      # a strided view: copy_ it into the contiguous send buffer


def unpack_slices(final_all, world: int):
    """The all-gathered finished slices [rank][master slice | cue slice] -> (master, cue) as contiguous tensors."""
    L = final_all.numel() // (2 * world)
    v = final_all.view(world, 2, L).transpose(0, 1)
    return v[0].reshape(-1), v[1].reshape(-1)
