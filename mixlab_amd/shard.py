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

Video does not exchange anything: ranks run independent VideoMixer instances, or -- one picture over all ranks -- ROW BANDS:
a cross-fade cascade is not associative (each stage truncates to u8), so the picture is not split by layer but by rows, in units
of one chroma row (two luma rows): 1080p = 540 chroma rows over 8 ranks = 68, 68, 68, 68, 67, 67, 67, 67.  Cross-fade and
colour conversion are per pixel, so a band needs exactly its own rows of every same-size layer; a layer that is SCALED into the
picture needs, per plane, the source rows its band's vertical taps reach (band_source_rows: the halo).
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


def row_bands(height: int, world: int) -> list[tuple[int, int]]:
    """(first luma row, luma rows) of each rank's band of a yuv420p picture `height` rows high: whole chroma rows, sizes differing
    by at most one chroma row, the larger bands first."""
    if height % 2:
        raise ValueError("a yuv420p picture has an even number of luma rows")
    ch = height // 2
    base, rem = divmod(ch, world)
    out, row = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((2 * row, 2 * n))
        row += n
    return out


def _tap_span(o: int, src: int, dst: int) -> tuple[int, int]:
    """[first, first + n) source indices output sample o reads (before clamping), from the scaler's spec (DESIGN.md "Scaler")."""
    pos = ((2 * o + 1) * src * 65536) // (2 * dst) - 32768
    ip = pos >> 16
    if src <= dst:
        return ip - 1, ip + 3
    n = 2 * -(-2 * src // dst) + 2
    return ip - n // 2 + 1, ip + n // 2 + 1


def scale_geometry(in_w: int, in_h: int, out_w: int, out_h: int) -> tuple[int, int, int, int]:
    """DynamicScaler geometry (src/video/encode.rs:354-374): (scaled_w, scaled_h, letterbox_x, letterbox_y), all even."""
    if out_w * in_h <= out_h * in_w:
        num, den = out_w, in_w
    else:
        num, den = out_h, in_h
    sw, sh = (num * in_w // den) & ~1, (num * in_h // den) & ~1
    return sw, sh, ((out_w - sw) // 2) & ~1, ((out_h - sh) // 2) & ~1


def band_source_rows(band: tuple[int, int], in_w: int, in_h: int, out_w: int, out_h: int) -> tuple[int, int] | None:
    """The slice of a SCALED layer a band needs: (first luma row, luma rows) of the (in_w x in_h) source whose letterboxed scale into
    (out_w x out_h) the band (first luma row, rows) cuts -- the union over the three planes of the rows the band's vertical taps
    reach after clamping to the plane, widened to whole chroma rows.  None when the band lies entirely in the letterbox bars."""
    _sw, sh, _lx, ly = scale_geometry(in_w, in_h, out_w, out_h)
    lo, hi = None, None
    for c in (0, 1):                                     # luma, chroma
        b0, b1 = band[0] >> c, (band[0] + band[1]) >> c
        s0, s1 = ly >> c, (ly + sh) >> c
        a, b = max(b0, s0), min(b1, s1)
        if a >= b:
            continue
        src_h, dst_h = in_h >> c, sh >> c
        f0, _ = _tap_span(a - s0, src_h, dst_h)
        _, l1 = _tap_span(b - 1 - s0, src_h, dst_h)
        r0 = min(max(f0, 0), src_h - 1) << c
        r1 = (min(max(l1 - 1, 0), src_h - 1) + 1) << c
        lo = r0 if lo is None else min(lo, r0)
        hi = r1 if hi is None else max(hi, r1)
    if lo is None:
        return None
    lo &= ~1
    hi = (hi + 1) & ~1
    return lo, hi - lo
