#!/usr/bin/env python
"""When did every wave of the speculative EqThree launch enter and leave, and on which XCD / CU / SIMD did it run?  Reads the chunk records' padding
(mx_graph_debug_eq_records).   python tools/wave_times.py --strips 128 --ticks 2048 --chunks 512"""
import argparse
import ctypes as C
import os
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--strips", type=int, default=128)
    ap.add_argument("--ticks", type=int, default=2048)
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--overlap-tail", action="store_true")
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    if args.chunks:
        os.environ["MX_EQ_SPEC_CHUNKS"] = str(args.chunks)
    import synth
    from bench import build_strips, gate_events
    from mixlab_amd import abi, video
    from mixlab_amd.workspace import Workspace
    T, SR, spt = args.ticks, 48000, 800
    ws, mix, srcs, trigs = build_strips(abi, Workspace, synth, args.strips, 0, SR, want_trigs=True)
    g = ws.build(max_ticks_per_run=T, flags=abi.FLAG_OVERLAP_TAIL if args.overlap_tail else 0)
    base = min(T, 256)
    for j, s in enumerate(srcs):
        g.write_source(s, np.tile(synth.noise(j, base * spt), (T + base - 1) // base)[: T * spt], T)
    for i in range(args.steps):
        ev = gate_events(abi, trigs, 0, i * T, T)
        if ev: g.schedule_params_batch(ev[0], ev[1])
        g.run_ticks(i * T, T)
    ran, _ = g.eq_spec_stats()
    g.sync()
    ran, _ = g.eq_spec_stats()
    n_chunks = ran // args.steps // args.strips
    p, nbytes = g.debug_eq_records()
    rec = np.empty(args.strips * n_chunks * 144, np.uint8)
    abi.check(abi.lib.mx_device_download(rec.ctypes.data_as(C.c_void_p), C.c_void_p(p), rec.size, None))
    pad = rec.reshape(-1, 144)[:, 136:144].copy().view(np.uint32).reshape(args.strips, n_chunks, 2)
    wpi = (n_chunks + 63) // 64
    rows = []
    for i in range(args.strips):
        for w in range(wpi):
            lanes = pad[i, 64 * w: 64 * w + 64]
            if len(lanes) < 3:
                continue
            t0, hw, xcc = int(lanes[0, 0]), int(lanes[1, 0]), int(lanes[2, 0])
            t1 = int(lanes[:, 1].max())
            rows.append((i * wpi + w, t0, (t1 - t0) & 0xffffffff, hw, xcc))
    a = np.array(rows, dtype=np.int64)
    tmin = a[:, 1].min()
    start = (a[:, 1] - tmin) & 0xffffffff
    dur = a[:, 2]
    print(f"strips {args.strips} ticks {T} chunks per strip {n_chunks} -> {len(a)} waves ({wpi} per strip)")
    print(f"enter (clock ticks after the first wave): min {start.min()} median {int(np.median(start))} p90 {int(np.percentile(start, 90))} max {start.max()}")
    print(f"life  (clock ticks): min {dur.min()} median {int(np.median(dur))} p90 {int(np.percentile(dur, 90))} max {dur.max()}")
    end = start + dur
    print(f"last wave leaves at {end.max()}  (first leaves at {end.min()})")
    hw = a[:, 3]
    simd, cu, sh, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, a[:, 4] & 15
    slot = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
    uniq, cnt = np.unique(slot, return_counts=True)
    print(f"distinct (xcc, se, sh, cu, simd) slots used: {len(uniq)}; waves per used slot: " + ", ".join(f"{k}: {int((cnt == k).sum())}" for k in sorted(set(cnt))))
    for k in sorted(set(cnt)):
        sel = np.isin(slot, uniq[cnt == k])
        print(f"  waves on slots holding {k}: life median {int(np.median(dur[sel]))}, enter median {int(np.median(start[sel]))}")
    print("per XCD: " + ", ".join(f"{x}: {int((xcc == x).sum())} waves" for x in sorted(set(xcc))))


main()
