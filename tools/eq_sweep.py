#!/usr/bin/env python
"""Tuning aid (not part of the product path): device time of the config-2 graph's launch groups for a sweep of
speculative-EqThree plans (MX_EQ_SPEC_CHUNKS), with held or toggling gates.

  python tools/eq_sweep.py [--ticks 2048] [--strips 1024] [--chunks 0,64,128,192,256,384] [--toggle] [--fast]
"""
import argparse
import ctypes as C
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=2048)
    ap.add_argument("--strips", type=int, default=1024)
    ap.add_argument("--chunks", default="0")
    ap.add_argument("--toggle", action="store_true")
    ap.add_argument("--fast", action="store_true")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--fp-contract", action="store_true", help="MX_FLAG_FP_CONTRACT")
    ap.add_argument("--no-profile", action="store_true", help="no per-group events: the wall clock of the steps only")
    ap.add_argument("--host-times", action="store_true", help="print how long every run_ticks call kept the host (us)")
    ap.add_argument("--overlap-tail", action="store_true", help="MX_FLAG_OVERLAP_TAIL: the Mixer bank of step k beside step k + 1's EqThree group")
    args = ap.parse_args()
    import synth
    from bench import build_strips, gate_events
    from mixlab_amd import abi
    from mixlab_amd.workspace import Workspace

    T, SR, spt = args.ticks, 48000, 800
    for ch in [int(c) for c in args.chunks.split(",")]:
        if ch:
            os.environ["MX_EQ_SPEC_CHUNKS"] = str(ch)
        else:
            os.environ.pop("MX_EQ_SPEC_CHUNKS", None)
        ws, mix, srcs, trigs = build_strips(abi, Workspace, synth, args.strips, 0, SR, want_trigs=True)
        g = ws.build(max_ticks_per_run=T, flags=(abi.FLAG_EQ_FAST if args.fast else 0) | (abi.FLAG_FP_CONTRACT if args.fp_contract else 0) | (abi.FLAG_OVERLAP_TAIL if args.overlap_tail else 0))
        base = min(T, 256)
        for j, s in enumerate(srcs):
            blk = synth.noise(j, base * spt)
            g.write_source(s, np.tile(blk, (T + base - 1) // base)[: T * spt], T)
        ev = [gate_events(abi, trigs, 0, i * T, T) if args.toggle else None for i in range(args.steps + 2)]
        for i in range(2):
            if ev[i]: g.schedule_params_batch(ev[i][0], ev[i][1])
            g.run_ticks(i * T, T)
        g.sync()
        g.profile_enable(not args.no_profile)
        t0 = time.perf_counter()
        host = []
        for i in range(args.steps):
            th = time.perf_counter()
            if ev[2 + i]: g.schedule_params_batch(ev[2 + i][0], ev[2 + i][1])
            g.run_ticks((2 + i) * T, T)
            host.append((time.perf_counter() - th) * 1e6)
        g.sync()
        if args.host_times: print("host us per call:", " ".join(f"{h:.0f}" for h in host))
        dt = (time.perf_counter() - t0) / args.steps
        by_kind, tot, n = g.profile_collect() if not args.no_profile else ({}, 0, 1)
        ran, rep = g.eq_spec_stats()
        print(f"strips={args.strips} T={T} overlap={args.overlap_tail} chunks={ch or 'auto'} toggle={args.toggle} fast={args.fast} fc={args.fp_contract} sb={os.environ.get('MX_EQ_SPEC_SB', 'auto')}: step {dt * 1e3:.3f} ms  " +
              "  ".join(f"{k} {v / n:.3f}" for k, v in sorted(by_kind.items())) + f"  | spec chunks {ran} repaired {rep}"
              f"  => {args.strips * T / dt / 1e6:.1f} M channel-ticks/s", flush=True)
        g.close()


if __name__ == "__main__":
    main()
