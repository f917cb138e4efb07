"""bench.py's gate schedule IS config 2 as written: Trigger k toggles every 30 ticks with phase k mod 60 -- every
toggle of every strip on ticks [t0, t0 + n) appears once, at its tick, with the right state; a toggle that falls exactly on a
submission boundary is scheduled at tick_in_run 0."""
import ctypes as C
import sys
import pathlib

import pytest

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))


@pytest.mark.parametrize("t0,n", [(0, 64), (60, 64), (2048, 2048), (4096, 30), (90, 1), (29, 2)])
def test_gate_events_equal_the_definition(t0, n):
    import bench
    from mixlab_amd import abi
    trigs, first = list(range(100, 100 + 64)), 3
    r = bench.gate_events(abi, trigs, first, t0, n)
    got = set()
    if r is not None:
        for e in r[2][0]:
            p = C.cast(int(e["params"]), C.POINTER(abi.TriggerParams)).contents
            got.add((int(e["node"]), int(e["tick_in_run"]), int(p.gate_open)))
        assert r[1] == len(got)
    want = set()
    for j, tr in enumerate(trigs):
        k = first + j
        gate = lambda tick: ((tick + k) // 30) % 2 == 1
        for c in range(n):
            if gate(t0 + c) != gate(t0 + c - 1):
                want.add((tr, c, 1 if gate(t0 + c) else 0))
    assert got == want


def test_headline_replay_follows_the_tick_by_tick_oracle_with_toggles_between_ticks():
    """tests/headline_replay.py (the checker bench.py runs after its timed region) drives the oracle in stretches between two gate toggles with a
    source that replays a resident buffer; that has to equal the plain tick-by-tick oracle with ModuleT::update before every tick."""
    import numpy as np
    import bench
    import headline_replay as hr
    import oracle
    import synth
    from mixlab_amd import abi
    from mixlab_amd.workspace import Workspace
    T, steps, SR, spt = 64, 3, 48000, 800
    for k in (0, 7, 29, 30, 59, 1023):
        ws1, mix1, srcs1, trigs1 = bench.build_strips(abi, Workspace, synth, 1, k, SR, total=1024, want_trigs=True)
        src = np.tile(synth.noise(k, 16 * spt), T // 16)
        out = [None]
        hr._replay_one(oracle, abi, ws1, (mix1, srcs1[0], trigs1[0], mix1 + 6), k, src, T, T * steps, T, True, out, 0)
        ws2, mix2, srcs2, trigs2 = bench.build_strips(abi, Workspace, synth, 1, k, SR, total=1024, want_trigs=True)
        og = oracle.OracleGraph(ws2)
        res = []
        for t in range(T * steps):
            og.update_params(trigs2[0], abi.TriggerParams(1 if hr.gate_open(t, k) else 0))
            og.set_source(srcs2[0], src[(t % T) * spt:(t % T + 1) * spt])
            og.run_tick(t)
            if t >= T * (steps - 1):
                res.append(og.output(mix2 + 6, 0))
        assert np.array_equal(np.concatenate(res).view(np.uint32), out[0].view(np.uint32)), f"strip {k}"
    assert hr.sample_strips(1024, 16)[0] == 0 and hr.sample_strips(1024, 16)[-1] == 1023 and len(set(hr.sample_strips(1024, 16))) == 16
