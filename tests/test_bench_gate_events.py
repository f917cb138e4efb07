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
