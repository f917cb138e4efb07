#!/usr/bin/env python
"""How long the speculative EqThree's repair pass takes on ONE strip of a given material (T = 2048 ticks @ 48 kHz): run under
rocprofv3 --kernel-trace; prints eq_spec stats per material."""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import synth
from mixlab_amd.workspace import Workspace

SR, SPT, T = 48000, 800, 2048
L = T * SPT
def material(kind):
    x = synth.noise(5, 256 * SPT); x = np.tile(x, 8)[:L].copy()
    if kind == "muted": x[:] = 0.0
    elif kind == "gaps":
        pos = 30000
        while pos < L:
            x[pos:pos + 96000] = 0.0; pos += 96000 + 144000
    elif kind == "onegap": x[400000:496000] = 0.0
    return x
for kind in sys.argv[1:] or ["noise", "muted", "gaps", "onegap"]:
    ws = Workspace(SR, 60)
    s = ws.source_mono(); e = ws.eq_three(3.0, -2.0, 1.5); ws.connect(s, 0, e, 0)
    g = ws.build(max_ticks_per_run=T)
    g.write_source(s, synth.noise(5, L) if kind == "muted" else material(kind), T)
    g.run_ticks(0, T)                      # a first run on programme, so that a muted strip has a state to decay from
    g.write_source(s, material(kind), T)
    r0 = g.eq_repair_stats()
    for i in range(3):
        g.run_ticks((i + 1) * T, T)
    g.sync()
    r1 = g.eq_repair_stats()
    print(kind, "over 3 runs:", {k: r1[k] - r0[k] for k in r1}, flush=True)
