#!/usr/bin/env python
"""COUNT: on ~50 M arguments of the shapes the modules form -- Oscillator `n 2.0 pi` (oscillator.rs:77) and FmSine `co t` (fm_sine.rs:50-52),
hours into the sample clock -- how often does the device's f32 sine differ from the oracle's (float)glibc_sin(x)?  Three device modes (MX_SIN_MODE): 1 = the plain
cast of ocml's f64 sine (rounds 1-5), 0 = Ziv's strategy (the default since round 6), 2 = the double-double path on every sample.
usage: python tools/sin_count.py [ticks=500] > profiles/r06/sin_count.json"""
import json
import os
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle  # noqa: E402  (the checker)
from mixlab_amd import abi  # noqa: E402
from mixlab_amd.workspace import Workspace  # noqa: E402

n_ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 500
SR, SPT, N = 44100, 735, 64
out = {"arguments": 0, "what": f"{N} Sine oscillators (20 Hz .. 20 kHz, log-spaced) + {N} FmSines (carriers 55 Hz .. 7 kHz, deviation up to 2 kHz, driven by triangle LFOs) x {n_ticks} ticks "
                                f"@ {SR} Hz in each of 4 windows of the sample clock (0, 1 h, 12 h, 5 days: arguments up to 5e10 rad); compared with (float)glibc_sin (the CPU oracle)", "modes": {}}
ws = Workspace(SR, 60)
oscs = [ws.oscillator(20.0 * 1000.0 ** (k / (N - 1.0)), abi.WAVE_SINE) for k in range(N)]
fms = []
for k in range(N):
    lfo = ws.oscillator(0.25 + 0.37 * k, abi.WAVE_TRIANGLE)
    c = 55.0 * 128.0 ** (k / (N - 1.0))
    fm = ws.fm_sine(c - 31.25 * k, c + 31.25 * k)
    ws.connect(lfo, 0, fm, 0)
    fms.append(fm)
windows = [0, 3600 * 60, 12 * 3600 * 60, 5 * 24 * 3600 * 60]
want = {}
og = oracle.OracleGraph(ws)
for w0 in windows:
    o_, f_ = [[] for _ in oscs], [[] for _ in fms]
    for k in range(n_ticks):
        og.run_tick(w0 + k)
        for j, o in enumerate(oscs):
            o_[j].append(og.output(o, 0).copy())
        for j, f in enumerate(fms):
            f_[j].append(og.output(f, 0)[0::2].copy())
    want[w0] = (np.concatenate([np.concatenate(x) for x in o_]), np.concatenate([np.concatenate(x) for x in f_]))
for mode, name in (("1", "plain cast of the device's f64 sine (rounds 1-5)"), ("0", "Ziv (default)"), ("2", "double-double on every sample")):
    os.environ["MX_SIN_MODE"] = mode
    g = ws.build(max_ticks_per_run=n_ticks)
    rec = {"name": name, "sine_differs": 0, "fm_sine_differs": 0, "max_ulp": 0, "samples": 0}
    for w0 in windows:
        g.run_ticks(w0, n_ticks)
        got_o = np.concatenate([g.read_output(o, 0, n_ticks, False) for o in oscs])
        got_f = np.concatenate([g.read_output(f, 0, n_ticks, True)[0::2] for f in fms])
        for got, wnt, key in ((got_o, want[w0][0], "sine_differs"), (got_f, want[w0][1], "fm_sine_differs")):
            bad = got.view(np.uint32) != wnt.view(np.uint32)
            rec[key] += int(np.count_nonzero(bad))
            if bad.any():
                rec["max_ulp"] = max(rec["max_ulp"], int(np.abs(got.view(np.int32)[bad].astype(np.int64) - wnt.view(np.int32)[bad].astype(np.int64)).max()))
            rec["samples"] += int(got.size)
    rec["differs_per_million"] = round((rec["sine_differs"] + rec["fm_sine_differs"]) / rec["samples"] * 1e6, 4)
    out["modes"][mode] = rec
    out["arguments"] = rec["samples"]
    g.close()
os.environ.pop("MX_SIN_MODE", None)
print(json.dumps(out, indent=1))
