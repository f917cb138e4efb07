"""What a strip costs whose Amplifier is modulated by a BUFFER (an oscillator as LFO) instead of an inline Envelope, and what a FEW strips of another epilogue mode cost the rest: 1024 strips x T ticks.  usage: python tools/ctl_probe.py [T] [flags]"""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import synth
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = 1024
import os
for kind in (os.environ.get("PROBE_KINDS", "env,lfo,none,mixed,mixed2,envgate").split(",")):
    ws = Workspace(48000, 60)
    mix = ws.mixer([(0.0, 0.5, False)] * n)
    srcs = []
    lfo = ws.oscillator(2.0, abi.WAVE_TRIANGLE) if kind == "lfo" else (ws.oscillator(1.5, abi.WAVE_SQUARE) if kind == "envgate" else None)
    for k in range(n):
        s = ws.source_mono(); e = ws.eq_three(1.0, -2.0, 3.0); p = ws.stereo_panner(); a = ws.amplifier(1.0, 0.5)
        ws.connect(s, 0, e, 0); ws.connect(e, 0, p, 0); ws.connect(e, 0, p, 1); ws.connect(p, 0, a, 0)
        if kind == "env" or (kind == "mixed" and k % 64 != 0):      # mixed: one strip in 64 has no Envelope on its Amplifier -- another epilogue mode in the same launch group
            tr = ws.trigger(True); en = ws.envelope(); ws.connect(tr, 0, en, 0); ws.connect(en, 0, a, 1)
        elif kind == "envgate":                    # an Envelope per strip whose gate is a module's OUTPUT (a square LFO), not a Trigger: a state machine over the samples
            en = ws.envelope(); ws.connect(lfo, 0, en, 0); ws.connect(en, 0, a, 1)
        elif kind == "lfo":
            ws.connect(lfo, 0, a, 1)
        if kind == "mixed2" and k % 64 == 0:      # one strip in 64 goes from its StereoPanner straight to the Mixer: epilogue "panner" among 63 "amplifier, constant depth"
            ws._conn.pop((a, 0), None)                # the Amplifier stays in the workspace, unconnected and unreachable from the Mixer
            ws.connect(p, 0, mix, k); srcs.append(s)
            continue
        ws.connect(a, 0, mix, k); srcs.append(s)
    g = ws.build(max_ticks_per_run=T, flags=flags)
    blk = synth.noise(1, 256 * ws.spt)
    x = np.tile(blk, (T + 255) // 256)[: T * ws.spt]
    for s in srcs:
        g.write_source(s, x, T)
    g.run_ticks(0, T); g.sync()
    g.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(1, 4):
        g.run_ticks(i * T, T)
    g.sync()
    dt = (time.perf_counter() - t0) / 3
    g.profile_enable(False)
    by_kind, _tot, n_prof = g.profile_collect()
    prof = {k: round(v / max(1, n_prof), 3) for k, v in sorted(by_kind.items()) if v > 0}
    print(kind, f"{dt * 1e3:.2f} ms per step, {n * T / dt / 1e6:.1f} M channel-ticks/s", g.eq_spec_stats(), prof, flush=True)
    g.close()
