"""Feasibility: config 3's FIR launch beside the resampler + Mixer of another run, as two independent graphs on two streams (what a generalised second-stream mode
could reach), against the same two graphs one after the other."""
import sys, time, pathlib, ctypes as C
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import synth
from mixlab_amd.workspace import Workspace

hip = C.CDLL("libamdhip64.so")
def mkstream():
    h = C.c_void_p(); assert hip.hipStreamCreate(C.byref(h)) == 0; return h
n_ch, SPT, T = 256, 735, 128
up, down, tpp = 160, 147, 16
n = up * tpp
m = np.arange(n) - (n - 1) / 2.0
fc = 0.5 / max(up, down) * 0.92
table = np.ascontiguousarray((2 * fc * np.sinc(2 * fc * m) * np.kaiser(n, 8.6) * up).reshape(tpp, up).T)
s1, s2 = mkstream(), mkstream()
# graph F: sources -> FIR (a Mixer behind it so that the outputs are consumed; its cost is small)
wf = Workspace(44100, 60); srcf, firs = [], []
for k in range(n_ch):
    taps = (synth.uniform(20 + k, 128, -1.0, 1.0) * np.exp(-np.arange(128) / 24.0) * 0.35).astype(np.float64)
    s = wf.source_stereo(); f = wf.fir(taps); wf.connect(s, 0, f, 0); srcf.append(s); firs.append(f)
gf = wf.build(max_ticks_per_run=T, stream=s1.value)
# graph R: sources -> resampler -> Mixer
wr = Workspace(44100, 60); srcr, rs = [], []
for k in range(n_ch):
    s = wr.source_stereo(); r = wr.resample(up, down, table); wr.connect(s, 0, r, 0); srcr.append(s); rs.append(r)
mix = wr.mixer([(0.0, 1.0, k % 2 == 0) for k in range(n_ch)])
for k, r in enumerate(rs):
    wr.connect(r, 0, mix, k)
gr = wr.build(max_ticks_per_run=T, stream=s2.value)
for k in range(n_ch):
    blk = np.tile(synth.noise(60 + k, 2 * SPT * 64), 2)[: 2 * SPT * T]
    gf.write_source(srcf[k], blk, T); gr.write_source(srcr[k], blk, T)
def sync():
    gf.sync(); gr.sync()
for i in range(3):
    gf.run_ticks(i * T, T); gr.run_ticks(i * T, T)
sync()
K = 20
t0 = time.perf_counter()
for i in range(K):
    gf.run_ticks((3 + i) * T, T); gf.sync(); gr.run_ticks((3 + i) * T, T); gr.sync()
seq = (time.perf_counter() - t0) / K * 1e3
t0 = time.perf_counter()
for i in range(K):
    gf.run_ticks((30 + i) * T, T)
gf.sync()
f_only = (time.perf_counter() - t0) / K * 1e3
t0 = time.perf_counter()
for i in range(K):
    gr.run_ticks((30 + i) * T, T)
gr.sync()
r_only = (time.perf_counter() - t0) / K * 1e3
t0 = time.perf_counter()
for i in range(K):
    gf.run_ticks((60 + i) * T, T); gr.run_ticks((60 + i) * T, T)
sync()
both = (time.perf_counter() - t0) / K * 1e3
print(f"FIR graph alone {f_only:.3f} ms, resampler + Mixer graph alone {r_only:.3f} ms, one after the other with syncs {seq:.3f} ms, both queues running {both:.3f} ms per step")
