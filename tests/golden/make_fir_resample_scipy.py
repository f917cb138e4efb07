#!/usr/bin/env python
"""An INDEPENDENT statement of the two build-specified audio modules (Fir, Resample -- the reference has no counterpart, so the oracle and the
kernels were written from the same paragraph): scipy.signal's own FIR filter and polyphase resampler, in f64, on seeded input.

    y_fir = lfilter(taps, 1, x)                       (out[n] = sum_k taps[k] x[n - k], zero history)
    y_rs  = upfirdn(proto, x, up, down)               (zero-stuff by `up`, filter with the prototype, keep every `down`-th sample)
            with proto[k * up + phase] = table[phase][k]  --  sample m of upfirdn's output is the spec's y[m] (n = floor(m down / up), phase = (m down) mod up)

scipy sums in its own order (not the spec's ascending-tap f64 multiply-then-add), so the comparison is "every f32 within 1 ULP", not bits
(tests/test_cpu_oracle_and_abi.py, tests/test_gpu_fir_resample.py).  Run here (scipy is in the image); the .npz it writes is the fixture.
"""
import pathlib

import numpy as np
import scipy.signal as sg

HERE = pathlib.Path(__file__).resolve().parent
rng = np.random.default_rng(20260928)
frames = 735 * 8                                      # eight 60 Hz ticks at 44.1 kHz
x = (rng.standard_normal((frames, 2)) * 0.25).astype(np.float32)
x[1000:1100] = 0.0                                    # a stretch of digital silence
x[3000] = [1.0, -1.0]                                 # and a full-scale click
fir_taps = (rng.standard_normal(128) * np.exp(-np.arange(128) / 24.0) * 0.3).astype(np.float64)
y_fir = np.stack([sg.lfilter(fir_taps, [1.0], x[:, c].astype(np.float64)) for c in range(2)], axis=1).astype(np.float32)

out = {"x": x, "fir_taps": fir_taps, "y_fir": y_fir}
for name, up, down, tpp, beta in (("a", 160, 147, 16, 8.6), ("b", 2, 3, 24, 6.0), ("c", 3, 1, 8, 5.0)):
    n = up * tpp
    m = np.arange(n) - (n - 1) / 2.0
    fc = 0.5 / max(up, down)
    proto = (2 * fc * np.sinc(2 * fc * m) * np.kaiser(n, beta) * up).astype(np.float64)
    out_frames = frames * up // down
    y = np.stack([sg.upfirdn(proto, x[:, c].astype(np.float64), up, down)[:out_frames] for c in range(2)], axis=1).astype(np.float32)
    out[f"rs_{name}_ratio"] = np.array([up, down, tpp], dtype=np.int64)
    out[f"rs_{name}_table"] = np.ascontiguousarray(proto.reshape(tpp, up).T)       # [phase][k]
    out[f"rs_{name}_y"] = y
np.savez_compressed(HERE / "fir_resample_scipy.npz", **out)
print({k: v.shape for k, v in out.items()})
