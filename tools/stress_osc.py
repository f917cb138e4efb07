"""Oscillator / FmSine stress through the module path: random and extreme frequencies (0, negative, sub-Hz, above Nyquist, 1e6), ticks up to
2^40 samples into the session, every waveform.  Saw / Triangle / Square / On / Off bit-exact (Square is the exact sign of sin x); Sine and FmSine
BIT-EXACT too since round 6 (mx_sin_f32.hpp) wherever the sine's argument stays below 2^40 rad -- beyond that the device's own sine is cast and 1 ULP is allowed.
MX_SIN_MODE=2 in the environment runs the double-double path on every sample.  Usage: python tools/stress_osc.py [first] [count]"""
import sys, pathlib, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle, synth
from mixlab_amd import abi
from test_gpu_audio_parity import assert_bit_exact, assert_ulp

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
WAVES = [abi.WAVE_SAW, abi.WAVE_TRIANGLE, abi.WAVE_ON, abi.WAVE_OFF, abi.WAVE_SINE, abi.WAVE_SQUARE]

bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    SR, SPT = [(44100, 735), (48000, 800)][int(rng.integers(0, 2))]
    freq = float(rng.choice([0.0, -440.0, 1e-3, 0.5, 22050.0, 24000.0, 1e6, 1e-12])) if rng.random() < 0.3 else float(rng.uniform(10, 20000))
    tick = int(rng.choice([0, 1, 60, 216000, 2**31 // 735, 2**40 // 800])) if rng.random() < 0.5 else int(rng.integers(0, 10**7))
    t = tick * SPT
    wave = WAVES[int(rng.integers(0, len(WAVES)))]
    what = f"seed {seed}: wave {wave} freq {freq} tick {tick} {SR} Hz"
    try:
        want_m, want_s = oracle.oscillator_run(freq, wave, SR, t, SPT)
        m = abi.Module(abi.KIND_OSCILLATOR, abi.OscillatorParams(freq, wave, 0), sample_rate=SR)
        got_m, got_s = np.empty(SPT, np.float32), np.empty(2 * SPT, np.float32)
        m.run_tick(t, [], [(abi.MX_MONO, got_m), (abi.MX_STEREO, got_s)])
        assert_bit_exact(got_s[0::2], got_m, what + " stereo L"); assert_bit_exact(got_s[1::2], got_m, what + " stereo R")
        arg_max = abs((t + SPT) / SR * freq * 2.0 * np.pi)
        if wave == abi.WAVE_SINE and arg_max >= 2.0 ** 40:
            assert_ulp(got_m, want_m, 1, what)
        else:
            assert_bit_exact(got_m, want_m, what)
        lo, hi = sorted([float(rng.uniform(20, 2000)), float(rng.uniform(20, 8000))])
        x = synth.noise(seed, SPT) * np.float32(rng.choice([1.0, 0.0, 3.0]))
        want = oracle.fm_sine_run(lo, hi, SR, t, x, SPT)
        fm = abi.Module(abi.KIND_FM_SINE, abi.FmSineParams(lo, hi), sample_rate=SR)
        got = np.empty(2 * SPT, np.float32)
        fm.run_tick(t, [(abi.MX_MONO, x)], [(abi.MX_STEREO, got)])
        amp = (hi - lo) / 2.0
        if (lo + amp + amp * 3.0) * 2.0 * np.pi * ((t + SPT) / SR) >= 2.0 ** 40:
            assert_ulp(got, want, 1, what + f" FmSine {lo}-{hi}")
        else:
            assert_bit_exact(got, want, what + f" FmSine {lo}-{hi}")
    except Exception:
        bad += 1; print(what); traceback.print_exc(limit=2)
        if bad >= 4:
            break
print(f"{count} oscillator + fm cases, {bad} failures")
sys.exit(1 if bad else 0)
