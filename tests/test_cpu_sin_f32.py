"""mx_sin_f32.hpp on the CPU (the header the device compiles, built here with g++): its double-double slow path IS the correctly rounded f32 of the real sine
(against a 130-digit pure-Python reference), for module-shaped arguments -- n 2.0 pi over hours of a tone, co t up to 1e6 rad and beyond -- and for arguments
picked next to f32 rounding boundaries; the Ziv form over the host libm equals both the slow path and (float)glibc_sin wherever it takes the fast path."""
import pathlib
import subprocess
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
from sin_reference import sin_f32  # noqa: E402


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = tmp_path_factory.mktemp("sin") / "sin_f32_check"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-o", str(exe), str(ROOT / "tests" / "helpers" / "sin_f32_check.cpp")], check=True)

    def run(xs):
        xs = np.ascontiguousarray(xs, dtype="<f8")
        out = subprocess.run([str(exe)], input=xs.tobytes(), capture_output=True, check=True).stdout
        return np.frombuffer(out, dtype="<f4").reshape(-1, 3)
    return run


def module_shaped_arguments(rng, n):
    """the f64 arguments the modules form: Oscillator n 2.0 pi with n = (t / SR) freq (oscillator.rs:69-70,77), FmSine (co) (t / SR) (fm_sine.rs:44-52)"""
    xs = []
    for _ in range(n // 2):
        sr = (44100.0, 48000.0)[int(rng.integers(2))]
        t = float(rng.integers(0, 2 ** int(rng.integers(8, 34))))
        f = float(rng.choice([27.5, 110.0, 440.0, 1000.0, 19999.0, float(rng.uniform(0.01, 22000.0))]))
        xs.append((t / sr) * f * 2.0 * np.pi)
        co = (f + float(rng.uniform(0.0, 500.0)) * float(np.float32(rng.uniform(-1.0, 1.0)))) * 2.0 * np.pi
        xs.append(co * (t / sr))
    return np.array(xs)


def test_slow_path_is_the_correctly_rounded_real_sine(checker):
    rng = np.random.default_rng(0x5151)
    xs = np.concatenate([module_shaped_arguments(rng, 1200), rng.uniform(-1e6, 1e6, 300), rng.uniform(-8.0, 8.0, 300), 10.0 ** rng.uniform(-30.0, 11.9, 200),
                         np.array([0.0, -0.0, 1e-310, 5e-324, np.pi, -np.pi / 2, 1.0e12, 3.0e-39, 1.5707963267948966, 6.283185307179586, 2.0 ** 39 + 0.5])])
    got = checker(xs)
    for i, x in enumerate(xs):
        want = sin_f32(float(x))
        assert got[i, 0].view(np.uint32) == want.view(np.uint32), (float(x).hex(), float(got[i, 0]), float(want))
        assert got[i, 1].view(np.uint32) == want.view(np.uint32), ("ziv", float(x).hex(), float(got[i, 1]), float(want))


def test_arguments_next_to_a_rounding_boundary(checker):
    """Arguments whose real sine lies within a few f64 ulp of the midpoint of two floats, constructed: x = asin(midpoint) and its f64 neighbours.  There the cast of
    an f64 sine depends on that sine's last bits (the casts of two good libms can differ); the slow path and the Ziv form must give the real sine's float."""
    rng = np.random.default_rng(0xB0DA)
    f = rng.uniform(0.3, 0.9, 80).astype(np.float32)
    mid = (f.astype(np.float64) + np.nextafter(f, np.float32(2.0)).astype(np.float64)) * 0.5
    x0 = np.arcsin(mid)
    pick = np.concatenate([x0 + k * np.spacing(x0) for k in range(-4, 5)])
    y = np.sin(pick)
    low = (y.view(np.uint64) & np.uint64(0x1FFFFFFF)).astype(np.int64)
    assert np.count_nonzero(np.abs(low - 0x10000000) <= 16) >= pick.size // 2          # the construction works: these ARE next to boundaries
    got = checker(pick)
    n_cast_differs = 0
    for i, x in enumerate(pick):
        want = sin_f32(float(x))
        assert got[i, 0].view(np.uint32) == want.view(np.uint32), (float(x).hex(), float(got[i, 0]), float(want))
        assert got[i, 1].view(np.uint32) == want.view(np.uint32), ("ziv", float(x).hex())
        n_cast_differs += int(got[i, 2].view(np.uint32) != want.view(np.uint32))
    # (float)glibc_sin is the real sine's float except where the real sine is within glibc's own error of the boundary -- a band these arguments were built to hit
    assert n_cast_differs <= pick.size // 4


def test_ziv_over_glibc_equals_the_cast_of_glibc_on_bulk_arguments(checker):
    """2 M module-shaped arguments: the Ziv form (fast path almost everywhere) against (float)glibc_sin -- what the oracle computes.  They may differ only where
    the slow path ran AND the real sine lies within glibc's error of a rounding boundary: none expected in 2 M."""
    rng = np.random.default_rng(7)
    t = rng.integers(0, 2 ** 31, 2_000_000).astype(np.float64)
    f = rng.uniform(20.0, 20000.0, 2_000_000)
    xs = (t / 44100.0) * f * 2.0 * np.pi
    got = checker(xs)
    assert int(np.count_nonzero(got[:, 1].view(np.uint32) != got[:, 2].view(np.uint32))) == 0
    assert int(np.count_nonzero(got[:, 0].view(np.uint32) != got[:, 2].view(np.uint32))) <= 1
