"""CPU suite (`-m "not gpu"`): the oracle against the reference's golden vectors, the oracle's
restatement of Engine::run_tick semantics, and that the C-ABI library loads and exports every
symbol include/mixlab_gpu.h declares (no compute calls without a GPU)."""
import ctypes
import pathlib
import re

import numpy as np
import pytest

import oracle
import synth
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace

ROOT = pathlib.Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
REF_FIXTURES = pathlib.Path("/root/reference/fixtures/module/eq_three")
SPT = 735


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ---------------- oracle pinned by the reference's golden pair (src/module/eq_three.rs:150-167) ----------------
def test_oracle_eq_three_matches_reference_golden_prefix_one_call():
    x = np.fromfile(GOLDEN / "eq_three_chronos_prefix131072.f32.raw", dtype="<f4")
    y = np.fromfile(GOLDEN / "eq_three_chronos-eq_prefix131072.f32.raw", dtype="<f4")
    st = oracle.eq_three_new(44100.0)
    out = oracle.eq_three_run(st, (4.0, 0.0, 4.0), x)   # Decibel(4.0), Decibel(0.0), Decibel(4.0)
    assert np.array_equal(bits(out), bits(y))


def test_oracle_eq_three_matches_reference_golden_prefix_ticked():
    x = np.fromfile(GOLDEN / "eq_three_chronos_prefix131072.f32.raw", dtype="<f4")
    y = np.fromfile(GOLDEN / "eq_three_chronos-eq_prefix131072.f32.raw", dtype="<f4")
    st = oracle.eq_three_new(44100.0)
    out = np.concatenate([oracle.eq_three_run(st, (4.0, 0.0, 4.0), x[o:o + SPT]) for o in range(0, x.size, SPT)])
    assert np.array_equal(bits(out), bits(y))


def test_oracle_contract_mode_is_within_one_ulp_of_the_reference_golden_prefix_and_of_the_exact_mode():
    """The oracle's CONTRACT mode (explicit fma; the checker for MX_FLAG_FP_CONTRACT): on the reference's golden pair its f32 output is
    within 1 ULP of the expected file -- measured: identical, the f32 store absorbs the last-bit f64 differences (SURVEY 8c found the
    same with -ffp-contract=fast) -- and the flag really changes the arithmetic: the carried f64 poles differ from the exact mode's."""
    x = np.fromfile(GOLDEN / "eq_three_chronos_prefix131072.f32.raw", dtype="<f4")
    y = np.fromfile(GOLDEN / "eq_three_chronos-eq_prefix131072.f32.raw", dtype="<f4")
    st_fc, st_ex = oracle.eq_three_new(44100.0), oracle.eq_three_new(44100.0)
    with oracle.fp_contract():
        out = oracle.eq_three_run(st_fc, (4.0, 0.0, 4.0), x)
    assert oracle.lib.orc_get_fp_contract() == 0                      # the context manager restores the default
    oracle.eq_three_run(st_ex, (4.0, 0.0, 4.0), x)
    d = synth.ulp_diff(out, y)
    assert d.max() <= 1 and np.count_nonzero(d) <= 1
    assert list(st_fc.lo) != list(st_ex.lo), "contract mode left the same f64 poles as the exact mode: the flag did nothing"
    # Envelope decay, Amplifier depth, FIR accumulation: one rounding instead of two, <= 1 ULP of the f32 stored
    n = synth.noise(3, 20000)
    gate = np.ones(20000, np.float32); gate[12000:] = 0.0
    for run in (lambda: oracle.envelope_run(oracle.EnvState(), (25.0, 500.0, 0.8, 200.0), 48000.0, 0, gate, 20000),
                lambda: oracle.amplifier_run(0.9, 0.6, n, np.abs(n[:10000])),
                lambda: oracle.fir_run(np.linspace(0.5, -0.25, 33), np.zeros(64, np.float32), n)):
        a = run()
        with oracle.fp_contract():
            b = run()
        assert synth.ulp_diff(a, b).max() <= 1


@pytest.mark.skipif(not REF_FIXTURES.exists(), reason="full fixture only exists where /root/reference is mounted")
def test_oracle_eq_three_matches_full_reference_fixture():
    x = np.fromfile(REF_FIXTURES / "chronos.f32.raw", dtype="<f4")
    y = np.fromfile(REF_FIXTURES / "chronos-eq.f32.raw", dtype="<f4")
    assert x.size == 355285
    st = oracle.eq_three_new(44100.0)
    assert np.array_equal(bits(oracle.eq_three_run(st, (4.0, 0.0, 4.0), x)), bits(y))
    # and the committed prefix really is a prefix of the reference files
    assert np.array_equal(np.fromfile(GOLDEN / "eq_three_chronos_prefix131072.f32.raw", dtype="<f4"), x[:131072])
    assert np.array_equal(np.fromfile(GOLDEN / "eq_three_chronos-eq_prefix131072.f32.raw", dtype="<f4"), y[:131072])


# ---------------- oracle module semantics (source text is the only spec for these) ----------------
def test_oracle_decibel_and_coeff():
    assert oracle.lib.orc_decibel_to_linear(0.0) == 1.0
    assert abs(oracle.lib.orc_decibel_to_linear(20.0) - 10.0) < 1e-12
    assert abs(oracle.lib.orc_lowpass_coeff(420.0, 44100.0) - 2.0 * np.sin(np.pi * 420.0 / 44100.0)) < 1e-15


def test_oracle_mixer_order_and_cue():
    a = np.array([1e8, 1.0], np.float32)
    b = np.array([-1e8, 1.0], np.float32)
    c = np.array([1.0, 1.0], np.float32)
    m, cue = oracle.mixer_run([(0.0, 1.0, False), (0.0, 1.0, True), (0.0, 1.0, True)], [a, b, c], 2)
    assert m[0] == 1.0 and cue[0] == np.float32(-1e8) + np.float32(1.0)   # ((0+1e8)-1e8)+1 in f32, sequentially
    m2, _ = oracle.mixer_run([(0.0, 1.0, False)] * 3, [a, c, b], 2)
    assert m2[0] == 0.0                                                    # different order, different f32 result
    # MixerChannelParams::default: fader 0.0 => silence (protocol/src/lib.rs:342-347)
    m3, c3 = oracle.mixer_run([(0.0, 0.0, False)], [a], 2)
    assert not m3.any() and not c3.any()


def test_oracle_envelope_shape():
    p = (25.0, 500.0, 0.8, 200.0)
    st = oracle.EnvState()
    gate = np.ones(44100, np.float32)
    out = oracle.envelope_run(st, p, 44100.0, 0, gate, gate.size)
    assert out[0] == 0.0 and abs(out[int(0.025 * 44100)] - 1.0) < 2e-3          # end of attack
    assert abs(out[-1] - 0.8) < 1e-6                                            # sustain after decay
    off = oracle.envelope_run(st, p, 44100.0, 44100, np.zeros(22050, np.float32), 22050)
    assert abs(off[0] - 0.8) < 1e-6 and off[int(0.2 * 44100) + 1] == 0.0        # released after 200 ms
    # exact-compare gate semantics: 0.5 is neither on nor off (envelope.rs:102,107)
    st2 = oracle.EnvState()
    assert not oracle.envelope_run(st2, p, 44100.0, 0, np.full(100, 0.5, np.float32), 100).any()


def test_oracle_oscillator_waveforms():
    mono, stereo = oracle.oscillator_run(441.0, abi.WAVE_SAW, 44100.0, 0, 200)
    assert mono[0] == 0.0 and np.array_equal(stereo[0::2], mono) and np.array_equal(stereo[1::2], mono)
    assert np.all(np.abs(mono) <= 1.0)
    sq, _ = oracle.oscillator_run(441.0, abi.WAVE_SQUARE, 44100.0, 0, 200)
    assert set(np.unique(sq)) <= {-1.0, 1.0} and sq[0] == 1.0     # sign(+0.0) = +1 (oscillator.rs:15-23)
    on, _ = oracle.oscillator_run(1.0, abi.WAVE_ON, 44100.0, 0, 8)
    off, _ = oracle.oscillator_run(1.0, abi.WAVE_OFF, 44100.0, 0, 8)
    assert (on == 1.0).all() and (off == 0.0).all()


def test_oracle_amplifier_control_semantics():
    x = synth.noise(1, 16)
    assert np.array_equal(oracle.amplifier_run(1.0, 0.0, x, None), x)                      # depth 0: unity
    ctl = np.zeros(8, np.float32)
    assert not oracle.amplifier_run(1.0, 1.0, x, ctl).any()                                # depth 1, control 0: silence
    assert np.array_equal(oracle.amplifier_run(1.0, 1.0, x, None), x)                      # disconnected control = 1.0


# ---------------- oracle graph runner == Engine::run_tick semantics ----------------
def test_graph_run_order_is_dfs_from_terminals():
    ws = Workspace()
    o = ws.oscillator(100.0, abi.WAVE_SINE)
    m = ws.mixer([(0.0, 1.0, False)])
    p = ws.plotter()
    ws.connect(o, 1, m, 0)
    ws.connect(m, 0, p, 0)
    assert oracle.OracleGraph(ws).run_order() == [o, m, p]


def test_graph_plotter_fires_on_sixth_tick_and_disconnected_reads_zero():
    ws = Workspace()
    o = ws.oscillator(100.0, abi.WAVE_ON)
    m = ws.mixer([(0.0, 1.0, True), (0.0, 1.0, True)])   # input 1 left unconnected
    p = ws.plotter()
    ws.connect(o, 1, m, 0); ws.connect(m, 1, p, 0)
    og = oracle.OracleGraph(ws)
    fired = []
    for t in range(12):
        og.run_tick(t)
        fired.append(og.plotter(p) is not None)
        assert (og.output(m, 1) == 1.0).all()            # cue = 1.0 + 0.0 (zero buffer)
    assert fired == [(t + 1) % 6 == 0 for t in range(12)]


def test_graph_cycle_back_edge_reads_disconnected():
    # amp0 -> amp1 -> amp0 (cycle) -> mixer: whichever amp runs first sees its input Disconnected
    # (engine.rs:479-482), i.e. zeros; nothing hangs and nothing reads last tick's buffer.
    ws = Workspace()
    a0 = ws.amplifier(1.0, 0.0); a1 = ws.amplifier(1.0, 0.0); m = ws.mixer([(0.0, 1.0, False)])
    ws.connect(a1, 0, a0, 0); ws.connect(a0, 0, a1, 0); ws.connect(a1, 0, m, 0)
    og = oracle.OracleGraph(ws)
    assert sorted(og.run_order()) == [a0, a1, m]
    og.run_tick(0); og.run_tick(1)
    assert not og.output(m, 0).any()


def test_graph_type_mismatch_is_refused():
    ws = Workspace()
    o = ws.oscillator(100.0, abi.WAVE_SINE); e = ws.eq_three(0, 0, 0)
    ws.connect(o, 1, e, 0)   # Stereo -> Mono: Workspace::connect refuses (workspace.rs:97-114)
    with pytest.raises(RuntimeError):
        oracle.OracleGraph(ws)


def test_synth_noise_is_deterministic_and_in_range():
    a, b = synth.noise(3, 1000), synth.noise(3, 1000)
    assert np.array_equal(a, b) and a.min() >= -1.0 and a.max() < 1.0 and a.dtype == np.float32
    assert not np.array_equal(a, synth.noise(4, 1000))
    assert synth.ulp_diff(np.array([1.0, -0.0], np.float32), np.array([np.nextafter(np.float32(1.0), np.float32(2.0)), 0.0], np.float32)).tolist() == [1, 0]


# ---------------- the C-ABI library: loads, and exports every declared symbol ----------------
def declared_functions():
    text = (ROOT / "include" / "mixlab_gpu.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mx_[a-z0-9_]+)\s*\(", text)))


def test_abi_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 20
    lib = ctypes.CDLL(str(abi.LIB_PATH))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/mixlab_gpu.h but not exported: {missing}"


def test_abi_version_and_error_channel_without_gpu():
    assert abi.lib.mx_abi_version() == 4
    n = abi.lib.mx_device_count()
    if n <= 0:   # CPU box: the call must fail cleanly and say why, not crash or fall back
        assert n == abi.MX_ERR_DEVICE and b"hip" in abi.lib.mx_last_error().lower()
        with pytest.raises(abi.MxError):
            Workspace().build()   # no silent CPU path


def test_param_struct_layouts_match_header():
    assert ctypes.sizeof(abi.MixerChannelParams) == 24 and ctypes.sizeof(abi.EqThreeParams) == 24
    assert ctypes.sizeof(abi.EnvelopeParams) == 32 and ctypes.sizeof(abi.AmplifierParams) == 16
    assert ctypes.sizeof(abi.OscillatorParams) == 16 and ctypes.sizeof(abi.FmSineParams) == 16
    assert ctypes.sizeof(abi.Node) == 16 and ctypes.sizeof(abi.Edge) == 16 and ctypes.sizeof(abi.GraphOpts) == 32


# ---------------- pixel path: identities and oracle geometry (CPU) ----------------
def test_div255_identity_used_by_crossfade_kernel():
    x = np.arange(255 * 255 + 1, dtype=np.uint32)   # every value a*f + b*(255-f) can take
    assert np.array_equal(x // 255, (x + 1 + (x >> 8)) >> 8)


def test_oracle_scaler_geometry_and_unify():
    import oracle_video as ov
    assert ov.scaler_geometry(1280, 720, 1920, 1080) == (1920, 1080, 0, 0)
    assert ov.scaler_geometry(640, 480, 1920, 1080) == (1440, 1080, 240, 0)      # pillarbox, even offset
    assert ov.scaler_geometry(1920, 1080, 560, 350) == (560, 314, 0, 18)          # monitor size (monitor.rs:21-22)
    assert ov.unify(640, 360, 321, 241) == (640, 360)
    assert ov.unify(321, 241, 100, 100) == (322, 242)                             # rounded UP to even (video_mixer.rs:286-290)


def test_oracle_crossfade_factor_and_blank():
    import oracle_video as ov
    f = oracle.lib.orc_crossfade_factor
    assert [f(1.0), f(0.0), f(0.5), f(2.0), f(-1.0), f(float("nan"))] == [255, 0, 127, 255, 0, 0]   # `as u8` saturates/truncates
    fr = ov.HostFrame(66, 34); fr.planes[0][:] = 7; ov.blank(fr)
    y, u, v = fr.visible()
    assert not y.any() and (u == 0x80).all() and (v == 0x80).all()
    a = ov.HostFrame(66, 34).fill(1)
    out = ov.HostFrame(66, 34); ov.blank(out); ov.crossfade(out, a, None, 1.0)     # fade 255: exactly A
    assert all(np.array_equal(o, i) for o, i in zip(out.visible(), a.visible()))
    out2 = ov.HostFrame(66, 34); ov.blank(out2); ov.crossfade(out2, a, None, 0.0)  # fade 0: exactly the blank B
    assert not out2.visible()[0].any() and (out2.visible()[1] == 0x80).all()


def test_oracle_bicubic_is_identity_preserving_and_bounded():
    import oracle_video as ov
    src = ov.HostFrame(64, 48)
    for p in src.planes:
        p[:] = 200
    dst = ov.HostFrame(128, 96); ov.dynamic_scale(src, dst)
    assert all((v == 200).all() for v in dst.visible())     # taps sum to exactly 1.0 in Q14: flat stays flat
    # downscaling widens the kernel (2 * ceil(2 * src / dst) + 2 taps, normalised to 1.0 in Q14): flat stays flat, and a
    # one-pixel checkerboard -- which a 4-tap kernel would alias into stripes -- averages out to mid grey
    small = ov.HostFrame(16, 12); ov.dynamic_scale(src, small)
    assert all((v == 200).all() for v in small.visible())
    assert ov.lib.orc_bicubic_tap_count(1920, 560) == 16 and ov.lib.orc_bicubic_tap_count(560, 1920) == 4 and ov.lib.orc_bicubic_tap_count(100, 100) == 4
    chk = ov.HostFrame(256, 256)
    yy, xx = np.mgrid[0:256, 0:256]
    chk.planes[0][:256, :256] = np.where((xx + yy) % 2 == 0, 16, 235)
    for p in chk.planes[1:]:
        p[:] = 128
    out = ov.HostFrame(64, 64); ov.dynamic_scale(chk, out)
    y = out.visible()[0][4:-4, 4:-4].astype(int)
    assert abs(int(y.mean()) - 125) <= 2 and y.max() - y.min() <= 6


# ------------------------------------------------------------------------------------------------
# the scaler SPEC, pinned independently of both implementations: tap tables derived from the text of DESIGN.md section 6 with
# exact rationals (tests/golden/make_bicubic_taps.py) -- the oracle's and the product's generators must both reproduce them
# ------------------------------------------------------------------------------------------------
def _golden_taps():
    import json
    return [json.loads(p.read_text()) for p in sorted((ROOT / "tests" / "golden").glob("bicubic_taps_*.json"))]


def test_bicubic_tap_golden_files_cover_up_down_identity_and_odd_sizes():
    geos = {(g["src"], g["dst"]) for g in _golden_taps()}
    assert {(720, 1080), (1080, 1080), (1000, 1001), (1080, 635), (1280, 100), (959, 539)} <= geos
    for g in _golden_taps():
        assert len(g["first"]) == g["dst"] and all(len(c) == g["n_taps"] and sum(c) == 16384 for c in g["coef"])


def test_oracle_bicubic_taps_equal_the_exact_rational_spec():
    import ctypes as C
    import oracle
    oracle.lib.orc_bicubic_taps_n.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    oracle.lib.orc_bicubic_tap_count.argtypes = [C.c_uint32, C.c_uint32]
    oracle.lib.orc_bicubic_tap_count.restype = C.c_uint32
    for g in _golden_taps():
        n = oracle.lib.orc_bicubic_tap_count(g["src"], g["dst"])
        assert n == g["n_taps"]
        first = C.c_int32(); coef = (C.c_int32 * n)()
        for o in range(g["dst"]):
            oracle.lib.orc_bicubic_taps_n(o, g["src"], g["dst"], C.byref(first), coef)
            assert first.value == g["first"][o] and list(coef) == g["coef"][o], f"oracle taps {g['src']} -> {g['dst']}, output {o}"


def test_product_scaler_taps_equal_the_exact_rational_spec():
    import ctypes as C
    from mixlab_amd import abi
    lib = abi.lib
    lib.mx_video_scaler_tap_count.argtypes = [C.c_uint32, C.c_uint32]
    lib.mx_video_scaler_tap_count.restype = C.c_uint32
    lib.mx_video_scaler_taps.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
    for g in _golden_taps():
        n = lib.mx_video_scaler_tap_count(g["src"], g["dst"])
        assert n == g["n_taps"]
        first = (C.c_int32 * g["dst"])(); coef = (C.c_int32 * (g["dst"] * n))(); nt = C.c_uint32()
        assert lib.mx_video_scaler_taps(g["src"], g["dst"], first, coef, C.byref(nt)) == 0 and nt.value == n
        assert list(first) == g["first"], f"product first-tap indices {g['src']} -> {g['dst']}"
        got = [list(coef[o * n:(o + 1) * n]) for o in range(g["dst"])]
        assert got == g["coef"], f"product coefficients {g['src']} -> {g['dst']}"


def test_oracle_deep_conversion_is_round_to_nearest_with_the_ignored_bits_ignored():
    """orc_deep_to_8 against the rule stated in include/mixlab_gpu.h, on every value of every depth: min(255, floor(v / 2^(b-8) + 1/2)) in exact integers, the
    same whatever sits in the bits a format ignores, the semi-planar chroma de-interleaved."""
    import oracle_video as ov
    for fmt, (lay, bits, shift) in {10: (0, 10, 0), 11: (1, 10, 0), 12: (2, 10, 0), 13: (0, 10, 6), 14: (0, 12, 0), 15: (1, 12, 0), 16: (2, 12, 0),
                                    17: (0, 16, 0), 18: (1, 16, 0), 19: (2, 16, 0), 20: (0, 16, 0)}.items():
        n = 1 << bits
        vals = np.arange(n, dtype=np.int64)
        want = np.minimum(255, (2 * vals + (1 << (bits - 8))) >> (bits - 7)).astype(np.uint8)       # floor(v / 2^(b-8) + 1/2)
        if bits == 10:
            assert want[1] == 0 and want[2] == 1 and want[1021] == 255 and want[1018] == 255 and want[1017] == 254
        w, h = 256, max(4, 2 * n // 256)                     # every value in the luma plane, twice
        cw, ch = (0 if lay == 2 else 1), (1 if lay == 0 else 0)
        y = np.resize(vals, (h, w)); u = np.resize(vals[::-1], (h >> ch, w >> cw)); v = np.resize(vals[3:], (h >> ch, w >> cw))
        for junk in (0, (1 << (16 - bits)) - 1, 0x2A & ((1 << (16 - bits)) - 1)):
            pack = (lambda a: ((a << shift) | junk)) if shift else (lambda a: (a | (junk << bits)))
            if fmt in (13, 20):
                uv = np.empty((h // 2, w), np.int64); uv[:, 0::2] = u; uv[:, 1::2] = v
                planes = [pack(y).astype(np.uint16), pack(uv).astype(np.uint16)]
            else:
                planes = [pack(y).astype(np.uint16), pack(u).astype(np.uint16), pack(v).astype(np.uint16)]
            f = ov.deep_to_8(planes, w, h, fmt)
            assert f.fmt == lay
            for got, src in zip(f.visible(), (y, u, v)):
                assert np.array_equal(got, want[src]), (fmt, junk)


def ulp_distance_f32(a, b):
    """distance in units in the last place between two float32 arrays (monotone integer mapping of the bit patterns)"""
    def key(x):
        i = np.ascontiguousarray(x, dtype=np.float32).view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7FFFFFFF), i)
    return np.abs(key(a) - key(b))


def test_oracle_fir_and_resampler_agree_with_scipy_within_one_ulp():
    """The two build-specified audio modules have no reference counterpart, so their oracle and kernels come from one paragraph of DESIGN.md.  This pins the
    MATH independently: tests/golden/fir_resample_scipy.npz holds scipy.signal.lfilter / upfirdn outputs (f64, scipy's own summation order) for seeded input,
    a 128-tap FIR and three resampling ratios (160/147, 2/3, 3/1); the oracle's f32 outputs are within 1 ULP of them everywhere (or within 2^-40 absolute at a
    zero crossing), and equal in all but a few samples per thousand."""
    import pathlib
    import oracle as o
    z = np.load(pathlib.Path(__file__).parent / "golden" / "fir_resample_scipy.npz")
    x = np.ascontiguousarray(z["x"]).reshape(-1)
    frames = z["x"].shape[0]
    def close(got, want, what):
        got, want = np.asarray(got, np.float32).reshape(-1), np.asarray(want, np.float32).reshape(-1)
        d = ulp_distance_f32(got, want)
        bad = (d > 1) & (np.abs(got.astype(np.float64) - want.astype(np.float64)) > 2.0 ** -40)
        assert not bad.any(), f"{what}: {int(bad.sum())} samples beyond 1 ULP, first {int(np.flatnonzero(bad)[0])}"
        assert (d != 0).mean() < 5e-3, f"{what}: {(d != 0).mean():.4f} of the samples differ"
    taps = z["fir_taps"]
    close(o.fir_run(taps, np.zeros(2 * (taps.size - 1), np.float32), x), z["y_fir"], "FIR vs scipy.signal.lfilter")
    for name in "abc":
        up, down, tpp = (int(v) for v in z[f"rs_{name}_ratio"])
        table = z[f"rs_{name}_table"]
        want = z[f"rs_{name}_y"]
        got = o.resample_run(table, up, down, np.zeros(2 * max(1, tpp - 1), np.float32), 0, 0, x, want.shape[0])
        close(got, want, f"resampler {up}/{down} vs scipy.signal.upfirdn")
    assert frames == 735 * 8


def test_oracle_packed_rgb_conversion_equals_the_exact_rational_matrix():
    """The build-specified packed RGB -> yuv444 conversion (what an rgb24 / bgra scaler input stands for): the oracle's hard-coded integers
    against tests/golden/rgb_matrix_bt709.json, which make_rgb_matrix.py derives from the BT.709 primaries with exact rationals."""
    import json
    import pathlib
    import numpy as np
    import oracle_video as ov
    m = json.load(open(pathlib.Path(__file__).parent / "golden" / "rgb_matrix_bt709.json"))
    rng = np.random.default_rng(7)
    rgb = rng.integers(0, 256, size=(6, 10, 3), dtype=np.uint8)
    rgb[0, :4] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 0, 255]]
    R, G, B = (rgb[..., k].astype(np.int64) for k in range(3))
    want = [((m[key][0] * R + m[key][1] * G + m[key][2] * B + 128) >> 8) + off for key, off in (("y", 16), ("cb", 128), ("cr", 128))]
    got = ov.packed_rgb_to_yuv444(rgb, 4).visible()
    for p in range(3):
        assert np.array_equal(got[p], want[p].astype(np.uint8)), f"plane {p}"
    bgra = np.concatenate([rgb[..., ::-1], rng.integers(0, 256, size=(6, 10, 1), dtype=np.uint8)], axis=2)   # alpha is ignored
    got2 = ov.packed_rgb_to_yuv444(bgra, 5).visible()
    for p in range(3):
        assert np.array_equal(got2[p], got[p])
    alpha = rng.integers(0, 256, size=(6, 10, 1), dtype=np.uint8)
    for fmt, pix in ((23, rgb[..., ::-1]), (24, np.concatenate([rgb, alpha], axis=2)), (25, np.concatenate([alpha, rgb], axis=2)), (26, np.concatenate([alpha, rgb[..., ::-1]], axis=2))):
        for p, plane in enumerate(ov.packed_rgb_to_yuv444(pix, fmt).visible()):       # bgr24, rgba, argb, abgr: the same pixels in another byte order
            assert np.array_equal(plane, got[p]), (fmt, p)
    assert int(want[0].min()) >= 16 and int(want[0].max()) <= 235


def test_crossfade_division_identity_the_packed_kernel_relies_on():
    """mx_k_video.hip fade_pk: x / 255 == (x1 + (x1 >> 8)) >> 8 with x1 = x + 1, for every x = a f + b (255 - f) <= 255 * 255, inside 16 bits."""
    import numpy as np
    x = np.arange(0, 255 * 255 + 1, dtype=np.int64)
    x1 = x + 1
    assert np.array_equal((x1 + (x1 >> 8)) >> 8, x // 255)
    assert int((x1 + (x1 >> 8)).max()) < 65536


def test_graft_entry_build_is_what_the_driver_calls_and_it_passes():
    """build() compiles (here: finds up to date) the library and the oracle and checks the loaded library against the header's ABI version."""
    import __graft_entry__ as entry
    entry.build()
