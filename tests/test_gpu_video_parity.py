"""GPU parity of the pixel path (integer work: bit-exact on the visible area).

Reference-following and unpinned by reference tests: blank, cross-fade, geometry, VideoMixer state
machine.  Build-specified (no reference arithmetic exists): bicubic scaler, YUV420P->RGBA."""
import numpy as np
import pytest

import oracle_video as ov
from mixlab_amd import abi, video

pytestmark = pytest.mark.gpu


def upload(hf: ov.HostFrame) -> video.DFrame:
    d = video.DFrame(hf.w, hf.h)
    y, u, v = hf.visible()
    return d.upload(y, u, v)


def assert_frame_equal(dev: video.DFrame, host: ov.HostFrame, what=""):
    assert (dev.width, dev.height) == (host.w, host.h), what
    for p, (g, w) in enumerate(zip(dev.download(), host.visible())):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{what}: plane {p}: {len(bad)} pixels differ, first {bad[:3].tolist()} got {g[tuple(bad[0])]} want {w[tuple(bad[0])]}"


def test_blank_frame():
    d = video.DFrame(130, 70)
    y, u, v = d.download()
    assert not y.any() and (u == 0x80).all() and (v == 0x80).all()
    d.upload(np.full((70, 130), 9, np.uint8), np.full((35, 65), 9, np.uint8), np.full((35, 65), 9, np.uint8))
    video.blank(d)
    y, u, v = d.download()
    assert not y.any() and (u == 0x80).all() and (v == 0x80).all()


@pytest.mark.parametrize("size", [(1920, 1080), (1280, 720), (66, 34), (640, 480), (2, 2)])
@pytest.mark.parametrize("fader", [1.0, 0.0, 0.5, 0.75, 0.1, 0.999, 1.5, -0.2])
def test_crossfade_bit_exact(size, fader):
    w, h = size
    ha, hb = ov.HostFrame(w, h).fill(1), ov.HostFrame(w, h).fill(2, seed=5)
    want = ov.HostFrame(w, h); ov.blank(want); ov.crossfade(want, ha, hb, fader)
    da, db, out = upload(ha), upload(hb), video.DFrame(w, h)
    video.crossfade(out, da, db, fader)
    assert_frame_equal(out, want, f"crossfade {size} fader {fader}")


@pytest.mark.parametrize("which", ["a_only", "b_only", "none"])
def test_crossfade_missing_channel_reads_blank(which):
    w, h = 322, 182
    ha = ov.HostFrame(w, h).fill(3)
    a = ha if which == "a_only" else None
    b = ha if which == "b_only" else None
    want = ov.HostFrame(w, h); ov.blank(want); ov.crossfade(want, a, b, 0.6)
    da = upload(ha)
    out = video.DFrame(w, h)
    video.crossfade(out, da if a else None, da if b else None, 0.6)
    assert_frame_equal(out, want, which)


@pytest.mark.parametrize("geom", [((1280, 720), (1920, 1080)), ((640, 480), (1920, 1080)), ((1920, 1080), (560, 350)),
                                  ((1000, 300), (640, 640)), ((64, 64), (64, 64)), ((322, 182), (1120, 700)),
                                  # tiles of 40 output rows (upscales whose 40-row windows fit one staging pass) and ratios either side of that class
                                  ((640, 360), (960, 540)), ((660, 372), (960, 542)), ((400, 300), (800, 600)), ((500, 282), (730, 412)), ((96, 54), (134, 76)),
                                  # downscales (widened kernel): monitor sizes, a 12.8x shrink, a ratio just above 1, awkward sizes
                                  ((1920, 1080), (1120, 700)), ((4096, 2160), (320, 180)), ((1922, 1082), (1920, 1080)), ((1002, 564), (400, 226))])
def test_dynamic_scale_letterbox_bit_exact_vs_build_spec(geom):
    (iw, ih), (ow, oh) = geom
    assert video.scale_geometry(iw, ih, ow, oh) == ov.scaler_geometry(iw, ih, ow, oh)
    src = ov.HostFrame(iw, ih).fill(4, seed=2)
    want = ov.HostFrame(ow, oh); ov.dynamic_scale(src, want)
    dsrc, out = upload(src), video.DFrame(ow, oh)
    video.scale(dsrc, out)
    assert_frame_equal(out, want, f"scale {geom}")


@pytest.mark.parametrize("fmt", [video.PIXFMT_YUV422P, video.PIXFMT_YUV444P, video.PIXFMT_NV12], ids=["yuv422p", "yuv444p", "nv12"])
@pytest.mark.parametrize("geom", [((1280, 720), (1920, 1080)), ((320, 180), (320, 180)), ((640, 480), (1920, 1080)), ((1920, 1080), (560, 350)),
                                  ((66, 34), (640, 640)), ((1922, 1082), (1920, 1080))])
def test_scaler_input_of_another_pixel_format_is_converted_plane_by_plane(geom, fmt):
    """SwsContext::new(input, output) carries the input's pixel format (codec/src/ffmpeg/scale.rs:16-39) and the DynamicScaler's
    settings compare unequal when only the format differs (encode.rs:342-352): a 4:2:2 / 4:4:4 frame -- also one of the output's own
    size -- comes out as the letterboxed yuv420p picture, every plane resampled from its own size (build-specified, DESIGN "Scaler")."""
    (iw, ih), (ow, oh) = geom
    src = ov.HostFrame(iw, ih, fmt).fill(4, seed=2)
    want = ov.HostFrame(ow, oh); ov.dynamic_scale(src, want)
    dsrc = video.DFrame(iw, ih, fmt=fmt).upload(*src.visible())
    assert [a.shape for a in dsrc.download()] == [a.shape for a in src.visible()]
    for a, b in zip(dsrc.download(), src.visible()):
        assert np.array_equal(a, b)
    out = video.DFrame(ow, oh)
    video.scale(dsrc, out)
    assert_frame_equal(out, want, f"scale {geom} fmt {fmt}")
    if fmt == video.PIXFMT_NV12:                  # semi-planar is a storage layout: the same samples as planar 4:2:0 give the same picture
        y, uv = src.visible()
        planar = ov.HostFrame(iw, ih); planar.planes[0][:, :iw] = y; planar.planes[1][:, :iw // 2] = uv[:, 0::2]; planar.planes[2][:, :iw // 2] = uv[:, 1::2]
        if (iw, ih) != (ow, oh):
            ref = ov.HostFrame(ow, oh); ov.dynamic_scale(planar, ref)
            for a, b in zip(want.visible(), ref.visible()):
                assert np.array_equal(a, b)
    sc = video.Scaler(ow, oh)                     # the persistent scaler: the input is never "the frame itself" when its format differs
    res = sc.scale(dsrc)
    assert res.device_planes()[0] != dsrc.device_planes()[0]
    assert_frame_equal(res, want, f"persistent scaler {geom} fmt {fmt}")


@pytest.mark.parametrize("fmt", [video.PIXFMT_YUV410P, video.PIXFMT_YUV411P, video.PIXFMT_YUV440P], ids=["yuv410p", "yuv411p", "yuv440p"])
@pytest.mark.parametrize("geom", [((1280, 720), (1920, 1080)), ((320, 180), (320, 180)), ((640, 480), (1920, 1080)), ((1920, 1080), (560, 350)), ((68, 36), (640, 640))])
def test_scaler_inputs_with_quarter_and_vertical_only_chroma_subsampling(geom, fmt):
    """The other planar 8-bit layouts an AVPixelFormat descriptor can describe (pixfmt.rs:97-111: log2_chroma_w / log2_chroma_h up to 2): yuv410p,
    yuv411p, yuv440p as scaler inputs -- every plane resampled from ITS size into the yuv420p output (chroma upscaled up to 4x: the gather form
    where the tiled one's window would not fit), stateless and through the persistent scaler, also at the output's own size."""
    (iw, ih), (ow, oh) = geom
    src = ov.HostFrame(iw, ih, fmt).fill(4, seed=2)
    want = ov.HostFrame(ow, oh); ov.dynamic_scale(src, want)
    dsrc = video.DFrame(iw, ih, fmt=fmt).upload(*src.visible())
    for a, b in zip(dsrc.download(), src.visible()):
        assert a.shape == b.shape and np.array_equal(a, b)
    out = video.DFrame(ow, oh)
    video.scale(dsrc, out)
    assert_frame_equal(out, want, f"scale {geom} fmt {fmt}")
    res = video.Scaler(ow, oh).scale(dsrc)
    assert res.device_planes()[0] != dsrc.device_planes()[0]
    assert_frame_equal(res, want, f"persistent scaler {geom} fmt {fmt}")
    with pytest.raises(abi.MxError):
        video.DFrame(iw + 2, ih, fmt=video.PIXFMT_YUV410P)      # a quarter-width chroma plane needs a width that is a multiple of 4


@pytest.mark.parametrize("fmt", [video.PIXFMT_YUYV422, video.PIXFMT_UYVY422], ids=["yuyv422", "uyvy422"])
@pytest.mark.parametrize("geom", [((1280, 720), (1920, 1080)), ((640, 480), (640, 480)), ((1920, 1080), (560, 350)), ((70, 37), (640, 640))], ids=["720p-up", "same-size", "monitor-downscale", "tiny"])
def test_packed_422_scaler_inputs_are_the_yuv422p_frame_with_the_same_samples(geom, fmt):
    """yuyv422 / uyvy422 (what capture devices deliver; one plane, 2 bytes per pixel): the frame stands for the yuv422p frame with the same samples -- a byte
    shuffle, nothing to specify -- and is resampled like it: stateless, persistent scaler (twice: the pooled frame is reused), VideoMixer input."""
    (iw, ih), (ow, oh) = geom
    rng = np.random.default_rng(iw + 3 * ow + fmt)
    pix = rng.integers(0, 256, size=(ih, 2 * iw), dtype=np.uint8)
    d = video.DFrame(iw, ih, fmt=fmt).upload_packed(pix)
    assert np.array_equal(d.download()[0].reshape(ih, 2 * iw), pix)
    as422 = ov.yuyv_to_422p(pix, fmt)
    yo = 0 if fmt == video.PIXFMT_YUYV422 else 1
    assert np.array_equal(as422.visible()[0], pix[:, yo::2]) and np.array_equal(as422.visible()[1], pix[:, 1 - yo::4]) and np.array_equal(as422.visible()[2], pix[:, 3 - yo::4])
    want = ov.HostFrame(ow, oh); ov.blank(want); ov.dynamic_scale(as422, want)
    out = video.DFrame(ow, oh)
    video.scale(d, out)
    assert_frame_equal(out, want, f"scale {geom} fmt {fmt}")
    sc = video.Scaler(ow, oh)
    for _ in range(2):
        assert_frame_equal(sc.scale(d), want, f"persistent scaler {geom} fmt {fmt}")
    other = ov.HostFrame(ow, oh).fill(1, seed=3)
    m = video.VideoMixer(a=0, b=1, fader=0.25)
    om = ov.OracleVideoMixer(a=0, b=1, fader=0.25)
    prog, _a, _b = m.run_tick(0, [(d, (1, 30), (0, 1)), (video.DFrame(ow, oh).upload(*other.visible()), (1, 30), (0, 1)), None, None])
    want_prog = om.run_tick(0, [(as422, (1, 30), (0, 1)), (other, (1, 30), (0, 1)), None, None])
    for p, (x, y) in enumerate(zip(prog.download(), want_prog.visible())):
        assert np.array_equal(x, y), f"VideoMixer program, plane {p}"
    with pytest.raises(abi.MxError):
        video.DFrame(iw + 1, ih, fmt=fmt)


def _deep_planes(rng, w, h, fmt):
    """random planes of a format deeper than 8 bits with the edge values in, and garbage in the bits the format says are ignored"""
    lay, bits, shift = video.DEEP[fmt]
    cw, ch = (0 if lay == 2 else 1), (1 if lay == 0 else 0)
    top = (1 << bits) - 1
    def plane(ph, pw):
        v = rng.integers(0, top + 1, size=(ph, pw), dtype=np.uint32)
        v.flat[:12] = [0, 1, (1 << (bits - 9)) - 1, 1 << (bits - 9), (1 << (bits - 8)) + 1, 3 << (bits - 9), top - (5 << (bits - 10)), top - (1 << (bits - 8)), top - (3 << (bits - 9)) + 1, top - 2, top - 1, top]
        junk = rng.integers(0, 1 << (16 - bits), size=(ph, pw), dtype=np.uint32) if bits < 16 else 0
        return (((v << shift) | junk) if shift else (v | (junk << bits))).astype(np.uint16)
    y, u, v = plane(h, w), plane(h >> ch, w >> cw), plane(h >> ch, w >> cw)
    if fmt in (video.PIXFMT_P010, video.PIXFMT_P016):
        uv = np.empty((h >> 1, w), np.uint16); uv[:, 0::2] = u; uv[:, 1::2] = v
        return [y, uv]
    return [y, u, v]


@pytest.mark.parametrize("fmt", sorted(video.DEEP), ids=["yuv420p10", "yuv422p10", "yuv444p10", "p010", "yuv420p12", "yuv422p12", "yuv444p12", "yuv420p16", "yuv422p16", "yuv444p16", "p016"])
@pytest.mark.parametrize("geom", [((1280, 720), (1920, 1080)), ((320, 180), (320, 180)), ((1920, 1080), (560, 350)), ((66, 38), (640, 640)), ((3840, 2160), (1920, 1080))],
                         ids=["720p-up", "same-size", "monitor-downscale", "tiny-pillarbox", "2160p-down"])
def test_deep_scaler_inputs_stand_for_the_8_bit_frame_of_their_layout(geom, fmt):
    """10- / 12- / 16-bit YUV in 16-bit little-endian words (what a decoder of a deep stream delivers; pixfmt.rs:107-111 reads the depth off the descriptor):
    BUILD-SPECIFIED as the 8-bit frame of the same layout with samples min(255, (v + 2^(b-9)) >> (b - 8)) -- the ignored bits of a word really ignored, the top
    values clipped, p010's value in the high bits, the semi-planar chroma de-interleaved -- which is then resampled like any 8-bit input: stateless, through the persistent
    scaler (twice: its pooled 8-bit frame is reused), at the output's own size (4:2:0: the converted frame IS the result) and as a VideoMixer input."""
    (iw, ih), (ow, oh) = geom
    rng = np.random.default_rng(iw * 7 + ow + fmt)
    planes = _deep_planes(rng, iw, ih, fmt)
    d = video.DFrame(iw, ih, fmt=fmt).upload(*planes)
    for a, b in zip(d.download(), planes):
        assert a.shape == b.shape and np.array_equal(a, b)
    as8 = ov.deep_to_8(planes, iw, ih, fmt)
    want = ov.HostFrame(ow, oh); ov.blank(want); ov.dynamic_scale(as8, want)
    out = video.DFrame(ow, oh)
    video.scale(d, out)
    assert_frame_equal(out, want, f"scale {geom} fmt {fmt}")
    sc = video.Scaler(ow, oh)
    for _ in range(2):
        res = sc.scale(d)
        assert res.device_planes()[0] != d.device_planes()[0]
        assert_frame_equal(res, want, f"persistent scaler {geom} fmt {fmt}")
    if (iw, ih) == (ow, oh) and as8.fmt == 0:
        for x, y in zip(out.download(), as8.visible()):
            assert np.array_equal(x, y)
    if iw <= 1920:
        other = ov.HostFrame(ow, oh).fill(2, seed=9)
        m = video.VideoMixer(a=1, b=0, fader=0.7)
        om = ov.OracleVideoMixer(a=1, b=0, fader=0.7)
        prog, _a, _b = m.run_tick(0, [(d, (1, 30), (0, 1)), (video.DFrame(ow, oh).upload(*other.visible()), (1, 30), (0, 1)), None, None])
        want_prog = om.run_tick(0, [(as8, (1, 30), (0, 1)), (other, (1, 30), (0, 1)), None, None])
        for p, (x, y) in enumerate(zip(prog.download(), want_prog.visible())):
            assert np.array_equal(x, y), f"VideoMixer program, plane {p}"
    with pytest.raises(abi.MxError):
        video.DFrame(iw + 1, ih, fmt=video.PIXFMT_P010)          # 4:2:0: even sizes


@pytest.mark.parametrize("src,dst", [((320, 180), (480, 270)), ((64, 64), (320, 180)), ((1280, 720), (560, 350)), ((320, 180), (320, 180)), ((321, 181), (320, 180))],
                         ids=["up-1.5x", "pillarbox", "monitor-downscale", "same-size", "odd-size"])
def test_gray8_scaler_input_stands_for_the_yuv444_frame_with_neutral_chroma(src, dst):
    """gray8 (one luma plane, any size): BUILD-SPECIFIED as the yuv444p frame with U = V = 0x80, resampled like any 4:4:4 input -- the luma as the
    oracle scales it, the chroma planes neutral wherever the picture is (flat stays flat: the taps sum to exactly 1) -- stateless, persistent
    scaler (twice: its pooled 4:4:4 frame is reused) and as a VideoMixer input."""
    rng = np.random.default_rng(src[0] * 11 + dst[0])
    w, h = src
    gray = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    d = video.DFrame(w, h, fmt=video.PIXFMT_GRAY8).upload_packed(gray)
    assert np.array_equal(d.download()[0][..., 0], gray)
    as444 = ov.HostFrame(w, h, 2)
    as444.planes[0][:, :w] = gray; as444.planes[1][:, :w] = 0x80; as444.planes[2][:, :w] = 0x80
    want = ov.HostFrame(*dst); ov.blank(want); ov.dynamic_scale(as444, want)
    out = video.DFrame(*dst)
    video.scale(d, out)
    for p, (x, y) in enumerate(zip(out.download(), want.visible())):
        assert np.array_equal(x, y), f"stateless scale, plane {p}"
    assert (out.download()[1] == 0x80).all() and (out.download()[2] == 0x80).all()
    sc = video.Scaler(*dst)
    for _ in range(2):
        res = sc.scale(d)
        for p, (x, y) in enumerate(zip(res.download(), want.visible())):
            assert np.array_equal(x, y), f"persistent scaler, plane {p}"
    other = ov.HostFrame(*dst).fill(3, seed=5)
    m = video.VideoMixer(a=0, b=1, fader=0.4)
    om = ov.OracleVideoMixer(a=0, b=1, fader=0.4)
    prog, _a, _b = m.run_tick(0, [(d, (1, 30), (0, 1)), (video.DFrame(*dst).upload(*other.visible()), (1, 30), (0, 1)), None, None])
    want_prog = om.run_tick(0, [(as444, (1, 30), (0, 1)), (other, (1, 30), (0, 1)), None, None])
    for p, (x, y) in enumerate(zip(prog.download(), want_prog.visible())):
        assert np.array_equal(x, y), f"VideoMixer program, plane {p}"


@pytest.mark.parametrize("geom", [((128, 6), (2, 220)), ((6, 128), (220, 2)), ((1920, 2), (64, 64))])
def test_scaled_size_without_rows_or_columns_is_the_blank_letterbox(geom):
    """A picture so thin that its aligned scaled size is zero rows or columns: the reference would panic (sws_getContext returns NULL for a
    zero dimension, scale.rs:22-33); build-specified: the blank output frame, from every entry point."""
    (iw, ih), (ow, oh) = geom
    g = video.scale_geometry(iw, ih, ow, oh)
    assert g == ov.scaler_geometry(iw, ih, ow, oh) and (g[0] == 0 or g[1] == 0)
    src = ov.HostFrame(iw, ih).fill(3, seed=1)
    want = ov.HostFrame(ow, oh); ov.dynamic_scale(src, want)
    assert not want.visible()[0].any() and (want.visible()[1] == 0x80).all()
    d = upload(src)
    out = video.DFrame(ow, oh); video.scale(d, out)
    assert_frame_equal(out, want, f"scale {geom}")
    assert_frame_equal(video.Scaler(ow, oh).scale(d), want, f"persistent scaler {geom}")
    gm, om = video.VideoMixer(a=0, b=1, fader=0.5), ov.OracleVideoMixer(a=0, b=1, fader=0.5)
    big = ov.HostFrame(ow, oh).fill(9, seed=2)
    prog, _, _ = gm.run_tick(0, [(upload(big), (1, 30), (0, 1)), (d, (1, 30), (0, 1)), None, None])
    assert_frame_equal(prog, om.run_tick(0, [(big, (1, 30), (0, 1)), (src, (1, 30), (0, 1)), None, None]), f"mixer {geom}")


def test_frame_formats_are_validated():
    with pytest.raises(abi.MxError):
        video.DFrame(33, 32, fmt=video.PIXFMT_YUV422P)      # 4:2:2 needs an even width
    video.DFrame(33, 17, fmt=video.PIXFMT_YUV444P)          # 4:4:4 takes any size
    f444 = video.DFrame(64, 64, fmt=video.PIXFMT_YUV444P)
    with pytest.raises(abi.MxError):
        video.crossfade(video.DFrame(64, 64), f444, None, 0.5)   # the cross-fade works on VideoMixer pictures: yuv420p
    with pytest.raises(abi.MxError):
        video.scale(video.DFrame(64, 64), f444)                  # the scaler's output picture is yuv420p


def test_persistent_scaler_is_the_monitor_rescale_and_follows_input_changes():
    # Monitor / StreamOutput shrink the program frame every tick with one DynamicScaler (encode.rs:287-295,338-397)
    sc = video.Scaler(560, 350)
    for k, (iw, ih) in enumerate([(1920, 1080), (1920, 1080), (1280, 720), (560, 350)]):
        src = ov.HostFrame(iw, ih).fill(k, seed=9)
        want = ov.HostFrame(560, 350); ov.dynamic_scale(src, want)
        d = upload(src)
        out = sc.scale(d)
        assert_frame_equal(out, want, f"scaler call {k} {iw}x{ih}")
        if (iw, ih) == (560, 350):
            assert out.device_planes()[0] == d.device_planes()[0]      # equal settings: the frame itself (encode.rs:342-345)


def test_scale_geometry_examples():
    # 16:9 into 16:9 fills; 4:3 into 16:9 pillarboxes on an even offset (encode.rs:354-374)
    assert video.scale_geometry(1280, 720, 1920, 1080) == (1920, 1080, 0, 0)
    assert video.scale_geometry(640, 480, 1920, 1080) == (1440, 1080, 240, 0)
    assert video.scale_geometry(1000, 300, 640, 640) == (640, 192, 0, 224)


@pytest.mark.parametrize("matrix", [None, [4096, 0, 0, 0, 0, 4096, 0, 0, 0, 0, 4096, 0],
                                     [3000, 800, 296, 40960, -200, 4500, -204, 0, 100, -300, 4296, -8192],
                                     # the f32 form of the matrix (exact while the row sums stay below 2^24 in units of 2^-13): negative sums, sums past 255,
                                     # a row at the edge of the bound; then rows beyond it (24-bit integer products) and beyond 24 bits (32-bit products)
                                     [-4096, 0, 0, 1044480, 0, -4096, 0, 1044480, 0, 0, -4096, 1044480],
                                     [8192, 8192, 8192, -1000000, -3000, -3000, -3000, 4095, 1, 1, 1, 2047],
                                     [10922, 10922, 10922, 30000, 4096, 0, 0, -2048, 0, 0, 4097, -2049],
                                     [20000, 20000, 20000, -7000000, 0, 4096, 0, 0, -20000, 0, 0, 2000000],
                                     [8400000, 0, 0, 0, 0, 4096, 0, 0, 0, 0, -8400000, 0]])
@pytest.mark.parametrize("size", [(1920, 1080), (66, 34), (130, 70)])
def test_yuv_to_rgba_bit_exact_vs_build_spec(size, matrix):
    w, h = size
    hf = ov.HostFrame(w, h).fill(6, seed=9)
    want = ov.to_rgba(hf, matrix)
    got = video.to_rgba(upload(hf), matrix)
    assert np.array_equal(got, want)


def test_video_mixer_state_machine_matches_oracle():
    """Frames of different sizes arriving at different rates, expiry by exact rationals, parameter changes."""
    SPT = 735
    gm, om = video.VideoMixer(a=0, b=1, fader=0.75), ov.OracleVideoMixer(a=0, b=1, fader=0.75)
    big = [ov.HostFrame(640, 360).fill(10 + k, seed=k) for k in range(4)]
    small = [ov.HostFrame(320, 240).fill(20 + k, seed=k) for k in range(4)]
    odd = ov.HostFrame(322, 182).fill(33)
    keep = []
    program_seen = 0
    for tick in range(40):
        t = tick * SPT
        ins_h = [None] * 4
        if tick >= 2 and tick % 2 == 0:            # channel 0: 30 fps, 640x360
            ins_h[0] = (big[(tick // 2) % 4], (1, 30), (0, 1))
        if 5 <= tick < 30 and tick % 3 == 2:       # channel 1: 20 fps, 320x240 (forces scale + pillarbox), offset inside the tick
            ins_h[1] = (small[(tick // 3) % 4], (1, 20), (1, 240))
        if tick == 12:                             # channel 2 briefly delivers an odd-ish size: re-target of every scaler
            ins_h[2] = (odd, (1, 10), (0, 1))
        if tick == 20:
            gm.update(a=1, b=0, fader=0.3); om.update(a=1, b=0, fader=0.3)
        if tick == 33:
            gm.update(a=2, b=None, fader=1.0); om.update(a=2, b=None, fader=1.0)
        ins_d = []
        for e in ins_h:
            if e is None:
                ins_d.append(None)
            else:
                d = upload(e[0]); keep.append(d)
                ins_d.append((d, e[1], e[2]))
        prog, fa, fb = gm.run_tick(t, ins_d)
        want = om.run_tick(t, ins_h)
        assert (prog is None) == (want is None), f"tick {tick}: program presence"
        if want is not None:
            program_seen += 1
            assert_frame_equal(prog, want, f"tick {tick}")
        # A / B pass-through = the raw input frame of that channel this tick (video_mixer.rs:80-90)
        pa = gm_params_a = (0 if tick < 20 else (1 if tick < 33 else 2))
        pb = (1 if tick < 20 else (0 if tick < 33 else None))
        assert (fa is None) == (ins_h[pa] is None)
        if fa is not None:
            assert_frame_equal(fa, ins_h[pa][0], f"tick {tick} A pass-through")
        assert (fb is None) == (pb is None or ins_h[pb] is None)
    assert program_seen >= 30
    # first two ticks: nothing to show yet => None (video_mixer.rs:113-119)


def test_video_mixer_frame_expiry_is_exact():
    SPT = 735
    gm, om = video.VideoMixer(a=0, b=None, fader=1.0), ov.OracleVideoMixer(a=0, b=None, fader=1.0)
    hf = ov.HostFrame(64, 64).fill(1)
    d = upload(hf)
    # duration 1/30 s = exactly 2 ticks: visible on ticks 0 and 1, expired at tick 2 (t/44100 >= 2/60)
    seen = []
    for tick in range(4):
        ins_d = [(d, (1, 30), (0, 1))] if tick == 0 else []
        ins_h = [(hf, (1, 30), (0, 1))] if tick == 0 else []
        prog, _, _ = gm.run_tick(tick * SPT, ins_d + [None] * (4 - len(ins_d)))
        want = om.run_tick(tick * SPT, ins_h + [None] * (4 - len(ins_h)))
        seen.append(prog is not None)
        assert (prog is None) == (want is None)
    assert seen == [True, True, False, False]


def test_video_mixer_module_compat_path_host_frames():
    """mx_module_* with MX_KIND_VIDEO_MIXER: the ModuleT::run_tick surface on host frames (InputRef::Video / OutputRef::Video)."""
    import ctypes as C
    from mixlab_amd import abi
    from mixlab_amd.video import VideoMixerParams

    def host_frame(hf, dur=(1, 30), off=(0, 1)):
        f = abi.Frame()
        f.width, f.height = hf.w, hf.h
        for p in range(3):
            f.data[p] = hf.planes[p].ctypes.data
            f.stride[p] = hf.planes[p].shape[1]
        f.dur_num, f.dur_den, f.off_num, f.off_den = dur[0], dur[1], off[0], off[1]
        return f

    p = VideoMixerParams(0, 1, 0.4)
    h = C.c_void_p()
    abi.check(abi.lib.mx_module_create(abi.KIND_VIDEO_MIXER, C.byref(p), C.sizeof(p), C.byref(h)))
    a, b = ov.HostFrame(320, 180).fill(1), ov.HostFrame(160, 120).fill(2)
    om = ov.OracleVideoMixer(a=0, b=1, fader=0.4)
    fa, fb = host_frame(a), host_frame(b)
    ins = (abi.Input * 4)(abi.Input(abi.MX_VIDEO, None, 0, C.cast(C.pointer(fa), C.c_void_p)),
                          abi.Input(abi.MX_VIDEO, None, 0, C.cast(C.pointer(fb), C.c_void_p)),
                          abi.Input(abi.MX_DISCONNECTED, None, 0, None), abi.Input(abi.MX_VIDEO, None, 0, None))
    outs_h = [ov.HostFrame(640, 360) for _ in range(3)]
    outs_f = [host_frame(o) for o in outs_h]
    outs = (abi.Output * 3)(*[abi.Output(abi.MX_VIDEO, None, 0, C.cast(C.pointer(f), C.c_void_p), 0) for f in outs_f])
    n = C.c_size_t()
    abi.check(abi.lib.mx_module_run_tick(h, 0, ins, 4, outs, 3, None, C.byref(n)))
    want = om.run_tick(0, [(a, (1, 30), (0, 1)), (b, (1, 30), (0, 1)), None, None])
    assert [o.video_present for o in outs] == [1, 1, 1]
    assert (outs_f[0].width, outs_f[0].height) == (want.w, want.h) == (320, 180)
    for p in range(3):
        hh, ww = want.h >> (1 if p else 0), want.w >> (1 if p else 0)
        assert np.array_equal(outs_h[0].planes[p][:hh, :ww], want.visible()[p])
        assert np.array_equal(outs_h[1].planes[p][: a.h >> (1 if p else 0), : a.w >> (1 if p else 0)], a.visible()[p])   # A pass-through
    assert (outs_f[2].width, outs_f[2].height) == (160, 120) and (outs_f[0].dur_num, outs_f[0].dur_den) == (1, 60)
    abi.lib.mx_module_destroy(h)


@pytest.mark.parametrize("seed", list(range(12)))
def test_video_mixer_random_scenarios_match_oracle(seed):
    """Seeded random arrivals on all four channels (sizes, durations, tick offsets), random A / B / fader changes, gaps long
    enough for stored frames to expire: program frame presence and pixels, tick by tick, against the oracle state machine."""
    rng = np.random.default_rng(seed)
    SPT = 735
    sizes = [(64, 64), (96, 54), (128, 72), (66, 34), (160, 120), (130, 70)]
    pool = [ov.HostFrame(w, h).fill(int(rng.integers(0, 50)), seed=int(rng.integers(0, 99))) for (w, h) in sizes for _ in range(2)]
    a, b, fader = 0, 1, float(rng.uniform(0, 1))
    gm, om = video.VideoMixer(a=a, b=b, fader=fader), ov.OracleVideoMixer(a=a, b=b, fader=fader)
    keep, shown = [], 0
    quiet_until = -1
    for tick in range(36):
        if rng.random() < 0.06:
            quiet_until = tick + int(rng.integers(3, 9))      # nobody delivers for a while: stored frames run out
        ins_h = [None] * 4
        for ch in range(4):
            if tick > quiet_until and rng.random() < (0.5, 0.3, 0.15, 0.05)[ch]:
                dur = [(1, 60), (1, 30), (1, 20), (1, 10), (2, 25), (1, 7)][int(rng.integers(0, 6))]
                off = [(0, 1), (1, 240), (1, 120), (1, 61)][int(rng.integers(0, 4))]
                ins_h[ch] = (pool[int(rng.integers(0, len(pool)))], dur, off)
        if rng.random() < 0.15:
            a = [None, 0, 1, 2, 3][int(rng.integers(0, 5))]; b = [None, 0, 1, 2, 3][int(rng.integers(0, 5))]
            fader = float([0.0, 1.0, rng.uniform(0, 1), 1.5, -0.1][int(rng.integers(0, 5))])
            gm.update(a=a, b=b, fader=fader); om.update(a=a, b=b, fader=fader)
        ins_d = []
        for e in ins_h:
            if e is None:
                ins_d.append(None)
            else:
                d = upload(e[0]); keep.append(d)
                ins_d.append((d, e[1], e[2]))
        prog, fa, fb = gm.run_tick(tick * SPT, ins_d)
        want = om.run_tick(tick * SPT, ins_h)
        assert (prog is None) == (want is None), f"seed {seed} tick {tick}: program presence"
        if want is not None:
            shown += 1
            assert (prog.width, prog.height) == (want.w, want.h), f"seed {seed} tick {tick}: target size"
            assert_frame_equal(prog, want, f"seed {seed} tick {tick}")
        assert (fa is None) == (a is None or ins_h[a] is None)
        assert (fb is None) == (b is None or ins_h[b] is None)
    assert shown > 5


@pytest.mark.parametrize("fmt", [video.PIXFMT_RGB24, video.PIXFMT_BGRA, video.PIXFMT_BGR24, video.PIXFMT_RGBA, video.PIXFMT_ARGB, video.PIXFMT_ABGR], ids=["rgb24", "bgra", "bgr24", "rgba", "argb", "abgr"])
@pytest.mark.parametrize("src,dst", [((320, 180), (480, 270)), ((322, 182), (320, 180)), ((64, 64), (320, 180)), ((1280, 720), (560, 350)), ((320, 180), (320, 180))],
                         ids=["up-1.5x", "down-slightly", "pillarbox", "monitor-downscale", "same-size"])
def test_packed_rgb_scaler_inputs_equal_the_oracle_composition(fmt, src, dst):
    """A packed RGB scaler input (codec/src/ffmpeg/scale.rs:16-39 takes any input format) is BUILD-SPECIFIED as the yuv444p frame of its
    per-pixel BT.709 conversion, resampled like any 4:4:4 input into the yuv420p output -- also when the sizes are equal (the picture
    settings differ by the format, encode.rs:342-352).  Stateless call, persistent scaler and a VideoMixer input."""
    rng = np.random.default_rng(src[0] * 7 + dst[0] + fmt)
    w, h = src
    bpp = 3 if fmt in (video.PIXFMT_RGB24, video.PIXFMT_BGR24) else 4
    pix = rng.integers(0, 256, size=(h, w, bpp), dtype=np.uint8)
    pix[: h // 3] = (np.add.outer(np.arange(h // 3), np.arange(w))[..., None] * np.array([1, 2, 3, 1][:bpp])).astype(np.uint8)   # a smooth part too
    d = video.DFrame(w, h, fmt=fmt).upload_packed(pix)
    assert np.array_equal(d.download()[0], pix)
    as444 = ov.packed_rgb_to_yuv444(pix, fmt)
    want = ov.HostFrame(*dst); ov.blank(want); ov.dynamic_scale(as444, want)
    out = video.DFrame(*dst)
    video.scale(d, out)
    for p, (x, y) in enumerate(zip(out.download(), want.visible())):
        assert np.array_equal(x, y), f"stateless scale, plane {p}"
    sc = video.Scaler(*dst)
    for _ in range(2):
        res = sc.scale(d)
        for p, (x, y) in enumerate(zip(res.download(), want.visible())):
            assert np.array_equal(x, y), f"persistent scaler, plane {p}"
    # as a VideoMixer input beside a yuv420p layer of the output's size
    other = ov.HostFrame(*dst).fill(3, seed=5)
    m = video.VideoMixer(a=0, b=1, fader=0.4)
    om = ov.OracleVideoMixer(a=0, b=1, fader=0.4)
    prog, _a, _b = m.run_tick(0, [(d, (1, 30), (0, 1)), (video.DFrame(*dst).upload(*other.visible()), (1, 30), (0, 1)), None, None])
    want_prog = om.run_tick(0, [(as444, (1, 30), (0, 1)), (other, (1, 30), (0, 1)), None, None])
    for p, (x, y) in enumerate(zip(prog.download(), want_prog.visible())):
        assert np.array_equal(x, y), f"VideoMixer program, plane {p}"


def test_one_scaler_fed_rgb24_then_gray8_of_the_same_size_keeps_gray_chroma_neutral():
    """A persistent scaler converts packed RGB and gray8 inputs into pooled yuv444p frames of the input's size.  A gray8 picture that follows an RGB
    one of the same size reuses the RGB picture's pool frame: its chroma planes must be re-blanked to 0x80 (gray8 IS the frame with U = V = 0x80),
    not keep the colour of the picture before."""
    rng = np.random.default_rng(77)
    (w, h), dst = (320, 180), (480, 270)
    pix = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    d_rgb = video.DFrame(w, h, fmt=video.PIXFMT_RGB24).upload_packed(pix)
    d_gray = video.DFrame(w, h, fmt=video.PIXFMT_GRAY8).upload_packed(gray)
    as444 = ov.HostFrame(w, h, 2)
    as444.planes[0][:, :w] = gray; as444.planes[1][:, :w] = 0x80; as444.planes[2][:, :w] = 0x80
    want_gray = ov.HostFrame(*dst); ov.blank(want_gray); ov.dynamic_scale(as444, want_gray)
    want_rgb = ov.HostFrame(*dst); ov.blank(want_rgb); ov.dynamic_scale(ov.packed_rgb_to_yuv444(pix, video.PIXFMT_RGB24), want_rgb)
    sc = video.Scaler(*dst)
    for rnd in range(3):
        res = sc.scale(d_rgb)
        for p, (x, y) in enumerate(zip(res.download(), want_rgb.visible())):
            assert np.array_equal(x, y), f"round {rnd}: rgb24, plane {p}"
        del res
        res = sc.scale(d_gray)
        got = res.download()
        for p, (x, y) in enumerate(zip(got, want_gray.visible())):
            assert np.array_equal(x, y), f"round {rnd}: gray8 after rgb24, plane {p}"
        assert (got[1] == 0x80).all() and (got[2] == 0x80).all()
        del res
