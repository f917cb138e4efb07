"""Geometry stress of the DynamicScaler (geometry + every resampling kernel form: tiled 4-tap, widened two-pass, batched, nv12 strided):
random input / output sizes and input pixel formats against the oracle's build-specified bicubic, stateless and through the persistent
scaler with its context re-targeted from call to call.  Usage: python tools/stress_scaler.py [first_seed] [count]"""
import sys, pathlib, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle_video as ov
from mixlab_amd import video

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200


def size(rng, fmt, big):
    hi = 2000 if big else 400
    w, h = int(rng.integers(1, hi // 2 + 1)) * 2, int(rng.integers(1, hi // 2 + 1)) * 2      # even: every format takes it
    if rng.random() < 0.15:
        w = int(rng.choice([2, 4, 6, 64, 128, 130, 1920, 1280]))
    if rng.random() < 0.15:
        h = int(rng.choice([2, 4, 6, 64, 72, 1080, 720]))
    if fmt in (6, 7, 8):                      # yuv410p / yuv411p / yuv440p: multiples of the quarter subsampling
        w, h = max(4, w & ~3), max(4, h & ~3)
    return w, h


def check(dev, want, what):
    assert (dev.width, dev.height) == (want.w, want.h), what
    for p, (g, w) in enumerate(zip(dev.download(), want.visible())):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{what}: plane {p}: {len(bad)} pixels differ, first {bad[:3].tolist()}"


bad = 0
scalers = {}
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    fmt = int(rng.choice([0, 0, 1, 2, 3, 6, 7, 8] + list(range(10, 23))))
    big = rng.random() < 0.15
    (iw, ih), (ow, oh) = size(rng, fmt, big), size(rng, 0, big)
    what = f"seed {seed}: {iw}x{ih} fmt {fmt} -> {ow}x{oh}"
    if "-v" in sys.argv:
        print(what, flush=True)
    try:
        assert video.scale_geometry(iw, ih, ow, oh) == ov.scaler_geometry(iw, ih, ow, oh), what + " geometry"
        if fmt in (21, 22):      # packed 4:2:2: random bytes; the yuv422p frame with the same samples goes through the oracle's scaler
            pix = rng.integers(0, 256, size=(ih, 2 * iw), dtype=np.uint8)
            src = ov.yuyv_to_422p(pix, fmt)
            dsrc = video.DFrame(iw, ih, fmt=fmt).upload_packed(pix)
        elif fmt >= 10:      # words deeper than 8 bits (random samples, junk in the ignored bits): the 8-bit frame they stand for goes through the oracle's scaler
            lay, bits, shift = video.DEEP[fmt]
            cw, ch = (0 if lay == 2 else 1), (1 if lay == 0 else 0)
            def plane(ph, pw):
                v = rng.integers(0, 1 << bits, size=(ph, pw), dtype=np.uint32)
                junk = rng.integers(0, 1 << (16 - bits), size=(ph, pw), dtype=np.uint32) if bits < 16 else 0
                return (((v << shift) | junk) if shift else (v | (junk << bits))).astype(np.uint16)
            planes = [plane(ih, iw), plane(ih >> ch, iw >> cw), plane(ih >> ch, iw >> cw)]
            if fmt in (13, 20):
                uv = np.empty((ih >> 1, iw), np.uint16); uv[:, 0::2] = planes[1]; uv[:, 1::2] = planes[2]; planes = [planes[0], uv]
            src = ov.deep_to_8(planes, iw, ih, fmt)
            dsrc = video.DFrame(iw, ih, fmt=fmt).upload(*planes)
        else:
            src = ov.HostFrame(iw, ih, fmt).fill(seed % 50, seed=seed)
            dsrc = video.DFrame(iw, ih, fmt=fmt).upload(*src.visible())
        want = ov.HostFrame(ow, oh); ov.dynamic_scale(src, want)
        out = video.DFrame(ow, oh)
        video.scale(dsrc, out)
        check(out, want, what)
        key = (ow, oh) if rng.random() < 0.7 else (int(rng.choice([64, 560])), int(rng.choice([64, 350])))   # persistent scalers see a stream of differing inputs
        if key not in scalers:
            scalers[key] = video.Scaler(*key)
        w2 = ov.HostFrame(*key); ov.dynamic_scale(src, w2)
        res = scalers[key].scale(dsrc)
        check(res, w2 if (iw, ih, src.fmt) != (key[0], key[1], 0) else src, what + f" persistent {key}")
        if len(scalers) > 40:
            scalers.clear()
    except Exception:
        bad += 1; traceback.print_exc(limit=3)
        if bad >= 3:
            break
print(f"{count} geometries, {bad} failures")
sys.exit(1 if bad else 0)
