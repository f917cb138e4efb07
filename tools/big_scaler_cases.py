import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent; sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle_video as ov
from mixlab_amd import video
cases = [((3840, 2160), (1920, 1080), 0), ((7680, 4320), (560, 350), 0), ((16384, 2), (1920, 1080), 0), ((2, 16384), (64, 4096), 0), ((4096, 4096), (4094, 4090), 0),
         ((1920, 1080), (7680, 4320), 0), ((3840, 2160), (3840, 2160), 3), ((5000, 3000), (1234, 2222), 2), ((16384, 16), (16384, 64), 1), ((1280, 720), (16000, 9000), 0)]
for (iw, ih), (ow, oh), fmt in cases:
    src = ov.HostFrame(iw, ih, fmt).fill(3, seed=5)
    want = ov.HostFrame(ow, oh); ov.dynamic_scale(src, want)
    d = video.DFrame(iw, ih, fmt=fmt).upload(*src.visible())
    out = video.DFrame(ow, oh); video.scale(d, out)
    ok = all(np.array_equal(a, b) for a, b in zip(out.download(), want.visible()))
    res = video.Scaler(ow, oh).scale(d)
    ok2 = all(np.array_equal(a, b) for a, b in zip(res.download(), want.visible()))
    print((iw, ih, fmt), "->", (ow, oh), "OK" if ok and ok2 else "MISMATCH", flush=True)
