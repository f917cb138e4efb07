"""Device time of one DynamicScaler call per geometry class: the tiled 4-tap kernel (upscales, 1:1), the widened two-pass kernels (downscales).  usage: python tools/scale_probe.py"""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
from mixlab_amd import video
import oracle_video as ov

for (iw, ih), (ow, oh) in [((1280, 720), (1920, 1080)), ((1920, 1080), (1920, 1080)), ((3840, 2160), (1920, 1080)), ((1920, 1080), (1280, 720)), ((1920, 1080), (560, 350)), ((2560, 1440), (1920, 1080)), ((640, 480), (1920, 1080))]:
    for fmt in (video.PIXFMT_YUV420P, video.PIXFMT_NV12):
        src = ov.HostFrame(iw, ih, fmt).fill(1, seed=1)
        d = video.DFrame(iw, ih, fmt=fmt).upload(*src.visible())
        sc = video.Scaler(ow, oh)
        for _ in range(3):
            sc.scale(d)
        torch.cuda.synchronize()
        n = 40
        t0 = time.perf_counter()
        for _ in range(n):
            sc.scale(d)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{iw}x{ih} fmt {fmt} -> {ow}x{oh}: {dt * 1e6:7.1f} us per call  ({(iw * ih + ow * oh) * 1.5 / dt / 1e9:6.1f} GB/s of pixels in + out)", flush=True)
