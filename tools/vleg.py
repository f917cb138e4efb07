#!/usr/bin/env python
"""The video leg of bench.py alone (config 4: 8 layers -> 7 cross-fades -> RGBA), for kernel iteration and profiler passes.
usage: python tools/vleg.py [frames] [repeats] [main | alpha | no_rest_fader]   (MX_VIDEO_MFMA_MATRIX=1 in the environment: the matrix on the matrix cores)"""
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import bench  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = sys.argv[3] if len(sys.argv) > 3 else "main"
torch.cuda.set_device(0)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for _ in range(reps):
        v = bench.video_leg(torch, None, 1, stream, 0, frames, 3, only=only)
        print(json.dumps({k: v[k] for k in ("value", "device_us_per_frame", "hbm_frac_moved_bytes_device")}), flush=True)
