#!/usr/bin/env python
"""The FIR + resampler leg of bench.py alone (config 3), for profiler passes.  usage: python tools/fleg.py [ticks] [steps]"""
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import bench  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.cuda.set_device(0)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    r = bench.fir_leg(torch, stream, 0, T, steps, 2)
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "kernel_ms_per_step", "fp_contract") if k in r}), flush=True)
