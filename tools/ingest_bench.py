"""Host -> device hand-over of decoded 1080p yuv420p frames: the staging ring (one page-locked copy per frame, asynchronous) beside
mx_dframe_upload (three pageable 2-D copies, synchronous).  Prints frames/s and GB/s; PCIe-inclusive by construction."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
from mixlab_amd import ingest, video

W, H, N = 1920, 1080, 300
rng = np.random.default_rng(0)
planes = [rng.integers(0, 256, (H, W), dtype=np.uint8), rng.integers(0, 256, (H // 2, W // 2), dtype=np.uint8), rng.integers(0, 256, (H // 2, W // 2), dtype=np.uint8)]
frame_bytes = W * H * 3 // 2

d = video.DFrame(W, H)
d.upload(*planes)
t0 = time.perf_counter()
for _ in range(N):
    d.upload(*planes)
dt = time.perf_counter() - t0
print(f"mx_dframe_upload      : {N / dt:8.0f} frames/s  {N * frame_bytes / dt / 1e9:6.2f} GB/s  ({dt / N * 1e6:.0f} us/frame, synchronous)")

for slots in (2, 4, 8):
    st = ingest.FrameStager(slots=slots)
    st.fence(None)
    keep = [st.upload(planes, W, H) for _ in range(slots)]
    st.sync()
    keep = []
    t0 = time.perf_counter()
    for _ in range(N):
        f = st.upload(planes, W, H)
        f.release()
    st.sync()
    dt = time.perf_counter() - t0
    print(f"frame stager, {slots} slots: {N / dt:8.0f} frames/s  {N * frame_bytes / dt / 1e9:6.2f} GB/s  ({dt / N * 1e6:.0f} us/frame incl. the host row packing)")

import ctypes as C
from mixlab_amd import abi
st = ingest.FrameStager(slots=4)
st.fence(None)
hf, ticket, h = abi.Frame(), C.c_uint32(), C.c_void_p()
t0 = time.perf_counter()
for _ in range(N):      # the two C calls a decoder integration makes per picture; it would write its picture into hf.data in between
    abi.check(abi.lib.mx_frame_stager_acquire(st._h, W, H, 0, C.byref(hf), C.byref(ticket)))
    abi.check(abi.lib.mx_frame_stager_commit(st._h, ticket, C.byref(h)))
    abi.lib.mx_dframe_release(h)
st.sync()
dt = time.perf_counter() - t0
print(f"acquire / commit      : {N / dt:8.0f} frames/s  {N * frame_bytes / dt / 1e9:6.2f} GB/s  ({dt / N * 1e6:.0f} us/frame, decoder writes into the slot: no host copy)")
