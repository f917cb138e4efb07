"""Boundary hygiene: every entry point of include/mixlab_gpu.h called with all-zero arguments (NULL handles, NULL buffers, zero sizes) and
with small garbage integers must return a status (or do nothing, for the void destroy calls) -- never crash the process.  The argument
lists come from the header text itself.  Usage: python tools/fuzz_abi.py   (needs the GPU box: some calls reach the runtime)"""
import ctypes as C, re, sys, pathlib, faulthandler
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
faulthandler.enable()
from mixlab_amd import abi
lib = abi.lib

text = (ROOT / "include" / "mixlab_gpu.h").read_text()
text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
protos = re.findall(r"\b(int|void|uint32_t|const char\s*\*)\s+(mx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text)
print(f"{len(protos)} prototypes")


def zero_for(decl):
    d = decl.strip()
    if d in ("void", ""):
        return None
    if "*" in d or "[" in d:
        return C.c_void_p(0)
    if "double" in d:
        return C.c_double(0.0)
    if "64" in d or "size_t" in d:
        return C.c_uint64(0)
    return C.c_uint32(0)


def garbage_for(decl, k):
    d = decl.strip()
    if "*" in d or "[" in d:
        return C.c_void_p(0)          # only NULL pointers: a wild non-NULL pointer is the caller's bug, not a hygiene question
    if "double" in d:
        return C.c_double([1e308, -1.0, float("nan")][k % 3])
    if "64" in d or "size_t" in d:
        return C.c_uint64([1, 0xFFFFFFFF, 1 << 40][k % 3])
    return C.c_uint32([1, 0xFFFFFFFF, 12345][k % 3])


n = 0
for ret, name, args in protos:
    fn = getattr(lib, name)
    fn.restype = None if ret == "void" else (C.c_char_p if "char" in ret else C.c_int)
    decls = [a for a in re.split(r",(?![^\[]*\])", args)] if args.strip() not in ("void", "") else []
    fn.argtypes = None
    for variant in range(4):
        vals = [zero_for(d) if variant == 0 else garbage_for(d, variant + i) for i, d in enumerate(decls)]
        vals = [v for v in vals if v is not None]
        r = fn(*vals)
        n += 1
print(f"{n} calls returned; last error: {(lib.mx_last_error() or b'').decode()[:100]}")

# ---- second pass: a VALID first handle, everything else NULL / zero / garbage scalars ----
from mixlab_amd import video, ingest
from mixlab_amd.workspace import Workspace
ws = Workspace(44100, 60)
s = ws.source_mono(); e = ws.eq_three(1.0, 2.0, 3.0); ws.connect(s, 0, e, 0)
sv = ws.source_video(); vm = ws.video_mixer(a=0, b=None, fader=0.5); ws.connect(sv, 0, vm, 0); mon = ws.monitor(64, 48); ws.connect(vm, 0, mon, 0); ws.connect(e, 0, mon, 1) if False else None
g = ws.build(max_ticks_per_run=2)
g.run_ticks(0, 1)
objs = {
    "mx_graph": g, "mx_dframe": video.DFrame(32, 32), "mx_video_mixer": video.VideoMixer(a=0, b=1, fader=0.5),
    "mx_video_scaler": video.Scaler(64, 64), "mx_media_source": ingest.MediaSource(), "mx_stream_input": ingest.StreamInput(),
    "mx_frame_stager": ingest.FrameStager(2), "mx_pcm_ring": abi.PcmRing(), "mx_module": abi.Module(abi.KIND_AMPLIFIER, abi.AmplifierParams(1.0, 0.0)),
}
handles = {k: v._h for k, v in objs.items()}
n2 = 0
for ret, name, args in protos:
    if name.endswith("_destroy") or name.endswith("_release") or name in ("mx_device_free", "mx_host_free"):
        continue
    decls = [a for a in re.split(r",(?![^\[]*\])", args)] if args.strip() not in ("void", "") else []
    if not decls:
        continue
    m = re.match(r"\s*(?:const\s+)?(mx_[a-z_]+)\s*\*\s*\w+\s*$", decls[0])
    if not m or m.group(1) not in handles:
        continue
    fn = getattr(lib, name)
    fn.argtypes = None
    fn.restype = None if ret == "void" else (C.c_char_p if "char" in ret else C.c_int)
    for variant in range(4):
        vals = [handles[m.group(1)]] + [zero_for(d) if variant == 0 else garbage_for(d, variant + i) for i, d in enumerate(decls[1:])]
        if "-v" in sys.argv:
            print(name, variant, [getattr(v, "value", v) for v in vals[1:]], flush=True)
        r = fn(*vals)
        n2 += 1
print(f"{n2} calls with a valid handle returned; last error: {(lib.mx_last_error() or b'').decode()[:100]}")
g.run_ticks(1, 1); g.sync()
print("the graph still runs")
