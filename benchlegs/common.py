"""What every leg shares: the config-2 channel strips, their gate schedule, the byte accounting, and the committed counter passes under profiles/."""
from __future__ import annotations

import json
import os
import pathlib
import sys
import time
from dataclasses import dataclass, field

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests", ROOT / "tools"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
F64_VALU_PEAK_TOPS = 39.3   # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz f64 instructions / s (an FMA counts once)

# algorithmic (module-boundary) bytes per instance per frame: every input port read once + every output port written once;
# mixer is per input channel, +16 / frame for its two outputs.  Used only with --no-fuse, where every port really is materialised.
BYTES_PER_FRAME = {"trigger": 4, "envelope": 8, "eq_three": 8, "stereo_panner": 16, "amplifier": 20, "mixer": 8}
# default (graph-compiler fusion): Trigger + Envelope + EqThree + StereoPanner + Amplifier are ONE kernel that reads the source (4 B / frame) and
# writes the strip as one float per frame (L == R): 8 B / frame = SURVEY 8d's 2M per EqThree channel-tick; the Mixer reads those 4 B.
BYTES_PER_FRAME_FUSED = {"eq_three": 4 + 4, "mixer": 4}

VIDEO_FADERS = [1.0, 0.75, 0.5, 0.5, 0.25, 0.9, 0.1]
VIDEO_MATRIX = [3900, 150, 46, 4096, 60, 3980, 56, -2048, 20, 120, 3956, 0]
VIDEO_SIZES = [(1920, 1080)] * 6 + [(1280, 720)] * 2


def dist_backend():
    """torch.distributed backend of an N > 1 run: RCCL ("nccl").  MX_BENCH_DIST_BACKEND=gloo (tests: N processes sharing ONE GPU, the library's exchange on the RCCL test
    double) carries the same barriers / maxima / gathers on the host."""
    return os.environ.get("MX_BENCH_DIST_BACKEND", "nccl")


def dist_device():
    return "cuda" if dist_backend() == "nccl" else "cpu"


def gate_open(tick, k):
    """the Trigger of strip k toggles every 30 ticks with phase k mod 60."""
    return ((tick + k) // 30) % 2 == 1


def gate_events(abi, trigs, first_strip, t0, n_ticks):
    """The toggles of every strip's Trigger that fall on ticks [t0, t0 + n_ticks) as one mx_param_event array for mx_graph_schedule_params_batch
    (tick_in_run 0 = the boundary before the submission's first tick: a strip whose toggle falls exactly on t0 gets it there).
    Returns (ctypes pointer, count, keep-alive tuple) or None.  Built with numpy: ~70 000 events per 2048-tick step."""
    import ctypes as C
    p_open, p_closed = abi.TriggerParams(1), abi.TriggerParams(0)
    po, pc = C.addressof(p_open), C.addressof(p_closed)
    k = first_strip + np.arange(len(trigs), dtype=np.int64)
    first = (30 - (t0 + k) % 30) % 30                             # first toggle at or after t0, per strip
    n_ev = np.maximum(0, (n_ticks - first + 29) // 30)            # toggles at first, first + 30, ... < n_ticks
    total = int(n_ev.sum())
    if total == 0:
        return None
    strip = np.repeat(np.arange(len(trigs)), n_ev)
    j = np.arange(total) - np.repeat(np.cumsum(n_ev) - n_ev, n_ev)
    tick = first[strip] + 30 * j
    opens = ((t0 + tick + k[strip]) // 30) % 2 == 1
    ev = np.zeros(total, dtype=np.dtype([("node", "<u4"), ("tick_in_run", "<u4"), ("params", "<u8"), ("params_len", "<u8")], align=True))
    assert ev.dtype.itemsize == C.sizeof(abi.ParamEvent)
    ev["node"] = np.asarray(trigs, dtype=np.uint32)[strip]; ev["tick_in_run"] = tick
    ev["params"] = np.where(opens, po, pc); ev["params_len"] = C.sizeof(abi.TriggerParams)
    return ev.ctypes.data_as(C.POINTER(abi.ParamEvent)), total, (ev, p_open, p_closed)


def build_strips(abi, Workspace, synth, n_strips, first_strip, sample_rate, ws=None, total=None, want_trigs=False):
    """Config-2 strips [first_strip, first_strip + n_strips) with the global seeded parameters, into a Mixer(n_strips);
    `ws`: add them to an existing workspace (group buses), `total`: size of the whole job the parameters are drawn for."""
    if total is None:
        total = 1024 if first_strip + n_strips <= 1024 else first_strip + n_strips
    eq_g = synth.uniform(10, 3 * total, -24.0, 6.0)
    mg = synth.uniform(11, total, -24.0, 6.0)
    mf = synth.uniform(12, total, 0.0, 1.0)
    if ws is None:
        ws = Workspace(sample_rate, 60)
    mix = ws.mixer([(float(mg[k]), float(mf[k]), k % 8 == 0) for k in range(first_strip, first_strip + n_strips)])
    srcs, trigs = [], []
    for j, k in enumerate(range(first_strip, first_strip + n_strips)):
        trig = ws.trigger(gate_open(0, k))          # gate at tick 0; toggles every 30 ticks with phase k mod 60 (gate_events)
        trigs.append(trig)
        env = ws.envelope()                         # defaults 25/500/0.8/200 (protocol/src/lib.rs:318-327)
        src = ws.source_mono()
        eq = ws.eq_three(float(eq_g[3 * k]), float(eq_g[3 * k + 1]), float(eq_g[3 * k + 2]))
        pan = ws.stereo_panner()
        amp = ws.amplifier(1.0, 0.5)
        ws.connect(trig, 0, env, 0)
        ws.connect(src, 0, eq, 0)
        ws.connect(eq, 0, pan, 0); ws.connect(eq, 0, pan, 1)
        ws.connect(pan, 0, amp, 0); ws.connect(env, 0, amp, 1)
        ws.connect(amp, 0, mix, j)
        srcs.append(src)
    if want_trigs:
        return ws, mix, srcs, trigs
    return ws, mix, srcs


def video_cascade(ws, faders=VIDEO_FADERS):
    """Config 4 as the reference expresses it: 8 video sources -> a cascade of 7 VideoMixer cross-fades -> the build-specified RGBA node."""
    srcs = [ws.source_video() for _ in VIDEO_SIZES]
    prev = srcs[0]
    for k in range(1, 8):
        m = ws.video_mixer(a=0, b=1, fader=faders[k - 1])
        ws.connect(prev, 0, m, 0); ws.connect(srcs[k], 0, m, 1)
        prev = m
    rgba = ws.video_to_rgba(VIDEO_MATRIX)
    ws.connect(prev, 0, rgba, 0)
    return srcs, rgba


def tiled_noise(synth, seed, T, spt, base_ticks=None):
    """A seeded noise block of min(T, 256) ticks repeated to fill T ticks (host-side generation stays in seconds)."""
    base = min(T, 256) if base_ticks is None else base_ticks
    blk = synth.noise(seed, base * spt)
    return np.tile(blk, (T + base - 1) // base)[: T * spt]


@dataclass
class Job:
    """The headline job's context, handed to every leg that runs beside it."""
    torch: object
    dist: object
    abi: object
    shard: object
    Workspace: object
    synth: object
    args: object
    rank: int
    world: int
    local_rank: int
    stream: object
    use_dist: bool
    T: int
    SR: int
    spt: int
    first: int
    local_strips: int
    toggling: bool
    flags: int
    ws: object = None
    g: object = None
    mix: int = 0
    srcs: list = field(default_factory=list)
    trigs: list = field(default_factory=list)
    nxt: int = 0                      # the next unused step index of the headline graph
    dt: float = 0.0                   # seconds of the timed region (max over ranks)

    def build(self, ws=None, T=None, flags=None, auto_overlap=None):
        """A graph on this job's stream; auto_overlap False = MX_OVERLAP_AUTO=0 (every launch group on one stream)."""
        if auto_overlap is False:
            os.environ["MX_OVERLAP_AUTO"] = "0"
        try:
            return (ws or self.ws).build(max_ticks_per_run=self.T if T is None else T, flags=self.flags if flags is None else flags,
                                         device=self.local_rank, stream=self.stream.cuda_stream)
        finally:
            os.environ.pop("MX_OVERLAP_AUTO", None)

    def bind_resident_sources(self, g2, srcs2=None):
        """Point another graph's sources at the headline graph's resident source buffers (bound, not copied)."""
        for s2, s in zip(srcs2 or self.srcs, self.srcs):
            g2.bind_source_device(s2, self.g.output_device_ptr(s, 0)[0])

    def events(self, i, T=None, trigs=None, first=None):
        if not self.toggling:
            return None
        T = self.T if T is None else T
        return gate_events(self.abi, self.trigs if trigs is None else trigs, self.first if first is None else first, i * T, T)


def run_steps(g, evs, T, i0, n, tick0=0):
    for i in range(i0, i0 + n):
        if evs[i] is not None:
            g.schedule_params_batch(evs[i][0], evs[i][1])
        g.run_ticks(tick0 + i * T, T)


def timed_steps(g, evs, T, warm, n, profile=False, tick0=0):
    """`warm` untimed steps, then n timed ones bracketed by mx_graph_sync; returns (seconds, kernel ms per step by kind)."""
    run_steps(g, evs, T, 0, warm, tick0)
    g.sync()
    g.profile_enable(profile)
    t0 = time.perf_counter()
    run_steps(g, evs, T, warm, n, tick0)
    g.sync()
    dt = time.perf_counter() - t0
    g.profile_enable(False)
    by_kind, _tot, n_prof = g.profile_collect()
    return dt, {k: v / max(1, n_prof) for k, v in sorted(by_kind.items()) if v > 0}


def rounded(d, nd=5):
    return {k: round(v, nd) for k, v in d.items()}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ---- the committed counter passes (profiles/rNN): counters cannot be read from inside the process -------------------------------------------

def _profile_dir():
    """the newest profiles/rNN that holds counter summaries (tools/profile_round.sh copies them there before it runs the default command)"""
    ds = sorted(d for d in (ROOT / "profiles").glob("r[0-9][0-9]") if (d / "pmc_traffic.json").exists())
    return ds[-1] if ds else ROOT / "profiles" / "r06"


PROFILE_DIR = _profile_dir()
PROFILE_TAG = f"profiles/{PROFILE_DIR.name}"


def kernel_hash(family):
    from kernel_hash import kernel_hash as kh
    return kh(family)


def _load(name, family="audio"):
    try:
        rec = json.load(open(PROFILE_DIR / name))
    except (OSError, ValueError):
        return None
    return rec if rec.get("kernel_sources_sha16") == kernel_hash(family) else None


def pmc_traffic(kernel, args, world, toggling, fc=None):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes (pmc_traffic.json of the newest round: `--pmc FETCH_SIZE` /
    `--pmc WRITE_SIZE`, separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide streaming reads); None when the run's
    configuration differs from the profiled one OR the kernel sources have changed since (their hash is in the JSON): a stale figure is worse than none."""
    fc = bool(args.fp_contract) if fc is None else fc
    name = "pmc_traffic_fc.json" if fc else "pmc_traffic.json"
    rec = _load(name)
    if rec is None:
        return None, None
    c = rec.get("config", {})
    same = (c.get("strips") == args.strips and c.get("ticks_per_step") == args.ticks_per_step and c.get("sample_rate") == args.sample_rate
            and c.get("fused") == (not args.no_fuse) and c.get("eq_fast") == bool(args.eq_fast) and c.get("n_gpus") == world
            and c.get("gates_toggle") == bool(toggling) and bool(c.get("fp_contract", False)) == fc)
    if not same or kernel not in rec.get("bytes_per_launch", {}):
        return None, None
    return rec["bytes_per_launch"][kernel], f"{PROFILE_TAG}/{name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; kernel sources unchanged since)"


def sq_profile(kernel_substr, fc, samples):
    """The committed SQ counter pass about the dominant kernel, per OUTPUT sample of the launch: VALU wave-instructions x 64 lanes / samples."""
    rec = _load("pmc_sq_fc.json" if fc else "pmc_sq_toggle.json")
    for k, v in (rec or {}).get("mean_per_dispatch", {}).items():
        if kernel_substr in k and v.get("SQ_INSTS_VALU", 0) > 1e6:
            return {"kernel": k[-70:], "valu_instructions_per_output_sample": round(v["SQ_INSTS_VALU"] * 64.0 / samples, 2)}
    return None


def sustained_clock_ghz(kernel_substr, fc=False):
    """The clock the chip held under a kernel in the committed counter pass (clock.json; kernels shorter than 100 us are not listed there)."""
    rec = _load("clock_fc.json" if fc else "clock.json")
    for k, v in (rec or {}).get("ghz_by_kernel", {}).items():
        if kernel_substr in k:
            return v["ghz"]
    return None
