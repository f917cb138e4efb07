"""mx_exchange_* on the GPU over a single-rank RCCL communicator (what one rank of the 8-GPU job runs, collectives included:
ncclCommInitRank, ncclAllGather, grouped ncclSend / ncclRecv, ncclAllReduce -- all called by libmixlab_gpu.so itself, which binds
librccl on first use): the combined bus of every exchange mode must be the local Mixer's output bit for bit (Mixer(1, unity) of one partial is
that partial), across pipelined steps that reuse the two slots.  world > 1 runs through the same code on the loopback transport:
tests/test_gpu_config5_sharded.py; the 2-rank layouts and the rank-ordered combine on CPU: tests/test_distributed_gloo.py.

Runs in a child process with a timeout: a communicator that cannot come up must fail this test, not hang the suite.  No torch."""
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent

WORKER = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import numpy as np
from mixlab_amd import abi
from mixlab_amd.workspace import Workspace
from mixlab_amd.exchange import BusExchange, unique_id

def strips(n, sr=48000):
    ws = Workspace(sr, 60)
    mix = ws.mixer([(-1.5 * k, 1.0 - 0.05 * k, k % 3 == 0) for k in range(n)])
    for k in range(n):
        osc = ws.oscillator(110.0 * (k + 1), abi.WAVE_SAW if k % 2 else abi.WAVE_TRIANGLE)
        eq = ws.eq_three(3.0 - k, 0.5 * k, -2.0 + k); pan = ws.stereo_panner()
        ws.connect(osc, 0, eq, 0); ws.connect(eq, 0, pan, 0); ws.connect(eq, 0, pan, 1); ws.connect(pan, 0, mix, k)
    return ws, mix

def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)

def rccl_mapped():
    return any("librccl" in line for line in open("/proc/self/maps"))

# the library is loaded, a graph has run -- and RCCL is not in the process yet: it is bound on the first RCCL exchange
_ws, _mix = strips(2); _g = _ws.build(max_ticks_per_run=2, device=0); _g.run_ticks(0, 2); _g.sync()
assert not rccl_mapped(), "librccl was loaded before anything asked for the RCCL transport"
unique_id()
assert rccl_mapped(), "mx_exchange_unique_id did not bind librccl"
print("ok lazy-rccl", flush=True)

# (third shape: MX_FLAG_OVERLAP_TAIL and runs long enough for the speculative EqThree launch -- the Mixer bank of a run is held back for the next run's launch, and the
#  exchange's pack and collectives go out behind it when it is released: by the next run, or by whoever asks for the step's result first)
for sr, T, n, fl in ((48000, 8, 6, 0), (44100, 3, 5, 0), (48000, 16, 6, abi.FLAG_OVERLAP_TAIL)):          # 44.1 kHz x an odd tick count: bus lengths that are not multiples of a cache line
    for mode in ("allgather", "slices", "allreduce"):
        ws, mix = strips(n, sr)
        g = ws.build(max_ticks_per_run=T, device=0, flags=fl)
        assert (g.tail_stream() is not None) == bool(fl)
        ref = ws.build(max_ticks_per_run=T)                    # the same shard, run on its own: what the bus must be
        ex = BusExchange(g, mix, T, 0, 1, mode=mode, nccl_id=unique_id())
        assert ex.world == 1 and ex.mode == mode
        local_out = []
        for i in range(4):
            ref.run_ticks(i * T, T)
            local_out.append((ref.read_output(mix, 0, T, True), ref.read_output(mix, 1, T, True)))
        assert not np.array_equal(local_out[0][0], local_out[1][0])        # the steps really differ (oscillators run on absolute time)

        def check(i):
            m, c = ex.result(i)
            wm, wc = local_out[i]
            assert np.array_equal(bits(m), bits(wm)), f"{{mode}}: master of step {{i}}"
            assert np.array_equal(bits(c), bits(wc)), f"{{mode}}: cue of step {{i}}"

        # steps 0 and 1 in flight together (two slots), then step 2 reuses slot 0, step 3 slot 1
        g.run_ticks(0, T); ex.submit(0)
        g.run_ticks(T, T); ex.submit(1)
        check(0)
        g.run_ticks(2 * T, T); ex.submit(2)
        check(1)
        g.run_ticks(3 * T, T); ex.submit(3)
        check(2); check(3)
        assert ex.max_ulp_vs(3, *local_out[3]) == 0
        assert ex.bytes_received_per_step() == 0                           # a single rank receives nothing
        assert ex.elapsed_ms(3) > 0.0
        ex.close()
        print("ok", sr, mode, "overlap" if fl else "", flush=True)

# The AUTOMATIC second-stream mode (64 strips, 16 ticks per run) under an RCCL exchange, ended mid-way by a host that takes a raw pointer to the bus: a held submit is
# released by the sync inside, the later submits pack on the graph's stream, every step's bus stays the local Mixer's (round 6: Graph::end_auto_tail)
ws, mix = strips(64)
T = 16
g = ws.build(max_ticks_per_run=T, device=0)
assert g.tail_stream() is not None
ref = ws.build(max_ticks_per_run=T, flags=abi.FLAG_NO_FUSE)
ex = BusExchange(g, mix, T, 0, 1, mode="allgather", nccl_id=unique_id())
want = []
for i in range(5):
    ref.run_ticks(i * T, T)
    want.append((ref.read_output(mix, 0, T, True), ref.read_output(mix, 1, T, True)))
for i in range(5):
    g.run_ticks(i * T, T); ex.submit(i)
    if i == 2:
        g.output_device_ptr(mix, 0)                      # step 2's submit is being held for the next run: this releases it and ends the mode
        assert g.tail_stream() is None
    if i >= 1:
        m, c = ex.result(i - 1)
        assert np.array_equal(bits(m), bits(want[i - 1][0])) and np.array_equal(bits(c), bits(want[i - 1][1])), f"auto mode ended mid-way: step {{i - 1}}"
m, c = ex.result(4)
assert np.array_equal(bits(m), bits(want[4][0])) and np.array_equal(bits(c), bits(want[4][1]))
ex.close()
print("ok auto-mode-ended-under-an-exchange", flush=True)

ws, mix = strips(2)
g = ws.build(max_ticks_per_run=4, device=0)
try:
    BusExchange(g, mix, 4, 0, 1, mode="bogus", nccl_id=unique_id())
    raise SystemExit("a bogus mode was accepted")
except ValueError:
    print("ok bogus-mode", flush=True)
"""


def test_single_rank_rccl_exchange_through_the_c_abi_returns_the_local_bus_over_pipelined_steps(tmp_path):
    script = tmp_path / "exchange_worker.py"
    script.write_text(WORKER.format(root=str(ROOT)))
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    for sr in (48000, 44100):
        for mode in ("allgather", "slices", "allreduce"):
            assert f"ok {sr} {mode}" in res.stdout, res.stdout[-2000:]
    for mode in ("allgather", "slices", "allreduce"):
        assert f"ok 48000 {mode} overlap" in res.stdout, res.stdout[-2000:]
    assert "ok bogus-mode" in res.stdout and "ok lazy-rccl" in res.stdout and "ok auto-mode-ended-under-an-exchange" in res.stdout, res.stdout[-2000:]


def test_a_refused_loopback_submit_leaves_the_group_usable():
    """Ranks of a loopback group must submit the same step numbers; a submit that breaks the rule is refused BEFORE anything is packed or
    recorded, so the group carries on (the refusal used to come after the state was changed and wedged the group)."""
    import numpy as np
    from mixlab_amd import abi
    from mixlab_amd.exchange import BusExchange, LoopbackGroup
    from mixlab_amd.workspace import Workspace
    T = 4
    graphs, mixes = [], []
    for r in range(2):
        ws = Workspace(48000, 60)
        mix = ws.mixer([(0.0, 1.0, True)])
        osc = ws.oscillator(220.0 * (r + 1), abi.WAVE_SAW); ws.connect(osc, 1, mix, 0)
        graphs.append(ws.build(max_ticks_per_run=T)); mixes.append(mix)
    grp = LoopbackGroup(2)
    ex = [BusExchange(graphs[r], mixes[r], T, r, 2, mode="allgather", loopback=grp) for r in range(2)]
    for g in graphs:
        g.run_ticks(0, T)
    ex[0].submit(0)
    with pytest.raises(abi.MxError):
        ex[1].submit(1)                      # rank 0 is waiting with step 0
    with pytest.raises(abi.MxError):
        ex[0].submit(0)                      # rank 0 already submitted
    ex[1].submit(0)                          # the group is intact: the round completes
    want_m = graphs[0].read_output(mixes[0], 0, T, True) + graphs[1].read_output(mixes[1], 0, T, True)
    for r in range(2):
        m, _c = ex[r].result(0)
        assert np.array_equal(m.view(np.uint32), want_m.view(np.uint32))
    for g in graphs:
        g.run_ticks(T, T)
    ex[1].submit(1); ex[0].submit(1)
    assert np.array_equal(ex[0].result(1)[0], ex[1].result(1)[0])
    for e in ex:
        e.close()
    grp.close()


def test_library_binds_rccl_lazily_and_exports_the_exchange():
    lib = str(ROOT / "mixlab_amd" / "libmixlab_gpu.so")
    out = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True).stdout
    assert "librccl" not in out, "librccl is a load-time dependency again: hosts without RCCL could not load the library"
    blob = open(lib, "rb").read()
    for name in (b"librccl.so", b"ncclAllGather", b"ncclSend", b"ncclRecv", b"ncclAllReduce", b"ncclCommInitRank"):
        assert name in blob, f"{name!r}: the library no longer binds that RCCL entry point"
    syms = subprocess.run(["nm", "-D", "--defined-only", str(ROOT / "mixlab_amd" / "libmixlab_gpu.so")], capture_output=True, text=True).stdout
    for name in ("mx_exchange_create", "mx_exchange_submit", "mx_exchange_wait", "mx_exchange_result", "mx_exchange_unique_id", "mx_loopback_group_create"):
        assert f" T {name}" in syms
