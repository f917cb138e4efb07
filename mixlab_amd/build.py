"""Build libmixlab_gpu.so (hand-written gfx950 kernels + the C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU.  The library is the product; there is no CPU
fallback: importing mixlab_amd.abi without it raises.

Every source is compiled to its own object under mixlab_amd/build/ (git-ignored), in parallel, and
only when it or a header changed; the objects are then linked into the shared library.
"""
from __future__ import annotations

import concurrent.futures
import os
import pathlib
import shutil
import subprocess
import sys

PKG = pathlib.Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = PKG / "build"
LIB = PKG / "libmixlab_gpu.so"

# -ffp-contract=off is load-bearing: the reference (Rust) never fuses mul+add, and parity with it is
# bit-exact only if the device code does not either.
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-result",
]


def sources() -> list[pathlib.Path]:
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def headers() -> list[pathlib.Path]:
    return sorted(CSRC.glob("*.hpp")) + [PKG.parent / "include" / "mixlab_gpu.h"]


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in sources() + headers())


def _hipcc() -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libmixlab_gpu.so")
    return hipcc


def _compile_one(hipcc: str, src: pathlib.Path, obj: pathlib.Path, verbose: bool) -> None:
    # host-only .cpp files are compiled as HIP too: they use the HIP runtime API and its headers
    cmd = [hipcc, *HIPCC_FLAGS, "-c", "-x", "hip", str(src), "-o", str(obj)]
    if verbose:
        print("[mixlab_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=str(PKG))


def build(force: bool = False, verbose: bool = True) -> pathlib.Path:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    OBJ.mkdir(exist_ok=True)
    hdr_t = max(p.stat().st_mtime for p in headers())
    jobs, objs = [], []
    for s in sources():
        o = OBJ / (s.name + ".o")
        objs.append(o)
        if force or not o.exists() or o.stat().st_mtime < max(s.stat().st_mtime, hdr_t):
            jobs.append((s, o))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for f in [ex.submit(_compile_one, hipcc, s, o, verbose) for (s, o) in jobs]:
            f.result()
    tmp = LIB.with_suffix(".so.tmp")
    # librccl is NOT a link-time dependency: the bus exchange (mx_exchange_*) calls ncclAllGather / ncclSend / ncclRecv itself through
    # entry points it binds with dlopen on first use (mx_exchange.cpp), so hosts without RCCL can build and load the library
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(tmp), *[str(o) for o in objs], "-ldl"]
    if verbose:
        print("[mixlab_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=str(PKG))
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
