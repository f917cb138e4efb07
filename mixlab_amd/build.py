"""Build libmixlab_gpu.so (hand-written gfx950 kernels + the C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU.  The library is the product; there is no CPU
fallback: importing mixlab_amd.abi without it raises.
"""
from __future__ import annotations

import os
import pathlib
import shutil
import subprocess
import sys

PKG = pathlib.Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libmixlab_gpu.so"

# -ffp-contract=off is load-bearing: the reference (Rust) never fuses mul+add, and parity with it is
# bit-exact only if the device code does not either.
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-result",
]


def sources() -> list[pathlib.Path]:
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = sources() + list(CSRC.glob("*.hpp")) + [PKG.parent / "include" / "mixlab_gpu.h"]
    return any(p.stat().st_mtime > t for p in deps)


def build(force: bool = False, verbose: bool = True) -> pathlib.Path:
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libmixlab_gpu.so")
    tmp = LIB.with_suffix(".so.tmp")
    cmd = [hipcc, *HIPCC_FLAGS, "-o", str(tmp)]
    for s in sources():
        # host-only .cpp files are compiled as HIP too: they use the HIP runtime API and its headers
        cmd += ["-x", "hip", str(s)]
    if verbose:
        print("[mixlab_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=str(PKG))
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
