#!/usr/bin/env python
"""sha256 (first 16 hex digits) over the kernel sources a profile's numbers depend on.  profiles/rNN/*.json record it; bench.py copies a
profile's figures (PMC traffic, sustained clock) into its line only while the hash of the sources it runs still equals the recorded one."""
import hashlib
import pathlib

CSRC = pathlib.Path(__file__).resolve().parent.parent / "mixlab_amd" / "csrc"
FAMILIES = {
    "audio": ["mx_k_eq_exact.hip", "mx_k_eq_common.hpp", "mx_env_math.hpp", "mx_k_eq_three.hip", "mx_k_mixer.hip", "mx_k_envelope.hip", "mx_kernels.hpp"],
    "video": ["mx_k_video.hip", "mx_video.hpp"],
    "fir": ["mx_k_fir.hip", "mx_env_math.hpp"],
}


def kernel_hash(family: str) -> str:
    h = hashlib.sha256()
    for name in FAMILIES[family]:
        h.update(name.encode()); h.update((CSRC / name).read_bytes())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    import json
    print(json.dumps({f: kernel_hash(f) for f in FAMILIES}))
