"""Cut the committed causal prefix of the reference's EqThree golden pair.

The reference's only module-level golden vectors are
fixtures/module/eq_three/chronos.f32.raw -> chronos-eq.f32.raw (355 285 mono f32-LE samples,
gains +4/0/+4 dB, one run_tick call, exact equality: src/module/eq_three.rs:150-167).
The filter is causal, so the first k inputs determine the first k outputs; we commit the first
PREFIX samples of each file (data, not source) so the pin travels to boxes without /root/reference.

Run here (container with /root/reference):  python tests/golden/make_eq_three_prefix.py
"""
import pathlib
import sys

PREFIX = 131072  # samples (~2.97 s @ 44.1 kHz), 512 KiB per file
REF = pathlib.Path("/root/reference/fixtures/module/eq_three")
OUT = pathlib.Path(__file__).resolve().parent

def main() -> int:
    for name in ("chronos.f32.raw", "chronos-eq.f32.raw"):
        data = (REF / name).read_bytes()
        assert len(data) == 355285 * 4, len(data)
        (OUT / f"eq_three_{name.replace('.f32.raw', '')}_prefix{PREFIX}.f32.raw").write_bytes(data[: PREFIX * 4])
    return 0

if __name__ == "__main__":
    sys.exit(main())
