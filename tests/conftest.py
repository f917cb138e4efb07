"""pytest configuration: `gpu` marker; builds the CPU oracle (test infrastructure) on demand."""
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    lib = ROOT / "oracle" / "libmixlab_oracle.so"
    srcs = list((ROOT / "oracle").glob("*.c")) + list((ROOT / "oracle").glob("*.h"))
    if not lib.exists() or any(s.stat().st_mtime > lib.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    yield
