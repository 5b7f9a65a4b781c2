import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

DATA = os.path.join(ROOT, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """CPU-side artefacts are cheap to build; build them once if a fresh checkout lacks them."""
    import subprocess
    need = [os.path.join(ROOT, "hexray_amd", "lib", "libhexray_host.so"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.run(["bash", os.path.join(ROOT, "build.sh"), "host"], check=True)
        subprocess.run(["bash", os.path.join(ROOT, "build.sh"), "oracle"], check=True)
    if not os.path.exists(os.path.join(ROOT, "hexray_amd", "lib", "libhexray_hip.so")):
        subprocess.run(["bash", os.path.join(ROOT, "build.sh"), "hip"], check=True)
    if not os.path.exists(os.path.join(ROOT, "hexray_amd", "bin", "hexray")):
        subprocess.run(["bash", os.path.join(ROOT, "build.sh"), "cli"], check=True)


_ensure_built()


@pytest.fixture(scope="session")
def data_dir():
    return DATA


def scene_path(name):
    return os.path.join(DATA, name)


@pytest.fixture(scope="session")
def gpu():
    """Initialise the HIP module on device 0; the GPU tests must run the native path or fail."""
    import hexray_amd as hx
    hx.init(0)
    return hx
