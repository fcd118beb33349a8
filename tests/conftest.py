import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


def pytest_sessionstart(session):
    """Build the CPU-emulation build of the kernel layer (tests only) if it is missing -- a few seconds with g++."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "tests", "_build", "libcolorneus_emu.so")
    if not os.path.isfile(lib) and shutil.which("g++") and shutil.which("make"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "color-neus_amd", "csrc"), "-j", "4", "emu"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
