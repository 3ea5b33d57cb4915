import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """libccqp_hip.so, built in-tree if it is not there yet (hipcc cross-compiles without a GPU)."""
    lib = os.path.join(ROOT, "ccqppy_amd", "libccqp_hip.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-j4"], cwd=ROOT, check=True, capture_output=True)
    return lib
