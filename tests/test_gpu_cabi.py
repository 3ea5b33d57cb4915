"""GPU: the C ABI used from plain C (no Python, no torch) -- tests/c/abi_smoke.c is compiled against
include/ccqp_hip.h + libccqp_hip.so and run as its own process."""
import os
import shutil
import subprocess

import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


def test_c_program_links_and_runs(tmp_path, built_lib):
    cc = shutil.which("gcc") or "cc"          # a plain C compiler: the ABI needs nothing else
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(built_lib)
    subprocess.run([cc, "-std=c99", os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
                    "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                    "-L" + libdir, "-lccqp_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                    "-Wl,-rpath," + libdir + ":/opt/rocm/lib", "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches = 0" in r.stdout and "gfx950" in r.stdout
