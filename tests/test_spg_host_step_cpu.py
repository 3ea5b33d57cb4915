"""The host half of the lock-step SPG driver (ccqp_bspg_host_step, csrc/ccqp_bspg.hip) against the scalar code of
CCQPSolverSPG._iterate evaluated by Python itself (reference: solvers.py:949-966): libm pow vs `**`, Python's min / max
semantics, numpy's legacy `uniform`, the bounded queue.  No GPU: the function is plain host code of the library."""
import ctypes
from collections import deque

import numpy as np
import pytest

from ccqppy_amd import _lib


def python_step(tol, tau, sig1, sig2, fk, queue, dd, dAd, dg, rng):
    """The statements of ccqppy_amd/solvers.py (CCQPSolverSPG._iterate) between the dot products and the next
    direction, with the types they have there: dot products np.float64, parameters Python floats."""
    with np.errstate(all="ignore"):
        dkdotdk, dkdotAdk, dkdotgk = np.float64(dd), np.float64(dAd), np.float64(dg)
        if np.sqrt(dkdotdk) <= tol:
            return "converged", fk, None, None
        fmax = max(queue)
        xi = (fmax - fk) / dkdotAdk
        beta = -dkdotgk / dkdotAdk
        betahat = tau * beta + np.sqrt((tau ** 2) * (beta ** 2) + 2 * xi)
        try:
            betak = rng.uniform(low=sig1, high=min(betahat, sig2))
        except OverflowError:
            return "overflow", fk, None, None
        fk = fk + (betak * betak * dkdotgk + 0.5 * (betak ** 2) * dkdotAdk)
        queue.append(fk)
        alpha = dkdotdk / dkdotAdk
        return "running", fk, alpha, betak


def bits(v):
    return np.float64(v).view(np.uint64)


def run_case(lib, rs, special=None, use_source=False):
    m = int(rs.integers(1, 8))
    tau, sig1, sig2 = float(rs.uniform(0.1, 0.9)), float(rs.uniform(0.001, 0.1)), float(rs.uniform(0.2, 0.9))
    tol = 1e-6
    scale = float(np.exp(rs.uniform(-20, 20)))
    qlen = int(rs.integers(1, m + 1))
    qvals = [np.float64(v) for v in rs.standard_normal(qlen) * scale]
    fk = qvals[-1]
    dd = float(abs(rs.standard_normal()) * np.exp(rs.uniform(-30, 10)))
    dAd = float(abs(rs.standard_normal()) * np.exp(rs.uniform(-30, 10)))
    dg = float(-abs(rs.standard_normal()) * np.exp(rs.uniform(-30, 10)))
    if special == "zero_dAd":
        dAd = 0.0
    elif special == "nan":
        dg = float("nan")
    elif special == "neg_curv":
        dAd = -dAd
    elif special == "tiny":
        dd = 1e-30
    elif special == "nan_queue":
        qvals[0] = np.float64("nan")
        fk = qvals[-1]
    seed = int(rs.integers(0, 2 ** 31))
    rng = np.random.RandomState(seed)
    u = np.random.RandomState(seed).random_sample()
    queue = deque(qvals, maxlen=m)
    status, fk_py, alpha_py, betak_py = python_step(tol, tau, sig1, sig2, fk, queue, dd, dAd, dg, rng)

    h = _lib.SpgHost()
    h.tol, h.max_mv, h.tau, h.sigma1, h.sigma2 = tol, np.inf, tau, sig1, sig2
    h.fk, h.m, h.qlen = float(fk), m, qlen
    q = np.zeros(8)
    q[:qlen] = qvals
    dots = np.array([dd, dAd, dg])
    uu = np.array([u])
    if use_source:                                       # the draw straight from a RandomState's bit generator
        from ccqppy_amd.lockstep import _draw_source
        twin = np.random.RandomState(seed)
        src = np.array([_draw_source(twin)], dtype=np.uint64)
        _lib.check(lib.ccqp_bspg_host_step(ctypes.addressof(h), q.ctypes.data, 8, dots.ctypes.data, src.ctypes.data,
                                           None, None, 1))
        # ... which is left where the Python loop's generator is
        assert twin.get_state()[2] == rng.get_state()[2] and np.array_equal(twin.get_state()[1], rng.get_state()[1])
    else:
        _lib.check(lib.ccqp_bspg_host_step(ctypes.addressof(h), q.ctypes.data, 8, dots.ctypes.data, None,
                                           uu.ctypes.data, None, 1))
    if status == "converged":
        assert h.status == 1 and h.pending == 0 and h.used == 0
        return status, False
    if status == "overflow":
        assert h.status == 3 and h.used == 0
        return status, False
    assert h.status == 0 and h.pending == 1 and h.used == 1
    assert bits(h.betak) == bits(betak_py)
    assert bits(h.fk) == bits(fk_py)
    assert bits(h.alpha) == bits(alpha_py)
    assert h.qlen == len(queue)
    assert np.array_equal(q[:h.qlen].view(np.uint64), np.array(list(queue), dtype=np.float64).view(np.uint64))
    # does this case tell pow(x, 2) from x*x?
    with np.errstate(all="ignore"):
        beta = -np.float64(dg) / np.float64(dAd)
        differs = (beta ** 2) != beta * beta or (betak_py ** 2) != betak_py * betak_py
    return status, bool(differs)


def test_host_step_matches_python_bit_for_bit():
    lib = _lib.load()
    rs = np.random.default_rng(20261004)
    seen = {"running": 0, "converged": 0, "overflow": 0}
    told_apart = 0
    specials = [None] * 12 + ["zero_dAd", "nan", "neg_curv", "tiny", "nan_queue"]
    for k in range(40000):
        status, differs = run_case(lib, rs, specials[k % len(specials)], use_source=k % 3 == 0)
        seen[status] += 1
        told_apart += differs
    assert seen["running"] > 20000 and seen["converged"] > 100 and seen["overflow"] > 100
    # pow(x, 2) != x*x for roughly one argument in a thousand: the sample must contain such cases, or the test
    # would not notice a library that squares by multiplication
    assert told_apart >= 10


def test_host_step_checks_its_arguments():
    lib = _lib.load()
    h = _lib.SpgHost()
    h.m, h.qlen = 0, 0
    q, d, u = np.zeros(8), np.zeros(3), np.zeros(1)
    rc = lib.ccqp_bspg_host_step(ctypes.addressof(h), q.ctypes.data, 8, d.ctypes.data, None, u.ctypes.data, None, 1)
    assert rc != 0 and b"queue" in lib.ccqp_last_error()
