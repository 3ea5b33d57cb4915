"""Shared helpers of the parity tests (test infrastructure)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64)).view(np.uint64)


def same_bits(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def ulp_diff(a, b):
    """max |a-b| in units of the spacing of b (NaN positions must coincide)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(b)
    if not m.any():
        return 0.0
    with np.errstate(invalid="ignore"):
        d = np.abs(a[m] - b[m]) / np.spacing(np.maximum(np.abs(b[m]), np.finfo(float).tiny))
    d = np.where(a[m] == b[m], 0.0, d)
    return float(np.max(d)) if d.size else 0.0


def load_ops():
    with open(os.path.join(GOLD, "ops_cases.json")) as f:
        cases = json.load(f)
    arrays = dict(np.load(os.path.join(GOLD, "ops.npz")))
    return cases, arrays


def load_solvers():
    with open(os.path.join(GOLD, "solvers_cases.json")) as f:
        cases = json.load(f)
    arrays = dict(np.load(os.path.join(GOLD, "solvers.npz")))
    return cases, arrays


def load_meta():
    with open(os.path.join(GOLD, "META.json")) as f:
        return json.load(f)


def same_host_as_golden():
    """Reductions through OpenBLAS are only bit-reproducible on the numpy/BLAS build that made
    the fixtures; elsewhere the oracle is compared with a tolerance instead."""
    meta = load_meta()
    if meta.get("numpy") != np.__version__:
        return False
    try:
        from threadpoolctl import threadpool_info
        now = [{k: d.get(k) for k in ("internal_api", "version", "num_threads", "architecture")}
               for d in threadpool_info()]
        return now == meta.get("blas")
    except Exception:  # noqa: BLE001
        return False
