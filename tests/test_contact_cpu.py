"""CPU side of the matrix-free operator row: A = D diag(minv) D^T + eps I handed to the
solvers as an object with `.dot` only.  The fixtures come from the REFERENCE run on that object
(`oracle/gen_golden.py contact`); here the oracle is held to them."""
import numpy as np
import pytest

from oracle import cases as C
from oracle import cpu_ref, gen_ref
from tests import helpers as H

CASES, ARR = H.load_contact()
SAME_HOST = H.same_host_as_golden()
_A = {}


def _hessian(pid):
    if pid not in _A:
        D, minv, eps = H.contact_fixture_parts(pid, ARR)
        _A[pid] = gen_ref.ContactHessian(D, minv, eps)
    return _A[pid]


def test_contact_fixture_inventory_and_generator():
    assert [c["id"] for c in CASES] == [c["id"] for c in C.contact_solver_cases()] and len(CASES) == 18
    for pid in sorted({c["problem"] for c in CASES}):
        _, m, n, seed = pid.split(":")
        A, b, _ = gen_ref.contact_problem(int(m), int(n), int(seed))
        D, minv, eps = H.contact_fixture_parts(pid, ARR)
        assert H.same_bits(minv, A.minv) and eps == A.eps
        assert np.array_equal(D.indptr, A.D.indptr) and np.array_equal(D.indices, A.D.indices)
        assert H.same_bits(D.data, A.D.data) and H.same_bits(b, ARR[pid + ".b"])
        v = np.random.default_rng(1).normal(size=int(m))
        np.testing.assert_allclose(A.dot(v), A.toarray().dot(v), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("case", [c for c in CASES if c["executed_mv"] <= 1100], ids=lambda c: c["id"])
def test_oracle_reproduces_reference_on_matrix_free_A(case):
    A = _hessian(case["problem"])
    b = ARR[case["problem"] + ".b"]
    m = b.shape[0]
    P = None if case["op"] is None else C.build_op(case["op"], C.oracle_factory(), dict(C.SUITE_ARRAYS))
    np.random.seed(case["rng_seed"])
    with np.errstate(all="ignore"):
        r = cpu_ref.SOLVERS[case["solver"]](A, b, case["tol"], case["max_mv"], P=P, trace=[], **case["kwargs"])
    want = ARR[case["id"] + ".x"]
    if SAME_HOST:
        assert r.solution_num_matrix_vector_multiplications == case["mv"]
        assert H.same_bits(r.solution, want)
        assert len(r.trace) == case["trace_len"]
        assert H.same_bits(C.trace_array(case["solver"], r.trace), ARR[case["id"] + ".trace"])
    else:
        lo, hi = case["mv_band"]
        assert lo - (hi - lo) <= r.solution_num_matrix_vector_multiplications <= hi + (hi - lo)
        np.testing.assert_allclose(r.solution, want, rtol=max(1e-10, 50 * case["x_spread"]), atol=1e-10)
    assert bool(r.solution_converged) == case["converged"]
