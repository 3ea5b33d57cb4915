"""GPU: seeded differential fuzzing of the whole stack against the CPU oracle -- random operator trees
(nested DisjointUnions of all seven operators, array / scalar / integer bounds), random solver, size,
tolerance and MV budget.  For every case the oracle is also run on 1-ulp-perturbed inputs: where ITS OWN
MV count does not move, the HIP result must match it exactly (count, executed count, converged flag) and
x to 1e-9; where it moves, the HIP count must lie in the oracle's band.  Exceptions must have the same
type."""
import contextlib
import io

import numpy as np
import pytest

from oracle import cases as C
from oracle import cpu_ref, gen_ref

pytestmark = pytest.mark.gpu

import os

# CCQP_FUZZ_CASES / CCQP_FUZZ_SPARSE widen the sweep for a one-off soak run (defaults: what the suite runs)
NCASES = int(os.environ.get("CCQP_FUZZ_CASES", "240"))
NSPARSE = int(os.environ.get("CCQP_FUZZ_SPARSE", "80"))


def _rand_leaf(rng, d, arrays, tag, allow_cone=True):
    kinds = ["Identity", "Lower", "Upper", "Box", "Sphere"] + (["Cone"] if allow_cone else [])
    k = kinds[rng.integers(len(kinds))]
    spec = {"kind": k, "dim": d}

    def bound(name, sign):
        mode = rng.integers(4)
        if mode == 0:
            return None                                   # the reference's default (+-1 arrays)
        if mode == 1:
            return {"fill": float(sign * rng.uniform(0.2, 1.5))}
        if mode == 2:
            return {"fill": int(sign * rng.integers(1, 3)), "int": True}
        key = "%s.%s" % (tag, name)
        arrays[key] = sign * rng.uniform(0.2, 1.5, size=d)
        return {"arr": key}
    if k in ("Lower", "Box"):
        spec["lb"] = bound("lb", -1)
    if k in ("Upper", "Box"):
        spec["ub"] = bound("ub", +1)
    if k == "Sphere":
        spec["radius"] = [None, 0.7, 2, 3.5][rng.integers(4)]
    if k == "Cone":
        spec["aspect_ratio"] = [None, 0.5, 2][rng.integers(3)]
    return spec


def _rand_tree(rng, n, arrays, tag, depth=0):
    if depth >= 2 or n < 4 or rng.random() < 0.35:
        return _rand_leaf(rng, n, arrays, tag)
    parts = []
    left = n
    i = 0
    while left > 0:
        d = int(min(left, rng.integers(1, max(2, n // 2) + 1)))
        parts.append(_rand_tree(rng, d, arrays, "%s.%d" % (tag, i), depth + 1))
        left -= d
        i += 1
    return {"kind": "Disjoint", "ops": parts}


def _has(spec, kind):
    if spec["kind"] == kind:
        return True
    return spec["kind"] == "Disjoint" and any(_has(s, kind) for s in spec["ops"])


def _run(fn):
    try:
        with np.errstate(all="ignore"):
            return fn(), None
    except Exception as e:  # noqa: BLE001
        return None, type(e).__name__


@pytest.mark.parametrize("case_id", range(NCASES))
def test_fuzz_case(case_id):
    _fuzz(case_id, sparse=False)


@pytest.mark.parametrize("case_id", range(NSPARSE))
def test_fuzz_sparse_case(case_id):
    """Same differential test with A an irregular scipy.sparse matrix (rows of 1..dozens of entries) handed to
    solve() as such: SparseOperator / ccqp_spmv_csr_f64 on the product side, scipy's CSR dot in the oracle."""
    _fuzz(50000 + case_id, sparse=True)


def _fuzz(case_id, sparse):
    from ccqppy_amd import problems, solvers
    rng = np.random.default_rng(9000 + case_id)
    n = int(rng.choice([5, 17, 64, 100, 200, 333]))
    seed = int(rng.integers(1, 10 ** 6))
    arrays = {}
    spec = _rand_tree(rng, n, arrays, "f%d" % case_id)
    sname = C.SOLVER_NAMES[rng.integers(len(C.SOLVER_NAMES))]
    tol = float(rng.choice([1e-5, 1e-7]))
    max_mv = int(rng.choice([25, 40, 300, 3000]))
    if _has(spec, "Cone"):
        max_mv = min(max_mv, 40)                       # the bugged Cone never converges and amplifies ulps
    kwargs = {"step_size": 0.3} if sname == "PGD" else {}
    x0 = None if rng.random() < 0.6 else rng.uniform(-0.5, 0.5, size=n)

    if sparse:
        A = gen_ref.irregular_spd(n, seed % 1000, density=float(rng.choice([0.02, 0.1, 0.3])))
        op_dev = A                                       # scipy matrix straight into solve()
        b = -A.dot(gen_ref.xstar(n, seed))
    else:
        A = gen_ref.spd_rows(n, seed)
        op_dev, b_dev, _ = problems.random_problem(n, seed)
        b = b_dev.cpu().numpy()
    Po = C.build_op(spec, C.oracle_factory(), arrays)
    Pp = C.build_op(spec, C.product_factory(), arrays)

    def oracle(Ax, bx):
        np.random.seed(77)
        return cpu_ref.SOLVERS[sname](Ax, bx, tol, max_mv, x0=None if x0 is None else x0.copy(), P=Po, **kwargs)

    o, oerr = _run(lambda: oracle(A, b))

    def product():
        np.random.seed(77)
        with contextlib.redirect_stdout(io.StringIO()):
            return getattr(solvers, "CCQPSolver" + sname)(tol, max_mv, **kwargs).solve(
                op_dev, b, x0=None if x0 is None else x0.copy(), convex_proj_op=Pp)

    r, rerr = _run(product)
    assert rerr == oerr, (spec, sname)
    if oerr is not None:
        return
    mvs = [o.solution_num_matrix_vector_multiplications]
    spread = 0.0
    prng = np.random.default_rng(case_id)
    for _ in range(4):
        bp = b * (1 + prng.integers(-1, 2, size=n) * 2.0 ** -52)
        if sparse:
            Ap = A.copy()
            Ap.data = A.data * (1 + prng.integers(-1, 2, size=A.data.shape) * 2.0 ** -52)
        else:
            Ap = A * (1 + prng.integers(-1, 2, size=A.shape) * 2.0 ** -52)
        po, perr = _run(lambda: oracle(Ap, bp))
        if perr is None:
            mvs.append(po.solution_num_matrix_vector_multiplications)
            # the spread sample: only where both solutions are finite (a diverging case is compared entry by entry for
            # finiteness below and contributes no spread -- an explicit exclusion), scaled by the largest entry first so
            # that the norms of huge but finite iterates cannot overflow to inf / inf
            if np.all(np.isfinite(po.solution)) and np.all(np.isfinite(o.solution)):
                scale = max(float(np.max(np.abs(o.solution))), 1e-300)
                den = max(float(np.linalg.norm(o.solution / scale)), 1e-300)
                sample = float(np.linalg.norm((po.solution - o.solution) / scale) / den)
                assert np.isfinite(sample), (spec, sname, "spread sample is not finite")
                spread = max(spread, sample)
    lo, hi = min(mvs), max(mvs)
    got = r.solution_num_matrix_vector_multiplications
    if lo == hi:
        assert got == lo, (spec, sname, got, lo)
        assert r.executed_matrix_vector_multiplications == o.executed_mv
        assert bool(r.solution_converged) == bool(o.solution_converged)
    else:
        assert lo - (hi - lo) <= got <= hi + (hi - lo), (spec, sname, got, lo, hi)
    fin = np.isfinite(o.solution)
    assert np.array_equal(np.isfinite(r.solution), fin)
    rtol = max(1e-9, 50 * spread)
    den = max(np.linalg.norm(o.solution[fin]), 1e-300)
    assert np.linalg.norm((r.solution - o.solution)[fin]) <= rtol * den + 1e-13, (spec, sname)
