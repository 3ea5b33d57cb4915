"""GPU: seeded differential fuzzing of the lock-step schedulers (batch.solve_batch_pipelined) against per-problem
solve().  Random solver (PGD / BBPGD / BBPGDf / APGD / anti-relaxation APGD / MPRGP / SPG with one generator per problem), batch composition (sizes 65..1300 mixed in one call, so several
lock-step groups), elementwise sets with array / scalar / integer / default bounds and -- PGD / BBPGD / APGD / AR --
Sphere sets and DisjointUnions of small and large blocks in the same call (the segmented forms), warm starts, per-problem
tolerances, matvec budget, run-ahead depth and matvec tuning word.  Every result must be bitwise the one of the
problem's own solve() (x, residual, converged flag, reported and executed MV counts)."""
import contextlib
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NCASES = int(os.environ.get("CCQP_LOCKSTEP_FUZZ_CASES", "42"))


def _set(rng, n, mprgp):
    from ccqppy_amd import solution_spaces as ss
    def arr(sign):
        return sign * rng.uniform(0.05, 0.6, size=n)
    def bound(sign):
        mode = rng.integers(4) if not mprgp else 3      # MPRGP: array bounds only (scalar ones raise, as the reference)
        if mode == 0:
            return None
        if mode == 1:
            return float(sign * rng.uniform(0.1, 0.8))
        if mode == 2:
            return int(sign * rng.integers(1, 3))
        return arr(sign)
    kind = rng.integers(4)
    if kind == 0:
        return ss.IdentityProjOp(n)
    if kind == 1:
        lb = bound(-1)
        return ss.LowerBoundProjOp(n) if lb is None else ss.LowerBoundProjOp(n, lb)
    if kind == 2:
        ub = bound(+1)
        return ss.UpperBoundProjOp(n) if ub is None else ss.UpperBoundProjOp(n, ub)
    lb, ub = bound(-1), bound(+1)
    if lb is None and ub is None:
        return ss.BoxProjOp(n)
    if lb is None:
        lb = -np.ones(n)
    if ub is None:
        ub = np.ones(n)
    return ss.BoxProjOp(n, lb, ub)


def _general_set(rng, n):
    """A non-elementwise set for the segmented lock-step forms (no Cone: a solve on it never converges, and with an
    infinite budget in the draw that would never end -- in the reference either)."""
    from ccqppy_amd import solution_spaces as ss
    kind = rng.integers(3)
    if kind == 0:
        return ss.SphereProjOp(n, float(rng.uniform(0.3, 5.0)))
    if kind == 1:
        k = n // 3
        ops = [ss.SphereProjOp(3, float(rng.uniform(0.1, 0.6)))] * k
        if n - 3 * k:
            ops.append(ss.IdentityProjOp(n - 3 * k))
        return ss.DisjointProjOp(*ops)
    a = int(rng.integers(1, n - 1))
    lo, hi = -rng.uniform(0.05, 0.6, size=a), rng.uniform(0.05, 0.6, size=a)
    return ss.DisjointProjOp(ss.BoxProjOp(a, lo, hi), ss.SphereProjOp(n - a, float(rng.uniform(0.3, 3.0))))


def _bits(x):
    x = x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)
    return x.view(np.uint64)


@pytest.mark.parametrize("case", range(NCASES))
def test_lockstep_equals_individual_solves(case):
    import torch
    from ccqppy_amd import batch, problems, solvers
    rng = np.random.default_rng(9000 + case)
    name = ["BBPGD", "APGD", "MPRGP", "APGDAntiRelaxation", "PGD", "BBPGDf", "SPG"][case % 7]
    base = getattr(solvers, "CCQPSolver" + name)
    spg_seeds = iter(())

    def cls(tol, cap):
        if name == "PGD":
            return base(tol, cap, 0.3)
        s = base(tol, cap)
        if name == "SPG":                               # its own generator: the k-th solver made draws from stream k
            s.rng = np.random.RandomState(next(spg_seeds))
        return s
    sizes = rng.choice([65, 96, 130, 257, 512, 777, 1300], size=rng.integers(1, 4), replace=False)
    probs, tols = [], []
    for n in sizes:
        for _ in range(int(rng.integers(1, 6))):
            n = int(n)
            op, b, _ = problems.random_problem(n, int(rng.integers(1 << 30)))
            x0 = None
            if rng.random() < 0.3:
                x0 = torch.as_tensor(rng.uniform(-0.5, 0.5, size=n), device=b.device)
            general = rng.random() < 0.35               # (every solver has a segmented lock-step form)
            probs.append((op, b, x0, _general_set(rng, n) if general else _set(rng, n, name == "MPRGP")))
            tols.append(float(10.0 ** rng.uniform(-9, -4)) if name != "MPRGP" else float(10.0 ** rng.uniform(-7, -4)))
    order = rng.permutation(len(probs))                 # sizes interleaved in the call
    probs = [probs[i] for i in order]
    tols = [tols[i] for i in order]
    max_mv = [np.inf, 5000, 60, 25, 9][rng.integers(5)] if name != "MPRGP" else [3000, 60, 25, 9][rng.integers(4)]
    if name == "SPG" and max_mv == np.inf:
        max_mv = 5000
    ahead = int(rng.choice([1, 2, 4, 9]))
    tune = int(rng.choice([0, (1 << 1) | (2 << 4), (2 << 1) | (7 << 4) | (8 << 8), (4 << 4) | (3 << 8)]))
    ref = []
    spg_seeds = iter(range(500, 500 + len(probs)))
    for (A, b, x0, P), tol in zip(probs, tols):
        with contextlib.redirect_stdout(io.StringIO()):
            ref.append(cls(tol, max_mv).solve(A, b, x0=x0, convex_proj_op=P))
    spg_seeds = iter(range(500, 500 + len(probs)))
    it = iter(tols)
    from ccqppy_amd import lockstep
    before = lockstep.stats["fallback_solves"]
    got = batch.solve_batch_pipelined(lambda: cls(next(it), max_mv), probs, ahead=ahead, tune=tune)
    # far from the matvec budget only a solve that does not converge at all may go back to the host loop
    assert max_mv < 3000 or lockstep.stats["fallback_solves"] - before <= sum(not c.solution_converged for c in got), \
        "a device-resident solve fell back to the host loop"
    assert len(got) == len(ref)
    for a, c in zip(ref, got):
        assert a.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
        # (lock-step SPG counts the matvec that ends a budget but does not execute it)
        skipped = 1 if name == "SPG" and not c.solution_converged else 0
        assert a.executed_matrix_vector_multiplications == c.executed_matrix_vector_multiplications + skipped
        assert bool(a.solution_converged) == bool(c.solution_converged)
        assert float(a.solution_residual) == float(c.solution_residual)
        assert np.array_equal(_bits(a.solution), _bits(c.solution))
        if name == "SPG":
            sa, sc = a.rng.get_state(), c.rng.get_state()
            assert sa[2] == sc[2] and np.array_equal(sa[1], sc[1])
