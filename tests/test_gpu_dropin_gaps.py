"""GPU: drop-in gaps closed in round 2.

* arbitrary callables as `convex_proj_op` (the reference accepts anything callable, solvers.py:115-116);
* public bound / radius attributes edited between calls are honoured (the reference re-reads them on every
  call, solution_spaces.py:200-201,276-277,363-366,430-433);
* `ProjOpBase.plot` (solution_spaces.py:48-74)."""
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import cpu_ref, gen_ref
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.mark.parametrize("sname", ["PGD", "APGD", "APGDAntiRelaxation", "BBPGD", "BBPGDf", "SPG"])
def test_lambda_clamp_equals_box(sname):
    """A lambda clamp goes through the host round trip, BoxProjOp through the kernels: same iterates.  (np.clip and
    the reference's mask expression agree bit for bit away from +-0 / inf / NaN, none of which occur here.)"""
    from ccqppy_amd import solvers, solution_spaces as ss
    n = 192
    A = gen_ref.spd_rows(n, 5)
    b = -A.dot(gen_ref.xstar(n, 5))
    lb, ub = -0.4 * np.ones(n), 0.3 * np.ones(n)
    cls = getattr(solvers, "CCQPSolver" + sname)
    np.random.seed(3)
    box = _quiet(cls(1e-8, 4000).solve, A, b, convex_proj_op=ss.BoxProjOp(n, lb, ub))
    np.random.seed(3)
    lam = _quiet(cls(1e-8, 4000).solve, A, b, convex_proj_op=lambda x: np.clip(x, lb, ub))
    assert bool(lam.solution_converged) == bool(box.solution_converged)
    assert lam.solution_num_matrix_vector_multiplications == box.solution_num_matrix_vector_multiplications
    assert np.linalg.norm(lam.solution - box.solution) <= 1e-12 * np.linalg.norm(box.solution)
    # and against the oracle driven by the same lambda (the reference's own use of a callable)
    np.random.seed(3)
    with np.errstate(all="ignore"):
        o = cpu_ref.SOLVERS[sname](A, b, 1e-8, 4000, P=lambda x: np.clip(x, lb, ub))
    assert lam.solution_num_matrix_vector_multiplications == o.solution_num_matrix_vector_multiplications
    assert np.linalg.norm(lam.solution - o.solution) <= 1e-10 * np.linalg.norm(o.solution)


def test_callable_object_with_normal_vector_under_mprgp_and_errors():
    from ccqppy_amd import solvers, solution_spaces as ss
    n = 128
    A = gen_ref.spd_rows(n, 9)
    b = -A.dot(gen_ref.xstar(n, 9))
    lb, ub = -0.5 * np.ones(n), 0.5 * np.ones(n)
    ref_box = cpu_ref.Box(n, lb, ub)

    class UserSet:                      # a user-defined set: callable + normal_vector, nothing else
        def __call__(self, x):
            return ref_box(x)

        def normal_vector(self, x):
            return ref_box.normal_vector(x)

    got = _quiet(solvers.CCQPSolverMPRGP(1e-6, 300).solve, A, b, convex_proj_op=UserSet())
    box = _quiet(solvers.CCQPSolverMPRGP(1e-6, 300).solve, A, b, convex_proj_op=ss.BoxProjOp(n, lb, ub))
    assert got.solution_num_matrix_vector_multiplications == box.solution_num_matrix_vector_multiplications
    assert np.linalg.norm(got.solution - box.solution) <= 1e-12 * np.linalg.norm(box.solution)
    # a bare function has no normal_vector: the reference's MPRGP raises AttributeError at :1095
    with pytest.raises(AttributeError):
        _quiet(solvers.CCQPSolverMPRGP(1e-6, 300).solve, A, b, convex_proj_op=lambda x: np.clip(x, lb, ub))
    # not callable at all: TypeError, as the reference's first `convex_proj_op(x)`
    with pytest.raises(TypeError):
        _quiet(solvers.CCQPSolverBBPGD(1e-6, 300).solve, A, b, convex_proj_op=42)
    # torch in, torch out
    P = ss.HostCallableProjOp(n, lambda x: np.clip(x, lb, ub))
    x = torch.randn(n, dtype=torch.float64, device="cuda")
    assert torch.equal(P(x), torch.clamp(x, -0.5, 0.5))


@pytest.mark.parametrize("n", [40, 5000])
def test_bounds_edited_in_place_between_calls(n):
    """The advisor's case: `P.lower_bound[:] = ...` between two projections / two solves."""
    from ccqppy_amd import solvers, solution_spaces as ss
    rng = np.random.default_rng(n)
    x = rng.normal(size=n)
    lb, ub = -0.5 * np.ones(n), 0.5 * np.ones(n)
    P = ss.BoxProjOp(n, lb, ub)
    assert H.same_bits(P(x), cpu_ref.Box(n, lb, ub)(x))
    P.lower_bound[:] = -0.1                          # in place: same object, new contents
    assert H.same_bits(P(x), cpu_ref.Box(n, lb, ub)(x))
    assert float(P(x).min()) == -0.1
    P.upper_bound[::2] = 0.05
    assert H.same_bits(P(x), cpu_ref.Box(n, lb, ub)(x))
    assert H.same_bits(P.normal_vector(P(x)), cpu_ref.Box(n, lb, ub).normal_vector(P(x)))
    P.upper_bound = 0.2 * np.ones(n)                 # reassigned
    assert H.same_bits(P(x), cpu_ref.Box(n, lb, P.upper_bound)(x))
    for k in range(40):                              # many short-lived arrays: ids get recycled by CPython
        P.upper_bound = (0.2 + 0.01 * k) * np.ones(n)
        assert float(P(x).max()) == P.upper_bound[0]
    # a solve after an edit sees the new bounds (time-stepping pattern: same operator object, new bounds)
    A = gen_ref.spd_rows(n, 2)
    b = -A.dot(gen_ref.xstar(n, 2))
    Q = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    first = _quiet(solvers.CCQPSolverBBPGD(1e-9, 3000).solve, A, b, convex_proj_op=Q)
    Q.lower_bound[:] = 0.0
    Q.upper_bound[:] = 0.25
    second = _quiet(solvers.CCQPSolverBBPGD(1e-9, 3000).solve, A, b, convex_proj_op=Q)
    o = cpu_ref.solve_bbpgd(A, b, 1e-9, 3000, P=cpu_ref.Box(n, np.zeros(n), 0.25 * np.ones(n)))
    assert second.solution_num_matrix_vector_multiplications == o.solution_num_matrix_vector_multiplications
    assert np.linalg.norm(second.solution - o.solution) <= 1e-10 * np.linalg.norm(o.solution)
    assert float(second.solution.min()) >= 0.0 and float(first.solution.min()) < 0.0


def test_radius_and_nested_sets_edited_between_calls():
    from ccqppy_amd import solution_spaces as ss
    rng = np.random.default_rng(0)
    x = 3.0 * rng.normal(size=12)
    S = ss.SphereProjOp(12, 1.0)
    assert H.same_bits(S(x), cpu_ref.Sphere(12, 1.0)(x))
    S.radius = 2.5
    assert H.same_bits(S(x), cpu_ref.Sphere(12, 2.5)(x))
    lb = -np.ones(5)
    B = ss.BoxProjOp(5, lb, np.ones(5))
    D = ss.DisjointProjOp(B, ss.SphereProjOp(7, 1.0))
    Do = lambda: cpu_ref.Disjoint(cpu_ref.Box(5, lb, np.ones(5)), cpu_ref.Sphere(7, D.proj_ops[1].radius))  # noqa: E731
    assert H.same_bits(D(x), Do()(x))
    lb[:] = 0.5                                      # the child's bound, in place
    assert H.same_bits(D(x), Do()(x))
    B(x[:5])                                         # the child refreshed on its own, then edited again
    lb[:] = -0.25
    B(x[:5])
    assert H.same_bits(D(x), Do()(x))
    D.proj_ops[1].radius = 0.125
    assert H.same_bits(D(x), Do()(x))


def test_plot_runs_headless(monkeypatch):
    """solution_spaces.py:48-74: scatter of projected random samples; host-side, matplotlib imported on demand."""
    import matplotlib
    matplotlib.use("Agg", force=True)
    import matplotlib.pyplot as plt
    from ccqppy_amd import solution_spaces as ss
    shown = []
    monkeypatch.setattr(plt, "show", lambda *a, **k: shown.append(len(plt.get_fignums())))
    np.random.seed(0)
    for op in (ss.BoxProjOp(3, -0.5 * np.ones(3), 0.5 * np.ones(3)), ss.SphereProjOp(2, 1.0), ss.ConeProjOp(3),
               ss.LowerBoundProjOp(1, np.zeros(1))):
        op.plot(5, -2.0, 2.0)
        ax = plt.gcf().axes[0]
        pts = np.concatenate([np.asarray(c._offsets3d).T for c in ax.collections])
        assert pts.shape == (5, 3)
        if isinstance(op, ss.BoxProjOp):
            assert np.all(np.abs(pts) <= 0.5)
        if isinstance(op, ss.SphereProjOp):
            assert np.all(np.linalg.norm(pts, axis=1) <= 1.0 + 1e-12)
        plt.close("all")
    assert len(shown) == 4
