"""CPU: host-side logic of the product that needs no GPU -- API surface, names, partitioning,
Disjoint flattening.  (Reference strings: solvers.py:174,347,537,673,823,979,1204;
solution_spaces.py:85,139,215,299,382,452,505.)"""
import inspect

import numpy as np

from ccqppy_amd import operators, solution_spaces as ss, solvers


def test_solver_surface_matches_reference():
    names = {solvers.CCQPSolverPGD: "PGD", solvers.CCQPSolverAPGD: "APGD",
             solvers.CCQPSolverAPGDAntiRelaxation: "Anti-relaxation APGD",
             solvers.CCQPSolverBBPGD: "BBGPD", solvers.CCQPSolverBBPGDf: "BBPDGf",
             solvers.CCQPSolverSPG: "SPG-QP", solvers.CCQPSolverMPRGP: "MPRGP"}
    for cls, nm in names.items():
        s = cls(1e-8)
        assert s.name == nm
        assert s.max_matrix_vector_multiplications == np.inf
        for prop in ("solution", "solution_residual", "solution_converged", "solution_time",
                     "solution_num_matrix_vector_multiplications"):
            assert getattr(s, prop) is None
        sig = inspect.signature(cls.solve)
        assert list(sig.parameters) == ["self", "A", "b", "x0", "convex_proj_op"]
    # (the reference's positional parameters first, in its order; then this engine's own keyword `pipeline`)
    assert list(inspect.signature(solvers.CCQPSolverPGD.__init__).parameters)[1:] == [
        "desired_residual_tol", "max_matrix_vector_multiplications", "step_size", "pipeline"]
    assert solvers.CCQPSolverPGD(1e-3).step_size == 0.01
    spg = solvers.CCQPSolverSPG(1e-3, 10, 7, 0.25, 0.02, 0.4)
    assert (spg.m, spg.t, spg.sigma1, spg.sigma2) == (7, 0.25, 0.02, 0.4)
    d = solvers.CCQPSolverSPG(1e-3)
    assert (d.m, d.t, d.sigma1, d.sigma2) == (5, 0.5, 0.01, 0.5)


def test_projection_surface_matches_reference():
    ops = [(ss.IdentityProjOp(4), "Identity"), (ss.LowerBoundProjOp(4), "Lower Bound"),
           (ss.UpperBoundProjOp(4), "Upper Bound"), (ss.BoxProjOp(4), "Box"),
           (ss.SphereProjOp(4), "Sphere"), (ss.ConeProjOp(4), "Cone")]
    for op, nm in ops:
        assert op.name == nm and op.embedded_dimension == 4 and op.dim == 4
    assert np.array_equal(ss.BoxProjOp(3).lower_bound, -np.ones(3))
    assert np.array_equal(ss.BoxProjOp(3).upper_bound, np.ones(3))
    assert ss.SphereProjOp(3).radius == 1 and ss.ConeProjOp(3).aspect_ratio == 1
    assert ss.SphereProjOp(3, 2.5).radius == 2.5 and ss.ConeProjOp(3, 0.5).aspect_ratio == 0.5
    dj = ss.DisjointProjOp(ss.BoxProjOp(5), ss.ConeProjOp(3), ss.DisjointProjOp(ss.IdentityProjOp(2),
                                                                               ss.SphereProjOp(3)))
    assert dj.name == "DisjointUnion" and dj.embedded_dimension == 13
    leaves = dj._leaves()
    assert [(type(o).__name__, off) for o, off in leaves] == [
        ("BoxProjOp", 0), ("ConeProjOp", 5), ("IdentityProjOp", 8), ("SphereProjOp", 10)]
    x = np.arange(3.0)
    assert ss.IdentityProjOp(3)(x) is x                      # same object, solution_spaces.py:125
    assert np.array_equal(ss.IdentityProjOp(3).normal_vector(x), np.zeros(3))
    import pytest
    with pytest.raises(NotImplementedError):
        ss.ConeProjOp(3).normal_vector(x)                    # solution_spaces.py:465


def test_row_partition_covers_rows_once():
    for n in (1, 7, 8, 64, 1000, 131072):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                r0, rows, per = operators.row_partition(n, world, r)
                assert rows >= 0 and per == -(-n // world)
                assert r0 == min(r * per, n)
                seen += list(range(r0, r0 + rows))
            assert seen == list(range(n))


def test_batch_partition_covers_problems_once():
    for nprob in (1, 5, 8, 1024):
        for world in (1, 2, 8):
            got = sorted(p for r in range(world) for p in operators.batch_partition(nprob, world, r))
            assert got == list(range(nprob))
    assert len(operators.batch_partition(1024, 8, 3)) == 128     # BASELINE config 5


def test_parameter_refresh_detects_in_place_edits_and_reassignment():
    """solution_spaces._refresh (host logic of the stale-bounds fix): the reference re-reads the public bound /
    radius attributes on every call (solution_spaces.py:200-201,276-277,363-366,430-433); the device copies are
    rebuilt when those were edited in place or reassigned -- and only then."""
    lb, ub = -np.ones(6), np.ones(6)
    P = ss.BoxProjOp(6, lb, ub)
    assert P._refresh() is False and P._refresh() is False        # first sight, then unchanged
    v0 = P._version
    lb[2] = -0.5                                                   # in place
    assert P._refresh() is True and P._version == v0 + 1
    assert P._refresh() is False
    P.upper_bound = np.ones(6)                                     # reassigned to an equal array: a new object
    assert P._refresh() is True
    P.upper_bound[0] = -0.0
    P._refresh()
    P.upper_bound[0] = 0.0                                         # sign of zero matters to the mask arithmetic
    assert P._refresh() is True
    Q = ss.LowerBoundProjOp(3, 2)                                  # scalar (int) bound
    assert Q._refresh() is False
    Q.lower_bound = 2.0                                            # int -> float changes `bound * mask`
    assert Q._refresh() is True
    S = ss.SphereProjOp(3, 1.0)
    assert S._refresh() is False
    S.radius = 1.0
    assert S._refresh() is False
    S.radius = 2.0
    assert S._refresh() is True
    D = ss.DisjointProjOp(P, S)
    assert D._refresh() is False and D._refresh() is False
    S.radius = 3.0
    S._refresh()                                                   # the child noticed on its own ...
    assert D._refresh() is True                                    # ... the parent still does (version counters)
    lb[0] = 0.25
    assert D._refresh() is True and D._refresh() is False


def test_host_callable_wrapper_surface():
    import pytest
    f = lambda x: np.clip(x, -1, 1)  # noqa: E731
    P = ss.HostCallableProjOp(4, f)
    assert P.embedded_dimension == 4 and isinstance(P, ss.ProjOpBase) and P.name == "<lambda>"
    x = np.array([-3.0, 0.5, 2.0, 1.0])
    assert np.array_equal(P(x), f(x))                              # numpy in: the callable itself, on the host
    with pytest.raises(AttributeError):
        P.normal_vector(x)
    with pytest.raises(TypeError):
        ss.HostCallableProjOp(4, 42)
