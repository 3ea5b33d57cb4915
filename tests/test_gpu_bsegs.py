"""GPU: the segmented lock-step scheduler (ccqp_bsegs_* in ccqp_bsegs.hip; batch.solve_batch_pipelined with BBPGD / PGD
on Sphere, Cone and DisjointUnion sets).  An execution strategy only: per problem x, the residual, the converged
flag, the reported and the executed MV counts must be those of that problem's own solve(), bit for bit -- for sets
with small blocks only (no norm launch), with large blocks (norm launches), whole-vector Sphere / Cone, several
reduction blocks, different sets inside one batch, warm and trivial starts, mixed tolerances, every MV cap.
Reference loops: solvers.py:583-669, 94-170 around solution_spaces.py:540-560, 417-435, 470-492."""
import contextlib
import io

import numpy as np
import pytest

from oracle import cpu_ref, gen_ref

pytestmark = pytest.mark.gpu


def _same(a, c):
    assert a.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
    assert a.executed_matrix_vector_multiplications == c.executed_matrix_vector_multiplications
    assert bool(a.solution_converged) == bool(c.solution_converged)
    assert float(a.solution_residual) == float(c.solution_residual) or \
        (np.isnan(a.solution_residual) and np.isnan(c.solution_residual))
    xa, xc = a.solution, c.solution
    xa = xa.cpu().numpy() if hasattr(xa, "cpu") else xa
    xc = xc.cpu().numpy() if hasattr(xc, "cpu") else xc
    assert np.array_equal(xa.view(np.uint64), xc.view(np.uint64))


def _set(n, p, family):
    """Feasible set number p of a family for vectors of length n."""
    from ccqppy_amd import solution_spaces as ss
    if family == "sphere":
        return ss.SphereProjOp(n, [0.5, 2.0, 40.0][p % 3])
    if family == "cone":
        return ss.ConeProjOp(n, [1, 0.5, 3.0][p % 3])
    if family == "spheres3":                      # the ensemble's friction-like family: many tiny blocks
        k = n // 3
        ops = [ss.SphereProjOp(3, 0.2 + 0.1 * (p % 4))] * k
        if n - 3 * k:
            ops.append(ss.IdentityProjOp(n - 3 * k))
        return ss.DisjointProjOp(*ops)
    if family == "mixed_nocone":                  # as "mixed", every block with a normal vector (MPRGP's sets)
        half = n // 2
        q = (n - half) // 3
        lo, hi = -0.3 * np.ones(half) - 0.01 * p, 0.4 * np.ones(half)
        return ss.DisjointProjOp(ss.BoxProjOp(half, lo, hi),
                                 ss.DisjointProjOp(ss.SphereProjOp(q, 0.7), ss.LowerBoundProjOp(q, -0.2 * np.ones(q))),
                                 ss.SphereProjOp(n - half - 2 * q, 0.3 + 0.05 * p))
    if family == "mixed":                         # small and large blocks, elementwise and norm-based, nested
        half = n // 2
        q = (n - half) // 3
        lo, hi = -0.3 * np.ones(half) - 0.01 * p, 0.4 * np.ones(half)
        return ss.DisjointProjOp(ss.BoxProjOp(half, lo, hi),
                                 ss.DisjointProjOp(ss.SphereProjOp(q, 0.7), ss.ConeProjOp(q, 2.0)),
                                 ss.SphereProjOp(n - half - 2 * q, 0.3 + 0.05 * p))
    raise KeyError(family)


def _problems(n, count, seed, family):
    from ccqppy_amd import problems
    out = []
    for p in range(count):
        op, b, _ = problems.random_problem(n, seed + p)
        fam = family if isinstance(family, str) else family[p % len(family)]
        out.append((op, b, None, _set(n, p, fam)))
    return out


def _reference(probs, tols, max_mv, solver="BBPGD"):
    from ccqppy_amd import solvers
    cls = getattr(solvers, "CCQPSolver" + solver)
    ref = []
    for (A, b, x0, P), tol in zip(probs, tols):
        with contextlib.redirect_stdout(io.StringIO()):
            ref.append(cls(tol, max_mv).solve(A, b, x0=x0, convex_proj_op=P))
    return ref


def _batch(probs, tols, max_mv, solver="BBPGD", **kw):
    from ccqppy_amd import batch, lockstep, solvers
    cls = getattr(solvers, "CCQPSolver" + solver)
    it = iter(tols)
    before = dict(lockstep.stats)
    out = batch.solve_batch_pipelined(lambda: cls(next(it), max_mv), probs, **kw)
    assert lockstep.stats == before, "a device-resident solve fell back to the host loop"
    return out


@pytest.mark.parametrize("family", ["sphere", "spheres3", "mixed"])
@pytest.mark.parametrize("n,count", [(36, 11), (256, 24), (1000, 9), (2049, 6), (4500, 4)])
def test_segmented_lockstep_equals_individual_solves(family, n, count):
    probs = _problems(n, count, 900 + n, family)
    tols = [1e-8] * count
    # "mixed" holds a Cone block, which is not a projection in the reference (:439): such solves run to their budget
    budget = 90 if family == "mixed" else 4000
    ref = _reference(probs, tols, budget)
    got = _batch(probs, tols, budget)
    assert len(got) == count
    for a, c in zip(ref, got):
        assert c.solution_converged or family == "mixed"
        _same(a, c)
    assert max(r.solution_num_matrix_vector_multiplications for r in ref) > 5


def test_cone_runs_to_its_budget_like_the_reference():
    """The reference's Cone is not a projection (solution_spaces.py:439,484): no solve converges; the budget exit path
    (solvers.py:641-644) and every iterate must still be those of solve()."""
    n, count = 300, 6
    probs = _problems(n, count, 17, "cone")
    for cap in (1, 2, 3, 7, 40):
        ref = _reference(probs, [1e-8] * count, cap)
        got = _batch(probs, [1e-8] * count, cap)
        for a, c in zip(ref, got):
            assert not c.solution_converged or cap > 40
            _same(a, c)


def test_different_sets_in_one_batch_and_sharded_entry():
    """Problems of one size with DIFFERENT sets (different work-list lengths; some without any norm block) advance in
    one lock-step run; `solve_batch_sharded` picks this scheduler by itself."""
    from ccqppy_amd import batch, lockstep, solvers
    n, count = 600, 12
    probs = _problems(n, count, 3, ["sphere", "spheres3", "mixed", "cone"])
    tols = [1e-7] * count
    ref = _reference(probs, tols, 60)
    got = _batch(probs, tols, 60)
    for a, c in zip(ref, got):
        _same(a, c)
    before = dict(lockstep.stats)
    sharded = batch.solve_batch_sharded(lambda: solvers.CCQPSolverBBPGD(1e-7, 60), lambda p: probs[p], count)
    assert lockstep.stats == before                       # neither declined nor fell back
    for p in range(count):
        _same(ref[p], sharded[p])


def test_mixed_tolerances_warm_and_trivial_starts_numpy_inputs_and_oracle():
    import torch
    n, count = 384, 15
    probs = _problems(n, count, 7, ["spheres3", "mixed", "sphere"])
    first = _reference(probs, [1e-9] * count, 300)
    for p in range(0, count, 3):                          # warm start every third problem at (nearly) its solution
        A, b, _, P = probs[p]
        probs[p] = (A, b, first[p].solution.clone(), P)
    tols = [[1e-10, 1e-6, 1e-3, 1e30][p % 4] for p in range(count)]       # 1e30: satisfied by x0 already
    ref = _reference(probs, tols, 300)
    got = _batch(probs, tols, 300, ahead=1)
    for a, c in zip(ref, got):
        _same(a, c)
    assert any(r.solution_num_matrix_vector_multiplications == 1 for r in ref)
    # numpy inputs, and the ORACLE as the yardstick: same MV count, x within 1e-10
    from ccqppy_amd import solution_spaces as ss
    A = gen_ref.spd_rows(192, 8)
    b = -A.dot(gen_ref.xstar(192, 8))
    P = ss.DisjointProjOp(*([ss.SphereProjOp(3, 0.25)] * 40 + [ss.SphereProjOp(72, 1.5)]))
    Po = cpu_ref.Disjoint(*([cpu_ref.Sphere(3, 0.25)] * 40 + [cpu_ref.Sphere(72, 1.5)]))
    got = _batch([(A, b, None, P)], [1e-8], 4000)
    assert isinstance(got[0].solution, np.ndarray) and not isinstance(got[0].solution, torch.Tensor)
    o = cpu_ref.solve_bbpgd(A, b, 1e-8, 4000, P=Po)
    assert o.solution_num_matrix_vector_multiplications == got[0].solution_num_matrix_vector_multiplications
    assert o.executed_mv == got[0].executed_matrix_vector_multiplications
    assert np.linalg.norm(got[0].solution - o.solution) <= 1e-10 * np.linalg.norm(o.solution)


def test_every_mv_cap():
    n, count = 300, 8
    probs = _problems(n, count, 5, ["spheres3", "sphere"])
    tols = [1e-9] * count
    full = _reference(probs, tols, 4000)
    longest = max(r.solution_num_matrix_vector_multiplications for r in full)
    shortest = min(r.solution_num_matrix_vector_multiplications for r in full)
    assert longest >= shortest >= 3
    for cap in range(1, longest + 3):
        ref = _reference(probs, tols, cap)
        got = _batch(probs, tols, cap)
        for a, c in zip(ref, got):
            _same(a, c)


def test_pgd_constant_step_on_general_sets():
    from ccqppy_amd import batch, lockstep, solvers
    n, count = 200, 10
    probs = _problems(n, count, 23, ["sphere", "spheres3"])
    ref, it = [], iter(range(count))
    for (A, b, x0, P) in probs:
        with contextlib.redirect_stdout(io.StringIO()):
            ref.append(solvers.CCQPSolverPGD(1e-7, 600, 0.2 + 0.01 * next(it)).solve(A, b, convex_proj_op=P))
    it = iter(range(count))
    before = dict(lockstep.stats)
    got = batch.solve_batch_pipelined(lambda: solvers.CCQPSolverPGD(1e-7, 600, 0.2 + 0.01 * next(it)), probs)
    assert lockstep.stats == before
    for a, c in zip(ref, got):
        assert c.solution_converged
        _same(a, c)


@pytest.mark.parametrize("solver", ["APGD", "APGDAntiRelaxation"])
@pytest.mark.parametrize("family", ["sphere", "spheres3"])
@pytest.mark.parametrize("n,count", [(36, 11), (256, 16), (1000, 7), (2049, 5)])
def test_segmented_lockstep_apgd_equals_individual_solves(solver, family, n, count):
    """APGD and its anti-relaxation variant (solvers.py:220-343, 393-533): the state machine of ccqp_bapgd_* with the
    residual and the trial step taken from the segmented engine (ccqp_bsegs_apgd_round_f64)."""
    probs = _problems(n, count, 1300 + n, family)
    tols = [1e-8] * count
    ref = _reference(probs, tols, 4000, solver)
    got = _batch(probs, tols, 4000, solver)
    for a, c in zip(ref, got):
        assert c.solution_converged
        _same(a, c)
    assert max(r.solution_num_matrix_vector_multiplications for r in ref) > 10
    for ahead in (1, 16):
        for a, c in zip(ref, _batch(probs, tols, 4000, solver, ahead=ahead)):
            _same(a, c)


def test_segmented_apgd_mixed_sets_tolerances_starts_and_budgets():
    import torch
    from ccqppy_amd import lockstep
    n, count = 300, 12
    probs = _problems(n, count, 77, ["spheres3", "sphere"])
    first = _reference(probs, [1e-9] * count, 4000, "APGD")
    for p in range(0, count, 3):
        A, b, _, P = probs[p]
        probs[p] = (A, b, first[p].solution.clone(), P)
    tols = [[1e-9, 1e-6, 1e-3][p % 3] for p in range(count)]
    ref = _reference(probs, tols, 4000, "APGD")
    from ccqppy_amd import batch, solvers
    before = lockstep.stats["fallback_solves"]
    it0 = iter(tols)
    got = batch.solve_batch_pipelined(lambda: solvers.CCQPSolverAPGD(next(it0), 4000), probs)
    # only a solve that reaches the matvec budget goes back to the host loop (the reference's own exit path)
    assert lockstep.stats["fallback_solves"] - before <= sum(not r.solution_converged for r in ref)
    for a, c in zip(ref, got):
        _same(a, c)
    assert isinstance(got[0].solution, torch.Tensor)
    # under a budget some problems take the reference's own exit path on the host (counted, by design); results equal
    longest = max(r.solution_num_matrix_vector_multiplications for r in ref)
    for cap in (3, 5, longest // 2, longest - 1, longest, longest + 2):
        want = _reference(probs, tols, cap, "APGD")
        it = iter(tols)
        have = batch.solve_batch_pipelined(lambda: solvers.CCQPSolverAPGD(next(it), cap), probs)
        for a, c in zip(want, have):
            _same(a, c)
    assert lockstep.stats["fallback_solves"] >= 0


def test_what_has_no_lockstep_form_is_declined_not_mis_solved():
    """SPG without private generators, and MPRGP on a set with a block whose normal_vector does not run on the device (a
    Cone block: the reference raises NotImplementedError, solution_spaces.py:465; scalar bounds: a TypeError): the
    argument check says so, and solve_batch_sharded goes to the host-driven scheduler (counted in lockstep.stats),
    which raises what the reference raises."""
    from ccqppy_amd import batch, lockstep, solvers, solution_spaces as ss
    probs = _problems(128, 3, 1, "spheres3")
    with pytest.raises(TypeError):
        batch.solve_batch_pipelined(lambda: solvers.CCQPSolverSPG(1e-6, 300), probs)
    cone = _problems(128, 3, 1, "mixed")
    with pytest.raises(TypeError):
        batch.solve_batch_pipelined(lambda: solvers.CCQPSolverMPRGP(1e-6, 300), cone)
    before = lockstep.stats["auto_declined"]
    with pytest.raises(NotImplementedError):
        batch.solve_batch_sharded(lambda: solvers.CCQPSolverMPRGP(1e-6, 300), lambda p: cone[p], 3)
    assert lockstep.stats["auto_declined"] == before + 1
    scalar = [(A, b, x0, ss.DisjointProjOp(ss.BoxProjOp(64, -0.5, 0.5), ss.SphereProjOp(64, 1.0))) for A, b, x0, _ in probs]
    with pytest.raises(TypeError):
        batch.solve_batch_pipelined(lambda: solvers.CCQPSolverMPRGP(1e-6, 300), scalar)


@pytest.mark.parametrize("family", ["sphere", "spheres3", "mixed_nocone"])
@pytest.mark.parametrize("n,count", [(36, 11), (256, 24), (1000, 9), (2049, 6), (4500, 4)])
def test_segmented_lockstep_mprgp_equals_individual_solves(family, n, count):
    """MPRGP in lock-step on Sphere / DisjointUnion sets (ccqp_bsegs_mprgp_*): four projections, the normal vector, the
    feasible-step search and the residual of every iteration through the segmented engine.  Per problem the bits of its
    own solve(): reported and executed matvecs, converged flag, residual, x.  Reference loop: solvers.py:1026-1200 around
    solution_spaces.py:389-403, 417-435, 512-525, 540-560."""
    probs = _problems(n, count, 1900 + n, family)
    tols = [1e-6] * count
    ref = _reference(probs, tols, 3000, "MPRGP")
    got = _batch(probs, tols, 3000, "MPRGP")
    assert len(got) == count
    for a, c in zip(ref, got):
        assert c.solution_converged
        _same(a, c)
    assert max(r.solution_num_matrix_vector_multiplications for r in ref) > 5


def test_segmented_lockstep_mprgp_budgets_starts_and_the_sharded_entry():
    """Different sets in one batch, warm and trivial starts, mixed tolerances, every kind of matvec budget (a problem that
    would reach it takes the reference's own exit path on the host, counted), and `solve_batch_sharded` picking the
    lock-step scheduler by itself (no declined batch any more)."""
    import torch
    from ccqppy_amd import batch, lockstep, solvers
    n, count = 384, 12
    probs = _problems(n, count, 77, ["spheres3", "mixed_nocone", "sphere"])
    first = _reference(probs, [1e-7] * count, 3000, "MPRGP")
    for p in range(0, count, 3):
        A, b, _, P = probs[p]
        probs[p] = (A, b, first[p].solution.clone(), P)
    tols = [[1e-7, 1e-5, 1e-3, 1e30][p % 4] for p in range(count)]
    ref = _reference(probs, tols, 3000, "MPRGP")
    got = _batch(probs, tols, 3000, "MPRGP", ahead=1)
    for a, c in zip(ref, got):
        _same(a, c)
    assert any(r.solution_num_matrix_vector_multiplications == 1 for r in ref)       # x0 already meets tol
    longest = max(r.solution_num_matrix_vector_multiplications for r in ref)
    for cap in (3, 4, 5, longest // 2, longest, longest + 3):
        want = _reference(probs, tols, cap, "MPRGP")
        it = iter(tols)
        have = batch.solve_batch_pipelined(lambda: solvers.CCQPSolverMPRGP(next(it), cap), probs)
        for a, c in zip(want, have):
            _same(a, c)
    before = dict(lockstep.stats)
    sharded = batch.solve_batch_sharded(lambda: solvers.CCQPSolverMPRGP(1e-6, 3000), lambda p: probs[p], count)
    assert lockstep.stats == before
    want = _reference(probs, [1e-6] * count, 3000, "MPRGP")
    for p in range(count):
        _same(want[p], sharded[p])
    assert isinstance(got[0].solution, torch.Tensor)


@pytest.mark.parametrize("solver", ["BBPGD", "APGD", "APGDAntiRelaxation"])
@pytest.mark.parametrize("family", ["sphere", "spheres3", "mixed"])
@pytest.mark.parametrize("n", [90, 700, 2048])
def test_single_solve_pipeline_option_on_general_sets(solver, family, n):
    """`pipeline=True` on a Sphere / DisjointUnion set: the solve runs as a segmented lock-step batch of one (dense
    operator, n <= 2048); beyond that size, and for a solve that reaches its budget under APGD, the host-driven loop.
    Same bits either way."""
    from ccqppy_amd import lockstep, problems, solvers
    op, b, _ = problems.random_problem(n, 4000 + n)
    P = _set(n, 1, family)
    cls = getattr(solvers, "CCQPSolver" + solver)
    budget = 70 if family == "mixed" else 4000            # (a Cone block inside: runs to the budget)
    before = lockstep.stats["fallback_solves"]
    with contextlib.redirect_stdout(io.StringIO()):
        a = cls(1e-8, budget).solve(op, b, convex_proj_op=P)
        c = cls(1e-8, budget, pipeline=True).solve(op, b, convex_proj_op=P)
    _same(a, c)
    if family != "mixed":
        assert c.solution_converged and lockstep.stats["fallback_solves"] == before
    big = problems.random_problem(2100, 1)
    with contextlib.redirect_stdout(io.StringIO()):
        a = cls(1e-8, 300).solve(big[0], big[1], convex_proj_op=_set(2100, 0, "sphere"))
        c = cls(1e-8, 300, pipeline=True).solve(big[0], big[1], convex_proj_op=_set(2100, 0, "sphere"))
    _same(a, c)


@pytest.mark.parametrize("n,count", [(37, 9), (120, 30), (700, 8), (2049, 4)])
def test_segmented_bbpgdf_fallback_variant(n, count):
    """BBPGDf (solvers.py:719-819) on Sphere / DisjointUnion sets.  On convex problems the restart from the best iterate
    never fires, so -- as for the elementwise form -- the matrices are made INDEFINITE (A - 2.2 I inside bounded
    sets), where the step length collapses again and again; budgets keep the solves finite."""
    import torch
    from ccqppy_amd import problems
    probs = []
    for p in range(count):
        op, b, _ = problems.random_problem(n, 5200 + n + p)
        A = op.A - 2.2 * torch.eye(n, dtype=torch.float64, device=op.A.device)
        probs.append((A, b, None, _set(n, p, ["sphere", "spheres3"][p % 2])))
    for budget in (35, 160):
        ref = _reference(probs, [1e-9] * count, budget, "BBPGDf")
        got = _batch(probs, [1e-9] * count, budget, "BBPGDf")
        for a, c in zip(ref, got):
            _same(a, c)
    # the restart branch must have been exercised: BBPGD and BBPGDf differ on these problems
    plain = _reference(probs, [1e-9] * count, 160, "BBPGD")
    assert any(not np.array_equal(a.solution.cpu().numpy(), c.solution.cpu().numpy()) for a, c in zip(plain, ref))
    convex = _problems(n, min(count, 6), 61 + n, ["sphere", "spheres3"])
    ref = _reference(convex, [1e-8] * len(convex), 4000, "BBPGDf")
    got = _batch(convex, [1e-8] * len(convex), 4000, "BBPGDf")
    for a, c in zip(ref, got):
        assert c.solution_converged
        _same(a, c)
