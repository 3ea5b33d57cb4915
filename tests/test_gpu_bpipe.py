"""GPU: solve_batch_pipelined (ccqp_bpipe_* / ccqp_bapgd_* in ccqp_bpipe.hip) advances many equal-size dense BBPGD
or APGD solves in lock-step with every scalar on the device.  It is an execution strategy only: per problem x, the residual, the
converged flag, the reported and the executed MV counts must be those of that problem's own solve(), bit for bit --
for every elementwise set, with one and several reduction blocks, odd sizes (unaligned rows), warm starts, problems
that finish at the start, mixed tolerances, and under an MV cap at every position."""
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
    assert float(a.solution_residual) == float(c.solution_residual)
    xa, xc = a.solution, c.solution
    xa = xa.cpu().numpy() if hasattr(xa, "cpu") else xa
    xc = xc.cpu().numpy() if hasattr(xc, "cpu") else xc
    assert np.array_equal(xa.view(np.uint64), xc.view(np.uint64))


def _sets(n, p):
    from ccqppy_amd import solution_spaces as ss
    lo, hi = -0.3 * np.ones(n) - 0.01 * (p % 5), 0.4 * np.ones(n) + 0.02 * (p % 3)
    return [ss.BoxProjOp(n, lo, hi), ss.LowerBoundProjOp(n, lo), ss.UpperBoundProjOp(n, hi), None,
            ss.BoxProjOp(n, -0.25, 0.5), ss.LowerBoundProjOp(n, 0)][p % 6]


def _problems(n, count, seed):
    from ccqppy_amd import problems
    out = []
    for p in range(count):
        op, b, _ = problems.random_problem(n, seed + p)
        out.append((op, b, None, _sets(n, p)))
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
    before = lockstep.stats["fallback_solves"]
    out = batch.solve_batch_pipelined(lambda: cls(next(it), max_mv), probs, **kw)
    # far from the matvec budget only a solve that does not converge at all may go back to the host loop
    assert max_mv < 5000 or lockstep.stats["fallback_solves"] - before <= sum(not c.solution_converged for c in out), \
        "a device-resident solve fell back to the host loop"
    return out


@pytest.mark.parametrize("n,count", [(37, 13), (256, 40), (1000, 12), (2049, 7), (4500, 5)])
def test_batched_pipeline_equals_individual_solves(n, count):
    probs = _problems(n, count, 700 + n)
    tols = [1e-8] * count
    ref = _reference(probs, tols, 5000)
    got = _batch(probs, tols, 5000)
    assert len(got) == count
    for a, c in zip(ref, got):
        assert c.solution_converged
        _same(a, c)
    assert len({r.solution_num_matrix_vector_multiplications for r in ref}) >= 1


def test_mixed_tolerances_warm_starts_and_trivial_starts():
    import torch
    n, count = 512, 24
    probs = _problems(n, count, 31)
    tols = [[1e-10, 1e-6, 1e-3, 1e30][p % 4] for p in range(count)]       # 1e30: satisfied by x0 already
    first = _reference(probs, [1e-9] * count, 5000)
    # warm start every third problem at (nearly) its solution
    for p in range(0, count, 3):
        A, b, _, P = probs[p]
        probs[p] = (A, b, first[p].solution.clone() if isinstance(first[p].solution, torch.Tensor)
                    else first[p].solution.copy(), P)
    ref = _reference(probs, tols, 5000)
    got = _batch(probs, tols, 5000)
    for a, c in zip(ref, got):
        _same(a, c)
    assert any(r.solution_num_matrix_vector_multiplications == 1 for r in ref)
    assert any(r.solution_num_matrix_vector_multiplications > 5 for r in ref)


def test_every_mv_cap():
    n, count = 300, 9
    probs = _problems(n, count, 5)
    tols = [1e-9] * count
    full = _reference(probs, tols, 5000)
    longest = max(r.solution_num_matrix_vector_multiplications for r in full)
    shortest = min(r.solution_num_matrix_vector_multiplications for r in full)
    assert longest > shortest >= 3                       # the cap separates finished from unfinished problems
    for cap in range(1, longest + 3):
        ref = _reference(probs, tols, cap)
        got = _batch(probs, tols, cap)
        for a, c in zip(ref, got):
            _same(a, c)


def test_numpy_problems_and_the_oracle():
    n, count = 192, 6
    probs, tols = [], [1e-8] * count
    for p in range(count):
        A = gen_ref.spd_rows(n, 40 + p)
        b = -A.dot(gen_ref.xstar(n, 40 + p))
        from ccqppy_amd import solution_spaces as ss
        probs.append((A, b, None, ss.BoxProjOp(n, -2.0 * np.ones(n), 2.0 * np.ones(n))))
    got = _batch(probs, tols, 5000, ahead=1)
    for (A, b, _, _), c in zip(probs, got):
        assert isinstance(c.solution, np.ndarray)
        o = cpu_ref.solve_bbpgd(A, b, 1e-8, 5000, P=cpu_ref.Box(n, -2.0 * np.ones(n), 2.0 * np.ones(n)))
        assert o.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
        assert np.linalg.norm(c.solution - o.solution) <= 1e-10 * np.linalg.norm(o.solution)
    ref = _reference(probs, tols, 5000)
    for a, c in zip(ref, got):
        _same(a, c)


def test_large_batch_and_back_to_back_batches():
    """300 problems (more than one wave of workgroups per launch), then a second, smaller batch on the same context:
    the progress mirror of the first must not leak into the second."""
    n = 128
    probs = _problems(n, 300, 9000)
    tols = [1e-8] * len(probs)
    got = _batch(probs, tols, 5000)
    ref = _reference(probs[::25], tols[::25], 5000)
    for a, c in zip(ref, got[::25]):
        _same(a, c)
    assert all(c.solution_converged for c in got)
    probs2 = _problems(n, 300, 9000)[:7]
    got2 = _batch(probs2, tols[:7], 5000)
    for a, c in zip(got[:7], got2):
        _same(a, c)


@pytest.mark.parametrize("n,count", [(37, 13), (256, 40), (1000, 12), (2049, 7), (4500, 5)])
def test_lockstep_apgd_equals_individual_solves(n, count):
    probs = _problems(n, count, 1700 + n)
    tols = [1e-8] * count
    ref = _reference(probs, tols, 5000, "APGD")
    got = _batch(probs, tols, 5000, "APGD")
    for a, c in zip(ref, got):
        assert c.solution_converged
        _same(a, c)
    assert max(r.solution_num_matrix_vector_multiplications for r in ref) > 20


def test_lockstep_apgd_mixed_tolerances_warm_starts_numpy_and_ahead():
    import torch
    n, count = 384, 18
    probs = _problems(n, count, 77)
    first = _reference(probs, [1e-9] * count, 5000, "APGD")
    for p in range(0, count, 3):                        # warm start every third problem at its solution
        A, b, _, P = probs[p]
        probs[p] = (A, b, first[p].solution.clone(), P)
    tols = [[1e-10, 1e-6, 1e-3, 1e30][p % 4] for p in range(count)]
    ref = _reference(probs, tols, 5000, "APGD")
    for ahead in (1, 4, 16):
        got = _batch(probs, tols, 5000, "APGD", ahead=ahead)
        for a, c in zip(ref, got):
            _same(a, c)
    assert min(r.solution_num_matrix_vector_multiplications for r in ref) <= 4
    # numpy in, numpy out, and the oracle
    from ccqppy_amd import solution_spaces as ss
    A = gen_ref.spd_rows(192, 8)
    b = -A.dot(gen_ref.xstar(192, 8))
    lo, hi = -2.0 * np.ones(192), 2.0 * np.ones(192)
    got = _batch([(A, b, None, ss.BoxProjOp(192, lo, hi))], [1e-8], 5000, "APGD")
    assert isinstance(got[0].solution, np.ndarray)
    o = cpu_ref.solve_apgd(A, b, 1e-8, 5000, P=cpu_ref.Box(192, lo, hi))
    assert o.solution_num_matrix_vector_multiplications == got[0].solution_num_matrix_vector_multiplications
    assert np.linalg.norm(got[0].solution - o.solution) <= 1e-10 * np.linalg.norm(o.solution)
    assert isinstance(first[0].solution, torch.Tensor)


def test_lockstep_apgd_under_a_matvec_budget():
    """Problems that converge inside the budget come from the lock-step run, the others from the reference's own exit
    path (plain solve()); either way the results are those of individual solves, at every budget."""
    n, count = 200, 8
    probs = _problems(n, count, 11)
    tols = [1e-9] * count
    full = _reference(probs, tols, 5000, "APGD")
    longest = max(r.solution_num_matrix_vector_multiplications for r in full)
    shortest = min(r.solution_num_matrix_vector_multiplications for r in full)
    assert longest > shortest
    for cap in list(range(3, 9)) + list(range(shortest - 2, longest + 3)):
        ref = _reference(probs, tols, cap, "APGD")
        got = _batch(probs, tols, cap, "APGD")
        for a, c in zip(ref, got):
            _same(a, c)


def test_single_solve_apgd_pipeline_option():
    """CCQPSolverAPGD(..., pipeline=True) = the lock-step scheduler with a batch of one; where it does not apply
    (anti-relaxation, cone, budget reached) the host-driven loop runs and the results are the same anyway."""
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    for n in (100, 1024, 3000):
        for p, (A, b, x0, P) in enumerate(_problems(n, 6, 40 + n)):
            for cap in (5000, 9, 30):
                from ccqppy_amd import lockstep
                before = lockstep.stats["fallback_solves"]
                with contextlib.redirect_stdout(io.StringIO()):
                    a = solvers.CCQPSolverAPGD(1e-8, cap).solve(A, b, x0=x0, convex_proj_op=P)
                    c = solvers.CCQPSolverAPGD(1e-8, cap, pipeline=True).solve(A, b, x0=x0, convex_proj_op=P)
                _same(a, c)
                assert cap < 5000 or lockstep.stats["fallback_solves"] == before   # really solved on the device
    op, b, _ = problems.random_problem(256, 3)
    with contextlib.redirect_stdout(io.StringIO()):
        a = solvers.CCQPSolverAPGDAntiRelaxation(1e-8, 5000).solve(op, b, convex_proj_op=ss.BoxProjOp(256, -0.5, 0.5))
        c = solvers.CCQPSolverAPGDAntiRelaxation(1e-8, 5000, pipeline=True).solve(
            op, b, convex_proj_op=ss.BoxProjOp(256, -0.5, 0.5))
        _same(a, c)
        a = solvers.CCQPSolverAPGD(1e-8, 64).solve(op, b, convex_proj_op=ss.ConeProjOp(256))
        c = solvers.CCQPSolverAPGD(1e-8, 64, pipeline=True).solve(op, b, convex_proj_op=ss.ConeProjOp(256))
        _same(a, c)


def test_batch_sharded_driver_with_the_pipelined_scheduler():
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd.batch import solve_batch_sharded
    n = 320

    def make(p):
        op, b, _ = problems.random_problem(n, 100 + p)
        return op, b, None, ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    for cls in (solvers.CCQPSolverBBPGD, solvers.CCQPSolverAPGD):
        res = solve_batch_sharded(lambda: cls(1e-8, 5000), make, 9, fused="pipelined", gather=True)
        assert sorted(res) == list(range(9))
        for p, r in res.items():
            with contextlib.redirect_stdout(io.StringIO()):
                a = cls(1e-8, 5000).solve(*make(p)[:2], convex_proj_op=make(p)[3])
            assert r.solution_converged
            assert r.solution_num_matrix_vector_multiplications == a.solution_num_matrix_vector_multiplications
            assert np.array_equal(r.solution.view(np.uint64), a.solution.cpu().numpy().view(np.uint64))


@pytest.mark.parametrize("n,count", [(37, 13), (64, 300), (256, 40), (513, 9), (1000, 12), (2048, 5)])
def test_one_workgroup_per_problem_equals_lock_step_and_individual_solves(n, count):
    """BBPGD's single-launch form (whole solve inside one workgroup): forced on and off, against solve(), with warm
    and trivial starts, mixed tolerances and every matvec budget."""
    probs = _problems(n, count, 3300 + n)
    tols = [[1e-8, 1e-5, 1e30][p % 3] for p in range(count)]
    ref = _reference(probs, tols, 5000)
    for single in (True, False):
        got = _batch(probs, tols, 5000, single_launch=single)
        for a, c in zip(ref, got):
            _same(a, c)
    if count > 20:
        return
    longest = max(r.solution_num_matrix_vector_multiplications for r in ref)
    for p in range(0, count, 2):
        A, b, _, P = probs[p]
        probs[p] = (A, b, ref[p].solution.clone(), P)
    for cap in (1, 2, 3, 5, longest - 1, longest, longest + 1, np.inf):
        ref2 = _reference(probs, tols, cap)
        got = _batch(probs, tols, cap, single_launch=True)
        for a, c in zip(ref2, got):
            _same(a, c)


@pytest.mark.parametrize("n,count", [(37, 13), (64, 200), (256, 30), (513, 9), (1000, 8), (2048, 4)])
def test_apgd_one_workgroup_per_problem_equals_lock_step_and_individual_solves(n, count):
    probs = _problems(n, count, 4300 + n)
    tols = [[1e-8, 1e-5, 1e30][p % 3] for p in range(count)]
    ref = _reference(probs, tols, 5000, "APGD")
    for single in (True, False):
        got = _batch(probs, tols, 5000, "APGD", single_launch=single)
        for a, c in zip(ref, got):
            _same(a, c)
    if count > 20:
        return
    longest = max(r.solution_num_matrix_vector_multiplications for r in ref)
    for p in range(0, count, 2):
        A, b, _, P = probs[p]
        probs[p] = (A, b, ref[p].solution.clone(), P)
    for cap in (3, 4, 5, 9, longest - 1, longest, longest + 1, np.inf):
        ref2 = _reference(probs, tols, cap, "APGD")
        got = _batch(probs, tols, cap, "APGD", single_launch=True)
        for a, c in zip(ref2, got):
            _same(a, c)


@pytest.mark.parametrize("solver", ["BBPGD", "APGD"])
def test_degenerate_sizes(solver):
    """n = 1, 2, 3, 5 (single rows, odd leading dimensions), both execution forms."""
    probs = []
    for i, n in enumerate((1, 2, 3, 5, 1, 2)):
        probs += _problems(n, 1, 800 + i)
    tols = [1e-9] * len(probs)
    ref = _reference(probs, tols, 5000, solver)
    for single in (True, False):
        got = _batch(probs, tols, 5000, solver, single_launch=single)
        for a, c in zip(ref, got):
            _same(a, c)


def test_pgd_constant_step_in_both_forms():
    """PGD = BBPGD's iteration with a constant step (per solver object), no A.g0 at the start."""
    from ccqppy_amd import batch, solvers
    for n, count in ((50, 9), (300, 7), (1500, 4)):
        probs = _problems(n, count, 6100 + n)
        steps = [0.3, 0.25, 0.35][:count] + [0.3] * max(0, count - 3)
        for cap in (5000, 1, 2, 7, 40):
            ref = []
            for (A, b, x0, P), st in zip(probs, steps):
                with contextlib.redirect_stdout(io.StringIO()):
                    ref.append(solvers.CCQPSolverPGD(1e-7, cap, st).solve(A, b, x0=x0, convex_proj_op=P))
            for single in (True, False):
                it = iter(steps)
                got = batch.solve_batch_pipelined(lambda: solvers.CCQPSolverPGD(1e-7, cap, next(it)), probs,
                                                  single_launch=single)
                for a, c in zip(ref, got):
                    _same(a, c)


@pytest.mark.parametrize("n,count", [(37, 13), (64, 120), (256, 20), (1000, 8), (2049, 5)])
def test_anti_relaxation_apgd_in_both_forms(n, count):
    """CCQPSolverAPGDAntiRelaxation: momentum restarts where gk.(x_{k+1} - x_k) > 0, best iterate returned."""
    probs = _problems(n, count, 7300 + n)
    tols = [[1e-8, 1e-5, 1e30][p % 3] for p in range(count)]
    ref = _reference(probs, tols, 5000, "APGDAntiRelaxation")
    for single in (True, False):
        got = _batch(probs, tols, 5000, "APGDAntiRelaxation", single_launch=single)
        for a, c in zip(ref, got):
            _same(a, c)
    if count > 20:
        return
    plain = _reference(probs, tols, 5000, "APGD")
    assert any(a.solution_num_matrix_vector_multiplications != c.solution_num_matrix_vector_multiplications
               for a, c in zip(ref, plain))              # the restarts do change the iteration
    longest = max(r.solution_num_matrix_vector_multiplications for r in ref)
    for p in range(0, count, 2):
        A, b, _, P = probs[p]
        probs[p] = (A, b, ref[p].solution.clone(), P)
    for cap in (3, 4, 5, 9, longest - 1, longest, longest + 1):
        ref2 = _reference(probs, tols, cap, "APGDAntiRelaxation")
        got = _batch(probs, tols, cap, "APGDAntiRelaxation", single_launch=(cap % 2 == 0))
        for a, c in zip(ref2, got):
            _same(a, c)


@pytest.mark.parametrize("n,count", [(37, 9), (120, 40), (700, 8), (2049, 4)])
def test_bbpgdf_fallback_variant_in_both_forms(n, count):
    """CCQPSolverBBPGDf: best iterate kept, restart from it when the step length collapses.  On convex problems the
    restart never fires; INDEFINITE matrices (A - shift I inside a box) make the BB step negative in most iterations,
    so the fall-back branch runs hundreds of times -- still bit for bit the host loop, at every budget."""
    import torch
    from ccqppy_amd.operators import DenseOperator
    probs = _problems(n, count, 8300 + n)
    for shift in (0.0, 1.8, 2.2):
        ps = []
        for p, (op, b, x0, P) in enumerate(probs):
            A = op.A - shift * torch.eye(n, dtype=torch.float64, device=op.A.device)
            ps.append((DenseOperator(A), b, x0, P if (shift == 0.0 or P is not None) else _sets(n, 0)))
        tols = [1e-8] * count
        for cap in ((5000,) if shift == 0.0 else (60, 7, 150)):
            ref = _reference(ps, tols, cap, "BBPGDf")
            for single in (True, False):
                got = _batch(ps, tols, cap, "BBPGDf", single_launch=single)
                for a, c in zip(ref, got):
                    _same(a, c)
            if shift == 2.2 and cap == 150:
                assert any(not r.solution_converged for r in ref)


def test_single_solve_bbpgdf_pipeline_option():
    """CCQPSolverBBPGDf(..., pipeline=True) = the lock-step scheduler with a batch of one (n <= 2048)."""
    import torch
    from ccqppy_amd import lockstep, solvers
    from ccqppy_amd.operators import DenseOperator
    for n in (90, 700, 3000):
        for p, (op, b, x0, P) in enumerate(_problems(n, 4, 90 + n)):
            for shift in (0.0, 2.2):
                A = DenseOperator(op.A - shift * torch.eye(n, dtype=torch.float64, device=op.A.device))
                P2 = P if P is not None else _sets(n, 0)
                for cap in ((5000, 9) if shift == 0.0 else (80, 9)):
                    before = lockstep.stats["fallback_solves"]
                    with contextlib.redirect_stdout(io.StringIO()):
                        a = solvers.CCQPSolverBBPGDf(1e-8, cap).solve(A, b, x0=x0, convex_proj_op=P2)
                        c = solvers.CCQPSolverBBPGDf(1e-8, cap, pipeline=True).solve(A, b, x0=x0, convex_proj_op=P2)
                    _same(a, c)
                    assert lockstep.stats["fallback_solves"] == before       # (n = 3000: the host loop, by the gate)


def test_single_solve_pgd_pipeline_option():
    """CCQPSolverPGD(..., pipeline=True) -- and the default None for a budget of at least 1000 -- = the lock-step scheduler
    with a batch of one (n <= 2048; n <= 64: one launch) against the host-driven loop pipeline=False: x bit for bit,
    counts, residual, converged flag; budgets that end the solve early take the reference's exit paths."""
    from ccqppy_amd import lockstep, solvers
    for n in (50, 301, 1024, 2048, 3000):
        for p, (op, b, x0, P) in enumerate(_problems(n, 3, 7300 + n)):
            P2 = P if P is not None else _sets(n, 0)
            for cap in (np.inf, 5000, 1, 2, 7, 60):
                before = lockstep.stats["fallback_solves"]
                with contextlib.redirect_stdout(io.StringIO()):
                    a = solvers.CCQPSolverPGD(1e-6, cap, 0.3, pipeline=False).solve(op, b, x0=x0, convex_proj_op=P2)
                    c = solvers.CCQPSolverPGD(1e-6, cap, 0.3, pipeline=True).solve(op, b, x0=x0, convex_proj_op=P2)
                    d = solvers.CCQPSolverPGD(1e-6, cap, 0.3).solve(op, b, x0=x0, convex_proj_op=P2)
                _same(a, c)
                _same(a, d)
                assert lockstep.stats["fallback_solves"] == before
            assert a.solution_num_matrix_vector_multiplications > 7


def test_mixed_sizes_are_grouped():
    probs = []
    for i, n in enumerate((96, 200, 96, 333, 200, 96)):
        probs += _problems(n, 1, 600 + i)
    for solver in ("BBPGD", "APGD"):
        ref = _reference(probs, [1e-8] * len(probs), 5000, solver)
        got = _batch(probs, [1e-8] * len(probs), 5000, solver)
        for a, c in zip(ref, got):
            _same(a, c)


def test_rejects_what_it_cannot_run():
    from ccqppy_amd import batch, problems, solvers, solution_spaces as ss
    op, b, _ = problems.random_problem(64, 1)
    op2, b2, _ = problems.random_problem(96, 2)
    with pytest.raises(TypeError):
        batch.solve_batch_pipelined(lambda: solvers.CCQPSolverSPG(1e-8), [(op, b, None, None)])
    with pytest.raises(TypeError):
        batch.solve_batch_pipelined(lambda: solvers.CCQPSolverAPGD(1e-8, reuse_matvec=True), [(op, b, None, None)])
    # (every solver runs on Sphere / Cone / DisjointUnion sets in the segmented lock-step forms -- tests/test_gpu_bsegs.py;
    # MPRGP needs a normal vector for every block: a Cone has none, solution_spaces.py:465)
    with pytest.raises(TypeError):
        batch.solve_batch_pipelined(lambda: solvers.CCQPSolverMPRGP(1e-6, 300), [(op, b, None, ss.ConeProjOp(64))])
    with pytest.raises(TypeError):
        batch.solve_batch_pipelined(lambda: solvers.CCQPSolverMPRGP(1e-6, 300),
                                    [(op, b, None, ss.DisjointProjOp(ss.SphereProjOp(32), ss.ConeProjOp(32)))])
    class Foreign:                        # a user-defined block: no device form, the union cannot run in lock-step
        embedded_dimension = 32

        def __call__(self, x):
            return x

    with pytest.raises(TypeError):
        batch.solve_batch_pipelined(lambda: solvers.CCQPSolverBBPGD(1e-8),
                                    [(op, b, None, ss.DisjointProjOp(ss.BoxProjOp(32), Foreign()))])
    with pytest.raises(ValueError):
        batch.solve_batch_pipelined(lambda: solvers.CCQPSolverBBPGD(1e-8), [(op, b, None, None), (op2, b, None, None)])
    assert batch.solve_batch_pipelined(lambda: solvers.CCQPSolverBBPGD(1e-8), []) == []
