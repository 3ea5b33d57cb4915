"""GPU: the pipelined BBPGD iteration (CCQPSolverBBPGD(..., pipeline=True); ccqp_pipe_* in ccqp_pipe.hip) keeps the
step length and the exit test on the device and queues iterations ahead.  It is an execution strategy only: x,
the residual, the converged flag, the reported and the executed MV counts must be those of the host-driven loop,
bit for bit -- for every elementwise set, at reduction sizes with one and several blocks, under an MV cap at
every position, and when the very first residual already meets the tolerance."""
import contextlib
import io

import numpy as np
import pytest

from oracle import cpu_ref, gen_ref

pytestmark = pytest.mark.gpu


def _pair(op, b, P, tol, max_mv, x0=None):
    from ccqppy_amd import solvers
    out = []
    for pipe in (False, True):
        with contextlib.redirect_stdout(io.StringIO()):
            out.append(solvers.CCQPSolverBBPGD(tol, max_mv, pipeline=pipe).solve(op, b, x0=x0, convex_proj_op=P))
    return out


def _same(a, c):
    assert a.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
    assert a.executed_matrix_vector_multiplications == c.executed_matrix_vector_multiplications
    assert bool(a.solution_converged) == bool(c.solution_converged)
    assert float(a.solution_residual) == float(c.solution_residual)
    xa, xc = a.solution, c.solution
    xa = xa.cpu().numpy() if hasattr(xa, "cpu") else xa
    xc = xc.cpu().numpy() if hasattr(xc, "cpu") else xc
    assert np.array_equal(xa.view(np.uint64), xc.view(np.uint64))


@pytest.mark.parametrize("n", [64, 1024, 2048, 2049, 8192, 20000])
@pytest.mark.parametrize("kind", ["Box", "Lower", "Upper", "Identity"])
def test_pipelined_bbpgd_equals_host_driven_loop(n, kind):
    from ccqppy_amd import problems, solution_spaces as ss
    op, b, _ = problems.random_problem(n, 4321)
    lo, hi = -0.3 * np.ones(n), 0.4 * np.ones(n)
    P = {"Box": ss.BoxProjOp(n, lo, hi), "Lower": ss.LowerBoundProjOp(n, lo), "Upper": ss.UpperBoundProjOp(n, hi),
         "Identity": ss.IdentityProjOp(n)}[kind]
    a, c = _pair(op, b, P, 1e-8, 5000)
    assert c.solution_converged
    _same(a, c)


def test_pipelined_bbpgd_under_every_mv_cap_and_trivial_start():
    from ccqppy_amd import problems, solution_spaces as ss
    n = 4096
    op, b, _ = problems.random_problem(n, 99)
    P = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    full, _ = _pair(op, b, P, 1e-10, 5000)
    total = full.solution_num_matrix_vector_multiplications
    assert total >= 8
    for cap in range(1, total + 3):                          # cap before, at and after convergence
        a, c = _pair(op, b, P, 1e-10, cap)
        _same(a, c)
    # numpy in, numpy out; warm start at the solution: the loop is never entered
    A = gen_ref.spd_rows(512, 5)
    bb = -A.dot(gen_ref.xstar(512, 5))
    Pn = ss.BoxProjOp(512, -2.0 * np.ones(512), 2.0 * np.ones(512))
    a, c = _pair(A, bb, Pn, 1e-8, 5000)
    _same(a, c)
    assert isinstance(c.solution, np.ndarray)
    a2, c2 = _pair(A, bb, Pn, 1e-6, 5000, x0=c.solution)
    _same(a2, c2)
    assert c2.solution_num_matrix_vector_multiplications == 1
    # and against the oracle
    o = cpu_ref.solve_bbpgd(A, bb, 1e-8, 5000, P=cpu_ref.Box(512, -2.0 * np.ones(512), 2.0 * np.ones(512)))
    assert o.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
    assert np.linalg.norm(c.solution - o.solution) <= 1e-10 * np.linalg.norm(o.solution)


def test_graph_replay_of_three_iterations_gives_the_same_bits():
    """pipeline="graph": three captured iterations replayed as one hipGraph launch (plus single iterations near
    an MV cap) -- same results again, also when torch's current stream is the default stream."""
    import torch
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    n = 2048
    op, b, _ = problems.random_problem(n, 31)
    P = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    for cap in (5000, 11, 7, 6, 5, 4, 3, 2):
        with contextlib.redirect_stdout(io.StringIO()):
            ref = solvers.CCQPSolverBBPGD(1e-10, cap).solve(op, b, convex_proj_op=P)
            got = solvers.CCQPSolverBBPGD(1e-10, cap, pipeline="graph").solve(op, b, convex_proj_op=P)
            with torch.cuda.stream(torch.cuda.Stream()):
                got2 = solvers.CCQPSolverBBPGD(1e-10, cap, pipeline="graph").solve(op, b, convex_proj_op=P)
        _same(ref, got)
        _same(ref, got2)


def test_pipelined_bbpgd_on_sparse_and_matrix_free_operators():
    """The CSR matvec (all its window modes, incl. the wave-per-row launch) and the factors of a composed operator
    sit behind the same "converged" word."""
    import scipy.sparse as sp
    from ccqppy_amd import problems, solution_spaces as ss
    from ccqppy_amd.operators import ComposedOperator, DiagonalOperator, SparseOperator
    for n, k, pattern in ((4096, 8, "hashed"), (32768, 26, "stencil27"), (65536, 32, "hashed")):
        op, b, _ = problems.random_sparse_problem(n, k, 1234, pattern=pattern)
        a, c = _pair(op, b, ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n)), 1e-8, 5000)
        assert c.solution_converged
        _same(a, c)
    irr = gen_ref.irregular_spd(3000, 4).tolil()
    irr[5, :] = 0.001 * np.random.default_rng(1).normal(size=3000)       # one row on the wave-per-row path
    irr[:, 5] = irr[5, :].T
    irr[5, 5] = 50.0
    irr = sp.csr_matrix(irr)
    bb = -irr.dot(gen_ref.xstar(3000, 4))
    op = SparseOperator(irr)
    assert op.summary[0] >= 1
    a, c = _pair(op, bb, ss.BoxProjOp(3000, -0.5 * np.ones(3000), 0.5 * np.ones(3000)), 1e-8, 5000)
    _same(a, c)
    A, b, _ = gen_ref.contact_problem(2000, 3000, 5)
    D = SparseOperator(A.D)
    op = ComposedOperator(D, DiagonalOperator(A.minv), D.T, shift=A.eps)
    for cap in (5000, 40, 7):
        a, c = _pair(op, b, ss.LowerBoundProjOp(2000, np.zeros(2000)), 1e-8, cap)
        _same(a, c)


def test_pipeline_request_falls_back_where_it_does_not_apply():
    """Sphere sets, the fallback variant and the batch schedulers take the host-driven loop."""
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd.batch import solve_batch_fused
    n = 1024
    op, b, _ = problems.random_problem(n, 7)
    a, c = _pair(op, b, ss.SphereProjOp(n, 3.0), 1e-8, 5000)
    _same(a, c)
    with contextlib.redirect_stdout(io.StringIO()):
        f1 = solvers.CCQPSolverBBPGDf(1e-8, 5000).solve(op, b, convex_proj_op=ss.BoxProjOp(n, -0.5, 0.5))
        f2 = solvers.CCQPSolverBBPGDf(1e-8, 5000, pipeline=True).solve(op, b, convex_proj_op=ss.BoxProjOp(n, -0.5, 0.5))
    _same(f1, f2)
    probs = [(op, b, None, ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n)))] * 3
    got = solve_batch_fused(lambda: solvers.CCQPSolverBBPGD(1e-8, 5000, pipeline=True), probs, lanes=3, groups=1)
    with contextlib.redirect_stdout(io.StringIO()):
        ref = solvers.CCQPSolverBBPGD(1e-8, 5000).solve(*probs[0][:2], convex_proj_op=probs[0][3])
    for g in got:
        _same(ref, g)


@pytest.mark.parametrize("case_id", range(40))
def test_pipelined_bbpgd_randomised(case_id):
    """Random size, set (array / scalar / default bounds), tolerance, MV cap and starting point."""
    from ccqppy_amd import problems, solution_spaces as ss
    rng = np.random.default_rng(700 + case_id)
    n = int(rng.integers(2, 6000))
    op, b, _ = problems.random_problem(n, int(rng.integers(1, 10 ** 6)))
    kind = ["Box", "Lower", "Upper", "Identity"][rng.integers(4)]
    mode = rng.integers(3)
    lo = -rng.uniform(0.1, 1.0, size=n) if mode == 0 else (-float(rng.uniform(0.1, 1.0)) if mode == 1 else None)
    hi = rng.uniform(0.1, 1.0, size=n) if mode == 0 else (float(rng.uniform(0.1, 1.0)) if mode == 1 else None)
    P = {"Box": ss.BoxProjOp(n, *[v for v in (lo, hi) if v is not None]) if mode != 2 else ss.BoxProjOp(n),
         "Lower": ss.LowerBoundProjOp(n, lo) if lo is not None else ss.LowerBoundProjOp(n),
         "Upper": ss.UpperBoundProjOp(n, hi) if hi is not None else ss.UpperBoundProjOp(n),
         "Identity": ss.IdentityProjOp(n)}[kind]
    tol = float(rng.choice([1e-5, 1e-8, 1e-11]))
    cap = int(rng.choice([2, 3, 5, 9, 17, 5000]))
    x0 = None if rng.random() < 0.5 else rng.uniform(-0.5, 0.5, size=n)
    a, c = _pair(op, b, P, tol, cap, x0=x0)
    _same(a, c)


def test_back_to_back_pipelined_solves_never_see_the_previous_solves_mirror():
    """ccqp_pipe_poll right after ccqp_pipe_reset used to be able to read the "converged" flag the PREVIOUS solve had
    left in the pinned mirror (the resetting kernel had not run yet): thousands of short solves in a row, every one
    must report its own iteration count."""
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    n = 128
    ops = [problems.random_problem(n, 500 + p)[:2] for p in range(40)]
    P = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    want = []
    with contextlib.redirect_stdout(io.StringIO()):
        for op, b in ops:
            want.append(solvers.CCQPSolverBBPGD(1e-8, 5000).solve(op, b, convex_proj_op=P)
                        .solution_num_matrix_vector_multiplications)
        for rep in range(60):
            for (op, b), mv in zip(ops, want):
                r = solvers.CCQPSolverBBPGD(1e-8, 5000, pipeline=True).solve(op, b, convex_proj_op=P)
                assert r.solution_num_matrix_vector_multiplications == mv and r.solution_converged
