"""GPU: batch-sharded mode (BASELINE config 5: many independent box-QPs, MPRGP, no collective).
The interleaving scheduler must not change any result: every problem's x, residual and MV counts
are bitwise those of a plain sequential solve(), for several lane counts; and the MPRGP solves agree
with the CPU oracle on the per-problem MV counts."""
import contextlib
import io

import numpy as np
import pytest

from oracle import cpu_ref, gen_ref

pytestmark = pytest.mark.gpu


def _problems(n, count):
    from ccqppy_amd import problems, solution_spaces as ss
    out = []
    for p in range(count):
        op, b, _ = problems.random_problem(n, 1234 + p)
        out.append((op, b, None, ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))))
    return out


@pytest.mark.parametrize("sname,args", [("MPRGP", (1e-6, 300)), ("BBPGD", (1e-8, 5000)), ("APGD", (1e-8, 5000)),
                                        ("BBPGDf", (1e-8, 5000)), ("PGD", (1e-8, 5000, 0.3))])
def test_batch_equals_sequential(sname, args):
    from ccqppy_amd import solvers
    from ccqppy_amd.batch import solve_batch
    cls = getattr(solvers, "CCQPSolver" + sname)
    probs = _problems(1024, 7)
    seq = []
    for A, b, x0, P in probs:
        with contextlib.redirect_stdout(io.StringIO()):
            seq.append(cls(*args).solve(A, b, x0, P))
    for lanes in (1, 3, 8):
        got = solve_batch(lambda: cls(*args), probs, lanes=lanes)
        assert len(got) == len(seq)
        for g, s in zip(got, seq):
            assert g.solution_num_matrix_vector_multiplications == s.solution_num_matrix_vector_multiplications
            assert g.executed_matrix_vector_multiplications == s.executed_matrix_vector_multiplications
            assert g.solution_converged == s.solution_converged
            assert float(g.solution_residual) == float(s.solution_residual)
            a = g.solution.cpu().numpy()
            bb = s.solution.cpu().numpy()
            assert np.array_equal(a.view(np.uint64), bb.view(np.uint64))


def test_batch_mprgp_matches_oracle_counts():
    from ccqppy_amd import solvers
    from ccqppy_amd.batch import solve_batch
    n, count = 2048, 5
    probs = _problems(n, count)
    got = solve_batch(lambda: solvers.CCQPSolverMPRGP(1e-6, 300), probs, lanes=4)
    for p, g in enumerate(got):
        A = gen_ref.spd_rows(n, 1234 + p)
        b = probs[p][1].cpu().numpy()
        o = cpu_ref.solve_mprgp(A, b, 1e-6, 300, P=cpu_ref.Box(n, -0.5 * np.ones(n), 0.5 * np.ones(n)))
        assert g.solution_num_matrix_vector_multiplications == o.solution_num_matrix_vector_multiplications
        assert g.executed_matrix_vector_multiplications == o.executed_mv
        assert np.linalg.norm(g.solution.cpu().numpy() - o.solution) <= 1e-10 * np.linalg.norm(o.solution)


def test_batch_partition_driver_single_rank():
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd.batch import solve_batch_sharded
    n = 512

    def make(p):
        op, b, _ = problems.random_problem(n, 100 + p)
        return op, b, None, ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    for fused in (True, False):
        res = solve_batch_sharded(lambda: solvers.CCQPSolverMPRGP(1e-6, 300), make, 6, lanes=3, fused=fused)
        assert sorted(res) == list(range(6))
        assert all(r.solution_converged for r in res.values())


@pytest.mark.parametrize("kind", ["Box", "Lower", "Upper", "Identity"])
def test_mprgp_fused_kernels_equal_literal_sequence(kind):
    """ccqp_mprgp_* / ccqp_free_gradient / ccqp_cg_direction are pure launch-count optimisations:
    solves with and without them must agree bit for bit (x, residual, MV counts)."""
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    for n in (1000, 4096, 20000):          # 20000 > 8192: exercises the two-stage reductions
        op, b, _ = problems.random_problem(n, 77)
        lo, hi = -0.3 * np.ones(n), 0.4 * np.ones(n)
        P = {"Box": ss.BoxProjOp(n, lo, hi), "Lower": ss.LowerBoundProjOp(n, lo),
             "Upper": ss.UpperBoundProjOp(n, hi), "Identity": ss.IdentityProjOp(n)}[kind]
        out = []
        for fused in (True, False):
            solvers.CCQPSolverMPRGP.FUSED = fused
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    out.append(solvers.CCQPSolverMPRGP(1e-7, 400).solve(op, b, convex_proj_op=P))
            finally:
                solvers.CCQPSolverMPRGP.FUSED = True
        a, c = out
        assert a.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
        assert a.executed_matrix_vector_multiplications == c.executed_matrix_vector_multiplications
        assert float(a.solution_residual) == float(c.solution_residual)
        assert np.array_equal(a.solution.cpu().numpy().view(np.uint64), c.solution.cpu().numpy().view(np.uint64))


@pytest.mark.parametrize("sname,args", [("MPRGP", (1e-6, 300)), ("BBPGD", (1e-8, 5000)), ("APGD", (1e-8, 5000)),
                                        ("APGDAntiRelaxation", (1e-8, 5000)), ("BBPGDf", (1e-8, 5000)),
                                        ("PGD", (1e-8, 5000, 0.3))])
def test_fused_batch_equals_sequential(sname, args):
    """solve_batch_fused merges the same operation of many solves into one batched launch
    (ccqp_batch_run_f64); results must be bitwise those of plain sequential solves."""
    from ccqppy_amd import solvers
    from ccqppy_amd.batch import solve_batch_fused
    cls = getattr(solvers, "CCQPSolver" + sname)
    probs = _problems(1024, 9) + _problems(3000, 3)          # two sizes in one batch (nblk 1 and 2)
    seq = []
    for A, b, x0, P in probs:
        with contextlib.redirect_stdout(io.StringIO()):
            seq.append(cls(*args).solve(A, b, x0, P))
    for lanes, groups in ((1, 1), (5, 2), (12, 2), (12, 3)):
        stats = {}
        got = solve_batch_fused(lambda: cls(*args), probs, lanes=lanes, groups=groups, stats=stats)
        assert stats["ops"] >= stats["launches"] > 0
        for g, s in zip(got, seq):
            assert g.solution_num_matrix_vector_multiplications == s.solution_num_matrix_vector_multiplications
            assert g.executed_matrix_vector_multiplications == s.executed_matrix_vector_multiplications
            assert g.solution_converged == s.solution_converged
            assert float(g.solution_residual) == float(s.solution_residual)
            assert np.array_equal(g.solution.cpu().numpy().view(np.uint64), s.solution.cpu().numpy().view(np.uint64))


def test_fused_batch_with_non_elementwise_sets_falls_back_in_order():
    """Sphere / Disjoint projections have no batched form: their kernels are launched immediately after
    the lane's recorded operations have been flushed, so program order (and the result) is unchanged."""
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd.batch import solve_batch_fused
    n = 900
    probs = []
    for p in range(5):
        op, b, _ = problems.random_problem(n, 50 + p)
        P = ss.SphereProjOp(n, 4.0) if p % 2 else ss.DisjointProjOp(
            ss.BoxProjOp(n // 2, -0.5 * np.ones(n // 2), 0.5 * np.ones(n // 2)), ss.SphereProjOp(n - n // 2, 2.0))
        probs.append((op, b, None, P))
    for cls, args in ((solvers.CCQPSolverBBPGD, (1e-8, 500)), (solvers.CCQPSolverMPRGP, (1e-6, 200))):
        seq = []
        for A, b, x0, P in probs:
            with contextlib.redirect_stdout(io.StringIO()):
                seq.append(cls(*args).solve(A, b, x0, P))
        got = solve_batch_fused(lambda: cls(*args), probs, lanes=4, groups=2)
        for g, s in zip(got, seq):
            assert g.solution_num_matrix_vector_multiplications == s.solution_num_matrix_vector_multiplications
            assert np.array_equal(g.solution.cpu().numpy().view(np.uint64), s.solution.cpu().numpy().view(np.uint64))


def test_spg_with_private_rng_is_reproducible_under_interleaving():
    """SPG draws one uniform per iteration from numpy's GLOBAL stream (solvers.py:959), so interleaved
    solves would consume it in scheduling order.  With the `rng` extension every solve owns its stream
    and the batch result equals the sequential one bit for bit."""
    from ccqppy_amd import solvers
    from ccqppy_amd.batch import solve_batch, solve_batch_fused
    probs = _problems(1024, 6)

    def make(i=[0]):
        s = solvers.CCQPSolverSPG(1e-8, 5000)
        s.rng = np.random.RandomState(100 + i[0] % 6)
        i[0] += 1
        return s
    seq = []
    for A, b, x0, P in probs:
        with contextlib.redirect_stdout(io.StringIO()):
            seq.append(make().solve(A, b, x0, P))
    for runner in (lambda: solve_batch(make, probs, lanes=3), lambda: solve_batch_fused(make, probs, lanes=6, groups=2)):
        got = runner()
        for g, s in zip(got, seq):
            assert g.solution_num_matrix_vector_multiplications == s.solution_num_matrix_vector_multiplications
            assert np.array_equal(g.solution.cpu().numpy().view(np.uint64), s.solution.cpu().numpy().view(np.uint64))


def test_gemv_bandwidth_floor():
    """Regression guard for the dominant kernel: n = 32768 (8 GiB) must stream at >= 5.0 TB/s
    (measured 6.7-7.0 TB/s on MI355X; 8 TB/s peak)."""
    import torch
    from ccqppy_amd import problems
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    n = 32768
    A = problems.spd_rows(n, 1)
    x = problems.xstar(n, 1)
    y = eng.empty(n)
    for _ in range(3):
        eng.gemv(A, x, y)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        eng.gemv(A, x, y)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 20
    tbs = 8.0 * (n * n + 2 * n) / ms / 1e9
    assert tbs >= 5.0, "GEMV at %.2f TB/s" % tbs
