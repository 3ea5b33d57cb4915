"""GPU: the "next" rows of SURVEY.md 8(f) -- the ensemble harness (counterpart of
benchmarks/benchmark_random_ccqp.py) and the analytic problem suite module -- to the same parity
bar as the hot path: per-cell MV counts equal to the CPU oracle run on the same problems."""
import contextlib
import io

import numpy as np
import pytest

from oracle import cpu_ref, gen_ref

pytestmark = pytest.mark.gpu


def test_problem_suite_module_matches_reference_data():
    from ccqppy_amd import problem_suite, solvers
    from oracle import cases as C
    for name, pd in C.SUITE.items():
        p = getattr(problem_suite, name)()
        A, b = C.suite_problem(name)
        assert p.number_of_unknowns == 3
        assert np.array_equal(p.A, A) and np.array_equal(p.b, b) and p.b.dtype == b.dtype
        assert np.array_equal(p.exact_solution, np.array(pd["xexact"]))
        with contextlib.redirect_stdout(io.StringIO()):
            r = solvers.CCQPSolverBBPGD(1e-8, 10000).solve(p.A, p.b, convex_proj_op=p.convex_proj_op)
        assert r.solution_converged and np.linalg.norm(r.solution - p.exact_solution) < 1e-5


@pytest.mark.parametrize("lanes", [1, 3])
def test_ensemble_grid_matches_oracle(lanes):
    from ccqppy_amd import solvers, solution_spaces as ss
    from ccqppy_amd.ensemble import BenchmarkRandomCCQP
    sizes = [6, 9, 12, 48]
    tol, max_mv, trials = 1e-5, 5000, 3
    slv = [solvers.CCQPSolverPGD(tol, max_mv, 0.2), solvers.CCQPSolverAPGD(tol, max_mv),
           solvers.CCQPSolverBBPGD(tol, max_mv), solvers.CCQPSolverBBPGDf(tol, max_mv),
           solvers.CCQPSolverMPRGP(tol, max_mv)]
    fams = [[ss.IdentityProjOp(d) for d in sizes], [ss.BoxProjOp(d) for d in sizes],
            [ss.SphereProjOp(d) for d in sizes],
            [ss.DisjointProjOp(*[ss.SphereProjOp(3)] * (d // 3)) for d in sizes]]
    bench = BenchmarkRandomCCQP(trials, slv, fams, lanes=lanes).run()
    if lanes > 1:
        # the lock-step scheduler (taken where it applies) and the stream-lane scheduler agree bit for bit
        other = BenchmarkRandomCCQP(trials, slv, fams, lanes=lanes, lockstep=False).run()
        assert np.array_equal(bench._problem_num_matrix_vector_mults, other._problem_num_matrix_vector_mults)
        assert np.array_equal(bench._problem_residual.view(np.uint64), other._problem_residual.view(np.uint64))
        assert np.array_equal(bench._problem_converged, other._problem_converged)
    shape = (len(slv), len(fams), len(sizes), trials)
    assert bench._problem_num_matrix_vector_mults.shape == shape
    assert bench._problem_time.dtype == np.float32 and bench._problem_converged.dtype == int
    ofam = [[cpu_ref.Identity(d) for d in sizes], [cpu_ref.Box(d) for d in sizes],
            [cpu_ref.Sphere(d) for d in sizes],
            [cpu_ref.Disjoint(*[cpu_ref.Sphere(3)] * (d // 3)) for d in sizes]]
    osolve = [lambda A, b, P: cpu_ref.solve_pgd(A, b, tol, max_mv, 0.2, P=P),
              lambda A, b, P: cpu_ref.solve_apgd(A, b, tol, max_mv, P=P),
              lambda A, b, P: cpu_ref.solve_bbpgd(A, b, tol, max_mv, P=P),
              lambda A, b, P: cpu_ref.solve_bbpgdf(A, b, tol, max_mv, P=P),
              lambda A, b, P: cpu_ref.solve_mprgp(A, b, tol, max_mv, P=P)]
    mism = 0
    for pi, n in enumerate(sizes):
        for t in range(trials):
            A, b, _ = gen_ref.problem(n, t)
            for f in range(len(fams)):
                for s in range(len(slv)):
                    with np.errstate(all="ignore"):
                        o = osolve[s](A, b, ofam[f][pi])
                    got = bench._problem_num_matrix_vector_mults[s, f, pi, t]
                    mism += int(got != o.solution_num_matrix_vector_multiplications)
                    assert bool(bench._problem_converged[s, f, pi, t]) == bool(o.solution_converged)
    # tol 1e-5 on well-conditioned problems: counts agree cell by cell (b differs in the last bit
    # between the device and host matvec, so allow a handful of off-by-one-iteration cells)
    assert mism <= 0.02 * np.prod(shape)
    out = bench.process_results(stream=io.StringIO())
    assert set(out) == {"wall-clock time [s]", "number of matrix-vector multiplications", "solution residual"}
    m = out["number of matrix-vector multiplications"]
    assert np.all(m["lower_95"] <= m["mean"]) and np.all(m["mean"] <= m["upper_95"])


def test_ensemble_spg_cells_with_private_generators():
    """spg_seed: trial t of every SPG cell draws from RandomState(spg_seed + t) -- the same results under the
    lock-step scheduler (ccqp_bspg.hip), the stream lanes and the sequential loop, and the MV counts of the oracle run
    with np.random.seed(spg_seed + t) before each solve."""
    from ccqppy_amd import solvers, solution_spaces as ss
    from ccqppy_amd.ensemble import BenchmarkRandomCCQP
    sizes = [7, 49, 130]
    tol, max_mv, trials, seed = 1e-5, 5000, 3, 7
    slv = [solvers.CCQPSolverSPG(tol, max_mv), solvers.CCQPSolverSPG(tol, max_mv, m=2, tau=0.4)]
    fams = [[ss.IdentityProjOp(d) for d in sizes], [ss.BoxProjOp(d) for d in sizes],
            [ss.SphereProjOp(d) for d in sizes],
            [ss.DisjointProjOp(*[ss.SphereProjOp(3)] * (d // 3), ss.IdentityProjOp(d % 3)) for d in sizes]]
    runs = [BenchmarkRandomCCQP(trials, slv, fams, lanes=3, spg_seed=seed).run(),
            BenchmarkRandomCCQP(trials, slv, fams, lanes=3, lockstep=False, spg_seed=seed).run(),
            BenchmarkRandomCCQP(trials, slv, fams, lanes=1, spg_seed=seed).run()]
    assert set(runs[0].cell_scheduler.values()) == {"lock-step"}
    assert set(runs[1].cell_scheduler.values()) == {"stream lanes"}
    assert set(runs[2].cell_scheduler.values()) == {"sequential"}
    for other in runs[1:]:
        assert np.array_equal(runs[0]._problem_num_matrix_vector_mults, other._problem_num_matrix_vector_mults)
        assert np.array_equal(runs[0]._problem_residual.view(np.uint64), other._problem_residual.view(np.uint64))
        assert np.array_equal(runs[0]._problem_converged, other._problem_converged)
    assert runs[0]._problem_converged.all()
    # without spg_seed the SPG cells stay on the host-driven schedulers (the global stream cannot be split)
    plain = BenchmarkRandomCCQP(1, slv[:1], [fams[1][1:2]], lanes=3).run()
    assert set(plain.cell_scheduler.values()) == {"stream lanes"}
    ofam = [[cpu_ref.Identity(d) for d in sizes], [cpu_ref.Box(d) for d in sizes], [cpu_ref.Sphere(d) for d in sizes],
            [cpu_ref.Disjoint(*[cpu_ref.Sphere(3)] * (d // 3), cpu_ref.Identity(d % 3)) for d in sizes]]
    kws = [{}, {"m": 2, "tau": 0.4}]
    mism = 0
    for pi, n in enumerate(sizes):
        for t in range(trials):
            A, b, _ = gen_ref.problem(n, t)
            for f in range(len(fams)):
                for s in range(len(slv)):
                    np.random.seed(seed + t)
                    with np.errstate(all="ignore"):
                        o = cpu_ref.solve_spg(A, b, tol, max_mv, P=ofam[f][pi], **kws[s])
                    mism += int(runs[0]._problem_num_matrix_vector_mults[s, f, pi, t] !=
                                o.solution_num_matrix_vector_multiplications)
    assert mism <= 0.05 * runs[0]._problem_num_matrix_vector_mults.size
