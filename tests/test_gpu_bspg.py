"""GPU: lock-step SPG-QP (ccqp_bspg_run_f64 in ccqp_bspg.hip; batch.solve_batch_pipelined with CCQPSolverSPG solvers
that carry their own RandomState).  An execution strategy only: per problem x, the residual, the converged flag, the
reported MV count AND the state its generator is left in must be those of that problem's own solve() with the same
generator, bit for bit -- elementwise and segmented sets, several reduction blocks, warm starts, mixed tolerances and
queue lengths, every MV cap (including the ones at which the reference dies), solves longer than one driver call.
Reference loop: solvers.py:878-975 as run by benchmarks/benchmark_random_ccqp.py:67-102."""
import contextlib
import io

import numpy as np
import pytest

from tests.test_gpu_bsegs import _problems, _set  # noqa: F401  (the same problem families)

pytestmark = pytest.mark.gpu


def _same(a, c):
    assert a.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
    assert bool(a.solution_converged) == bool(c.solution_converged)
    assert float(a.solution_residual) == float(c.solution_residual)
    xa, xc = a.solution, c.solution
    xa = xa.cpu().numpy() if hasattr(xa, "cpu") else xa
    xc = xc.cpu().numpy() if hasattr(xc, "cpu") else xc
    assert np.array_equal(xa.view(np.uint64), xc.view(np.uint64))
    sa, sc = a.rng.get_state(), c.rng.get_state()
    assert sa[2] == sc[2] and np.array_equal(sa[1], sc[1]), "generator left at another position"


def _elementwise(n, p, family):
    from ccqppy_amd import solution_spaces as ss
    if family == "box":
        return ss.BoxProjOp(n, -0.5 * np.ones(n) - 0.01 * p, 0.5 * np.ones(n))
    if family == "lower":
        return ss.LowerBoundProjOp(n, -0.2 * np.ones(n))
    if family == "identity":
        return None
    raise KeyError(family)


def _make_problems(n, count, seed, family):
    from ccqppy_amd import problems
    if family in ("box", "lower", "identity"):
        out = []
        for p in range(count):
            op, b, _ = problems.random_problem(n, seed + p)
            out.append((op, b, None, _elementwise(n, p, family)))
        return out
    return _problems(n, count, seed, family)


def _solver(tol, max_mv, seed, **kw):
    from ccqppy_amd import solvers
    s = solvers.CCQPSolverSPG(tol, max_mv, **kw)
    s.rng = np.random.RandomState(seed)
    return s


def _reference(probs, tols, max_mv, seeds, kws):
    ref = []
    for (A, b, x0, P), tol, seed, kw in zip(probs, tols, seeds, kws):
        with contextlib.redirect_stdout(io.StringIO()):
            ref.append(_solver(tol, max_mv, seed, **kw).solve(A, b, x0=x0, convex_proj_op=P))
    return ref


def _batch(probs, tols, max_mv, seeds, kws, expect_fallbacks=0):
    from ccqppy_amd import batch, lockstep
    it = iter(zip(tols, seeds, kws))
    before = dict(lockstep.stats)

    def make():
        tol, seed, kw = next(it)
        return _solver(tol, max_mv, seed, **kw)
    out = batch.solve_batch_pipelined(make, probs)
    assert lockstep.stats["fallback_solves"] == before["fallback_solves"] + expect_fallbacks
    return out


@pytest.mark.parametrize("family", ["box", "lower", "identity", "sphere", "cone", "spheres3", "mixed"])
@pytest.mark.parametrize("n", [96, 257, 1024, 2500])
def test_lockstep_spg_is_the_problems_own_solve(family, n):
    count = 7
    probs = _make_problems(n, count, 40 + n, family)
    tols = [1e-6, 1e-8, 1e-4, 1e-7, 1e-8, 1e-5, 1e-8]
    seeds = [11, 12, 13, 14, 15, 16, 17]
    kws = [{}, {"m": 1}, {"m": 3, "tau": 0.3}, {"sigma1": 0.05, "sigma2": 0.8}, {}, {"m": 8}, {"tau": 0.7}]
    max_mv = 400 if family in ("cone", "mixed") else 20000       # (the bugged Cone need not converge)
    ref = _reference(probs, tols, max_mv, seeds, kws)
    got = _batch(probs, tols, max_mv, seeds, kws)
    for a, c in zip(got, ref):
        _same(a, c)
    if max_mv == 20000:
        assert all(s.solution_converged for s in got)


@pytest.mark.parametrize("max_mv", [4, 5, 6, 11, 12, 37])
@pytest.mark.parametrize("family", ["box", "mixed"])
def test_every_budget(family, max_mv):
    n, count = 300, 5
    probs = _make_problems(n, count, 7, family)
    tols = [1e-12, 1e-3, 1e-12, 1e-2, 1e-12]
    seeds = list(range(5))
    kws = [{}] * count
    ref = _reference(probs, tols, max_mv, seeds, kws)
    got = _batch(probs, tols, max_mv, seeds, kws)
    for a, c in zip(got, ref):
        _same(a, c)
    assert any(not s.solution_converged for s in got)


@pytest.mark.parametrize("max_mv", [1, 2, 3])
def test_budgets_at_which_the_reference_dies(max_mv):
    """max_mv <= 3: the loop breaks before the first d.d and `residual = sqrt(dkdotdk)` raises UnboundLocalError
    (solvers.py:938-941,970) -- the lock-step run hands such problems to the host loop, which raises."""
    probs = _make_problems(128, 3, 3, "box")
    with pytest.raises(UnboundLocalError):
        _batch(probs, [1e-8] * 3, max_mv, [1, 2, 3], [{}] * 3, expect_fallbacks=1)


@pytest.mark.parametrize("rounds", [9, 1])
def test_solves_that_span_several_driver_calls(monkeypatch, rounds):
    """Some problems need further calls of the C driver while others have long finished: the rounds per call shrunk
    so that ordinary solves exercise the resume path."""
    from ccqppy_amd import lockstep
    monkeypatch.setattr(lockstep, "_SPG_ROUNDS", rounds)
    n, count = 200, 6
    probs = _make_problems(n, count, 90, "box")
    tols = [1e-13, 1e-3, 1e-13, 1e-6, 1e-13, 1e-2]
    seeds = [5, 6, 7, 8, 9, 10]
    kws = [{}] * count
    max_mv = 700
    ref = _reference(probs, tols, max_mv, seeds, kws)
    mvs = [r.solution_num_matrix_vector_multiplications for r in ref]
    assert max(mvs) > 2 + 3 * rounds and min(mvs) < max(mvs) // 2
    got = _batch(probs, tols, max_mv, seeds, kws)
    for a, c in zip(got, ref):
        _same(a, c)


def test_warm_starts_numpy_inputs_and_mixed_sets():
    from ccqppy_amd import problems
    n, count = 513, 6
    fams = ["sphere", "mixed", "spheres3"]
    probs = []
    rs = np.random.default_rng(3)
    for p in range(count):
        op, b, _ = problems.random_problem(n, 500 + p)
        x0 = rs.standard_normal(n) * 0.1 if p % 2 else None
        probs.append((op, b.cpu().numpy() if p % 3 == 0 else b, x0, _set(n, p, fams[p % 3])))
    tols = [1e-6] * count
    seeds = list(range(20, 20 + count))
    kws = [{}] * count
    ref = _reference(probs, tols, 300, seeds, kws)
    got = _batch(probs, tols, 300, seeds, kws)
    for a, c in zip(got, ref):
        _same(a, c)
        assert type(a.solution) is type(c.solution)


def test_large_batch_many_reduction_blocks():
    """n = 5000: three stage-1 blocks per dot product; 40 problems."""
    n, count = 5000, 40
    probs = _make_problems(n, 4, 77, "box") * 10
    tols = [1e-7] * count
    seeds = list(range(100, 100 + count))
    kws = [{}] * count
    ref = _reference(probs[:8], tols[:8], 20000, seeds[:8], kws[:8])
    got = _batch(probs, tols, 20000, seeds, kws)
    for a, c in zip(got[:8], ref):
        _same(a, c)
    # the same problem with another generator: another path, the same solution to the tolerance
    x0 = got[0].solution.cpu().numpy()
    x4 = got[4].solution.cpu().numpy()
    assert not np.array_equal(x0, x4) and np.linalg.norm(x0 - x4) <= 1e-5 * np.linalg.norm(x0)


def test_rejects_what_it_cannot_run():
    from ccqppy_amd import batch, solvers
    probs = _make_problems(128, 2, 1, "box")
    with pytest.raises(TypeError, match="RandomState"):        # the global stream cannot be split
        batch.solve_batch_pipelined(lambda: solvers.CCQPSolverSPG(1e-8, 100), probs)
    shared = np.random.RandomState(1)

    def make_shared():
        s = solvers.CCQPSolverSPG(1e-8, 100)
        s.rng = shared
        return s
    with pytest.raises(TypeError, match="share"):
        batch.solve_batch_pipelined(make_shared, probs)

    def make_m0():
        s = solvers.CCQPSolverSPG(1e-8, 100, m=0)
        s.rng = np.random.RandomState(1)
        return s
    with pytest.raises(ValueError, match="queue length"):
        batch.solve_batch_pipelined(make_m0, probs)


def test_a_draw_that_raises_in_the_reference_raises_here():
    """An indefinite Hessian drives tau^2 beta^2 + 2 xi below zero: betahat is NaN and numpy's uniform raises
    OverflowError('Range exceeds valid bounds') (solvers.py:959) -- after some successful draws, or at the first one.
    The lock-step run raises the same exception and leaves the generator where the reference's loop leaves it."""
    import torch
    from ccqppy_amd import batch
    rs = np.random.default_rng(11)
    seen_late = seen_first = 0
    for trial in range(12):
        n = 96
        Q = rs.standard_normal((n, n))
        A = torch.from_numpy((Q + Q.T) * 0.5 - (0.0 if trial % 2 else 3.0) * np.eye(n)).cuda()
        b = torch.from_numpy(rs.standard_normal(n)).cuda()
        ref = _solver(1e-9, 400, 50 + trial)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                ref.solve(A, b)
            continue                                     # (this one happened to stop some other way)
        except OverflowError as e:
            msg = str(e)
        pos = ref.rng.get_state()
        fresh = np.random.RandomState(50 + trial).get_state()
        first = pos[2] == fresh[2] and np.array_equal(pos[1], fresh[1])
        seen_first += first
        seen_late += not first
        made = []

        def make():
            made.append(_solver(1e-9, 400, 50 + trial))
            return made[-1]
        with pytest.raises(OverflowError) as info:
            batch.solve_batch_pipelined(make, [(A, b, None, None)])
        assert str(info.value) == msg
        got = made[0].rng.get_state()
        assert got[2] == pos[2] and np.array_equal(got[1], pos[1])
    assert seen_late >= 2


@pytest.mark.parametrize("family", ["box", "identity", "sphere", "mixed"])
@pytest.mark.parametrize("n", [130, 1500])
def test_single_solves_with_pipeline_true(family, n):
    """CCQPSolverSPG(pipeline=True): a batch of one -- with a private generator and with numpy's global one."""
    from ccqppy_amd import lockstep, solvers
    probs = _make_problems(n, 3, 300 + n, family)
    max_mv = 150 if family == "mixed" else 5000
    for k, (A, b, x0, P) in enumerate(probs):
        out = []
        for pipeline in (False, True):
            np.random.seed(77 + k)
            s = solvers.CCQPSolverSPG(1e-7, max_mv, pipeline=pipeline)
            if k == 1:
                s.rng = np.random.RandomState(5)
            before = dict(lockstep.stats)
            with contextlib.redirect_stdout(io.StringIO()):
                s.solve(A, b, x0=x0, convex_proj_op=P)
            assert lockstep.stats == before
            out.append((s, np.random.get_state(), s.rng.get_state() if s.rng is not None else None))
        (a, ga, pa), (c, gc, pc) = out
        assert a.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
        assert torch_equal_bits(a.solution, c.solution) and float(a.solution_residual) == float(c.solution_residual)
        assert bool(a.solution_converged) == bool(c.solution_converged)
        assert ga[2] == gc[2] and np.array_equal(ga[1], gc[1])           # the global stream: same position
        if pa is not None:
            assert pa[2] == pc[2] and np.array_equal(pa[1], pc[1])
        # the pipelined solve really took the lock-step driver: no per-iteration dot-product reads were issued
        assert c.executed_matrix_vector_multiplications <= a.executed_matrix_vector_multiplications


def torch_equal_bits(x, y):
    x = x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)
    y = y.cpu().numpy() if hasattr(y, "cpu") else np.asarray(y)
    return np.array_equal(x.view(np.uint64), y.view(np.uint64))


@pytest.mark.parametrize("count", [31, 32, 33, 70])
def test_two_alternating_halves_give_the_same_bits_as_one_group(monkeypatch, count):
    """From 32 problems on the C driver advances the batch as two halves that take turns on the device; a batch with
    very different solve lengths in the two halves (one half finishes long before the other), segmented and
    elementwise, against the one-group form and a sample of per-problem solves."""
    from ccqppy_amd import lockstep
    n = 150
    for family in ("box", "spheres3"):
        probs = _make_problems(n, count, 1000 + count, family)
        tols = [1e-3 if p < count // 2 else 1e-9 for p in range(count)]
        seeds = list(range(count))
        kws = [{}] * count
        monkeypatch.setattr(lockstep, "_SPG_ONE_GROUP", False)
        two = _batch(probs, tols, 3000, seeds, kws)
        monkeypatch.setattr(lockstep, "_SPG_ONE_GROUP", True)
        one = _batch(probs, tols, 3000, seeds, kws)
        for a, c in zip(two, one):
            _same(a, c)
        idx = [0, count // 2 - 1, count // 2, count - 1]
        ref = _reference([probs[i] for i in idx], [tols[i] for i in idx], 3000, [seeds[i] for i in idx], [{}] * 4)
        for i, c in zip(idx, ref):
            _same(two[i], c)


def test_batch_sharded_driver_picks_the_lockstep_form_for_spg_with_private_generators():
    """solve_batch_sharded(fused="auto"): SPG solvers that carry their own RandomState go to the lock-step driver, SPG
    solvers on numpy's global stream stay on the host-driven scheduler (counted as declined, not as a fall-back)."""
    from ccqppy_amd import batch, lockstep, solvers
    n, count = 140, 9
    probs = _make_problems(n, count, 60, "box")
    seeds = iter(range(200, 200 + count))

    def make():
        return _solver(1e-7, 500, next(seeds))
    before = dict(lockstep.stats)
    got = batch.solve_batch_sharded(make, lambda p: probs[p], count, gather=True)
    assert lockstep.stats == before                      # taken by the lock-step scheduler, nothing handed back
    ref = _reference(probs, [1e-7] * count, 500, list(range(200, 200 + count)), [{}] * count)
    for p, c in enumerate(ref):
        assert got[p].solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
        assert float(got[p].solution_residual) == float(c.solution_residual)
        assert np.array_equal(np.asarray(got[p].solution).view(np.uint64), c.solution.cpu().numpy().view(np.uint64))
    np.random.seed(3)
    plain = batch.solve_batch_sharded(lambda: solvers.CCQPSolverSPG(1e-7, 500), lambda p: probs[p], count)
    assert lockstep.stats["auto_declined"] == before["auto_declined"] + 1
    assert lockstep.stats["fallback_solves"] == before["fallback_solves"]
    assert all(s.solution_converged for s in plain.values())
