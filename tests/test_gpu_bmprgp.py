"""GPU: lock-step MPRGP (ccqp_bmprgp_* in ccqp_bmprgp.hip; batch.solve_batch_pipelined with CCQPSolverMPRGP).  Every
branch decision of the reference's MPRGP-BB loop -- proportioning test, conjugate-gradient step, feasible-step
search, expansion half-step, proportioning step, BB step length, exit test -- is taken on the device; per problem
x, the residual, the converged flag, the reported and the executed MV counts must be those of that problem's own
solve(), bit for bit."""
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


def _sets(n, p, width=0.3):
    from ccqppy_amd import solution_spaces as ss
    lo, hi = -width * np.ones(n) - 0.01 * (p % 5), (width + 0.1) * np.ones(n) + 0.02 * (p % 3)
    return [ss.BoxProjOp(n, lo, hi), ss.LowerBoundProjOp(n, lo), ss.UpperBoundProjOp(n, hi), ss.IdentityProjOp(n),
            ss.BoxProjOp(n, 0.5 * lo, 0.25 * hi)][p % 5]


def _problems(n, count, seed, width=0.3):
    from ccqppy_amd import problems
    out = []
    for p in range(count):
        op, b, _ = problems.random_problem(n, seed + p)
        out.append((op, b, None, _sets(n, p, width)))
    return out


def _reference(probs, tols, max_mv):
    from ccqppy_amd import solvers
    ref = []
    for (A, b, x0, P), tol in zip(probs, tols):
        with contextlib.redirect_stdout(io.StringIO()):
            ref.append(solvers.CCQPSolverMPRGP(tol, max_mv).solve(A, b, x0=x0, convex_proj_op=P))
    return ref


def _batch(probs, tols, max_mv, **kw):
    from ccqppy_amd import batch, lockstep, solvers
    it = iter(tols)
    before = lockstep.stats["fallback_solves"]
    out = batch.solve_batch_pipelined(lambda: solvers.CCQPSolverMPRGP(next(it), max_mv), probs, **kw)
    # far from the matvec budget only a solve that does not converge at all may go back to the host loop
    assert max_mv < 2000 or lockstep.stats["fallback_solves"] - before <= sum(not c.solution_converged for c in out), \
        "a device-resident solve fell back to the host loop"
    return out


@pytest.mark.parametrize("n,count", [(100, 15), (257, 20), (1000, 10), (2049, 10), (4500, 5)])
def test_lockstep_mprgp_equals_individual_solves(n, count):
    probs = _problems(n, count, 2700 + n)
    tols = [1e-7] * count
    ref = _reference(probs, tols, 2000)
    got = _batch(probs, tols, 2000)
    for a, c in zip(ref, got):
        assert c.solution_converged
        _same(a, c)
    assert max(r.solution_num_matrix_vector_multiplications for r in ref) > 10


@pytest.mark.parametrize("n,count,segmented", [(300, 9, False), (1500, 5, False), (384, 7, True)])
def test_two_alternating_halves_change_no_bit(n, count, segmented, monkeypatch):
    """The A/B option of lockstep._bmprgp_run: the batch as two runs on two streams, the host alternating between them
    (a CPU test covers the polling logic; this one the streams, the staging on lane engines and the reassembly) -- with
    the halves' matvecs chained through the device's matvec stream (ccqp_batch_serialize_matvecs on the lanes: one after
    the other, each half's small launches under the other's matvec) and free-running.  Same bits as one run -- on
    elementwise sets and through the segmented form."""
    from ccqppy_amd import lockstep, solution_spaces as ss
    probs = _problems(n, count, 4100 + n)
    if segmented:
        probs = [(A, b, x0, ss.DisjointProjOp(ss.SphereProjOp(n // 2, 0.6 + 0.1 * p), ss.BoxProjOp(n - n // 2, -0.3 * np.ones(n - n // 2), 0.4 * np.ones(n - n // 2))))
                 for p, (A, b, x0, _) in enumerate(probs)]
    tols = [1e-6] * count
    one = _batch(probs, tols, 3000, single_launch=False)
    monkeypatch.setattr(lockstep, "_HALVES_DEFAULT", True)
    monkeypatch.setattr(lockstep, "_HALF_MIN_PROBLEMS", 2)
    monkeypatch.setattr(lockstep, "_HALF_MIN_BYTES", 0.0)
    monkeypatch.setattr(lockstep, "_HALF_MIN_N", 0)
    halved0 = lockstep.stats["halved_runs"]
    for chained, ahead in ((True, 1), (True, 4), (False, 4)):
        monkeypatch.setattr(lockstep, "_HALVES_CHAINED", chained)
        two = _batch(probs, tols, 3000, single_launch=False, ahead=ahead)
        for a, c in zip(one, two):
            _same(a, c)
    assert lockstep.stats["halved_runs"] == halved0 + 3
    ref = _reference(probs[:2], tols[:2], 3000)
    for a, c in zip(ref, one[:2]):
        _same(a, c)


@pytest.mark.parametrize("n,count", [(300, 70), (300, 3), (1024, 1), (2048, 5)])
def test_fused_stage_launch_of_one_chunk_problems_changes_no_bit(n, count):
    """Problems of ONE reduction chunk (n <= 2048), however many: the three decision stages and the update of a round are
    ONE launch with one workgroup per problem (bmprgp_tail_fused_kernel, bp_meet_seq) instead of four launches.  Same
    partial sums in the same order: the results of the four-launch form (context option MPRGP_FUSED_TAIL = 1) and of each
    problem's own solve()."""
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    probs = _problems(n, count, 5200 + n, width=0.25)
    tols = [1e-6] * count
    fused = _batch(probs, tols, 3000, single_launch=False)
    eng.set_option("MPRGP_FUSED_TAIL", 1)
    try:
        plain = _batch(probs, tols, 3000, single_launch=False)
    finally:
        eng.set_option("MPRGP_FUSED_TAIL", 0)
    for a, c in zip(plain, fused):
        assert c.solution_converged
        _same(a, c)
    ref = _reference(probs[::17], tols[::17], 3000)
    for a, c in zip(ref, fused[::17]):
        _same(a, c)


def test_fused_stage_launch_forced_on_longer_problems_in_a_child_process():
    """The fused launch walks up to 8 chunks per problem (bp_meet_seq); by default it is used for one-chunk problems only
    (it measured slower at config 5's shape).  The context option MPRGP_FUSED_TAIL = 2 forces it wherever it can run (a
    child process, so that no other test sees the option): 64 problems of n = 2049 (2 chunks) and n = 4500 (3 chunks)
    against their own solve()."""
    import os
    import subprocess
    import sys
    code = r"""
import contextlib, io, sys
import numpy as np, torch
sys.path.insert(0, %r)
from ccqppy_amd import batch, lockstep, problems, solvers, solution_spaces as ss
from ccqppy_amd._engine import get_engine, set_default_option
set_default_option("MPRGP_FUSED_TAIL", 2)
assert get_engine(0).options() == {"MPRGP_FUSED_TAIL": 2}
for n in (2049, 4500):
    probs = []
    for p in range(64):
        op, b, _ = problems.random_problem(n, 7700 + n + p)
        probs.append((op, b, None, ss.BoxProjOp(n, -0.25 * np.ones(n) - 0.01 * (p %% 5), 0.35 * np.ones(n))))
    before = lockstep.stats["fallback_solves"]
    got = batch.solve_batch_pipelined(lambda: solvers.CCQPSolverMPRGP(1e-6, 3000), probs, single_launch=False)
    assert lockstep.stats["fallback_solves"] == before
    for p in range(0, 64, 9):
        with contextlib.redirect_stdout(io.StringIO()):
            one = solvers.CCQPSolverMPRGP(1e-6, 3000).solve(*probs[p][:2], convex_proj_op=probs[p][3])
        c = got[p]
        assert c.solution_converged and torch.equal(one.solution.view(torch.int64), c.solution.view(torch.int64))
        assert one.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
        assert one.executed_matrix_vector_multiplications == c.executed_matrix_vector_multiplications
        assert float(one.solution_residual) == float(c.solution_residual)
print("FUSED-OK")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "FUSED-OK" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("n,count", [(256, 9), (1000, 7), (2050, 5), (130, 70)])
def test_lds_dma_forms_of_the_batched_matvec_change_no_bit(n, count):
    """Tuning word loads = 3: the batched matvec with its rows staged through per-wave LDS rings by LDS-DMA loads
    (gemv_lds_body / gemv_lds_multi_body) -- ring codes 0-3: workgroups per problem; 4-7: ONE persistent grid whose waves
    walk the live problems' matrices in order.  Full steps, a column tail (1000, 2050, 130), problems that converge at
    different rounds (dead problems drop out of the persistent form's list).  Same bits as the default form."""
    probs = _problems(n, count, 6100 + n, width=0.25)
    tols = [[1e-6, 1e-4, 1e30][p % 3] for p in range(count)]
    base = _batch(probs, tols, 3000, single_launch=False)
    for code in range(8):
        for per in ((0, 3) if code < 4 else (0,)):
            got = _batch(probs, tols, 3000, single_launch=False, tune=(3 << 1) | (code << 4) | (per << 8))
            for a, c in zip(base, got):
                _same(a, c)
    assert any(c.solution_num_matrix_vector_multiplications == 1 for c in base)


def test_matvec_timing_hook_of_the_c_abi():
    """ccqp_mv_timing_begin / _end: HIP event pairs around the batched matvec launches of a lock-step run, on their own
    stream -- what tools/config_profile_driver.py turns into config 5's live roofline object.  Results are untouched."""
    import ctypes
    from ccqppy_amd import _lib
    n, count = 700, 6
    probs = _problems(n, count, 31)
    tols = [1e-6] * count
    plain = _batch(probs, tols, 3000, single_launch=False)
    lib = _lib.load()
    _lib.check(lib.ccqp_mv_timing_begin(64))
    assert lib.ccqp_mv_timing_begin(64) != 0              # one session at a time
    timed = _batch(probs, tols, 3000, single_launch=False)
    buf, cnt = (ctypes.c_double * 64)(), ctypes.c_int()
    _lib.check(lib.ccqp_mv_timing_end(buf, 64, ctypes.byref(cnt)))
    assert lib.ccqp_mv_timing_end(buf, 64, ctypes.byref(cnt)) != 0     # closed
    rounds = max(c.executed_matrix_vector_multiplications for c in timed)
    # (one launch per executed matvec in the plain rounds; the fused rounds of this small batch multiply up to three
    # products per launch)
    assert 2 <= cnt.value <= 64 and cnt.value >= (min(rounds, 64) - 1) // 3
    ms = [buf[i] for i in range(cnt.value)]
    assert all(0.0 < t < 50.0 for t in ms)
    for a, c in zip(plain, timed):
        _same(a, c)


def test_tight_bounds_exercise_every_branch():
    """Narrow boxes: many active constraints, so proportioning steps, expansion half-steps and full CG steps all
    occur (the executed-minus-reported MV difference counts the uncounted A.dx products of each)."""
    n, count = 600, 16
    for width in (0.02, 0.1):
        probs = _problems(n, count, 41, width)
        tols = [1e-8] * count
        ref = _reference(probs, tols, 3000)
        for ahead in (1, 4):
            got = _batch(probs, tols, 3000, ahead=ahead)
            for a, c in zip(ref, got):
                _same(a, c)


def test_mixed_tolerances_warm_and_trivial_starts_numpy_and_oracle():
    n, count = 384, 15
    probs = _problems(n, count, 7)
    first = _reference(probs, [1e-9] * count, 3000)
    for p in range(0, count, 3):
        A, b, _, P = probs[p]
        probs[p] = (A, b, first[p].solution.clone(), P)
    tols = [[1e-10, 1e-6, 1e-3, 1e30][p % 4] for p in range(count)]
    ref = _reference(probs, tols, 3000)
    got = _batch(probs, tols, 3000)
    for a, c in zip(ref, got):
        _same(a, c)
    assert any(r.solution_num_matrix_vector_multiplications == 1 for r in ref)
    from ccqppy_amd import solution_spaces as ss
    A = gen_ref.spd_rows(192, 8)
    b = -A.dot(gen_ref.xstar(192, 8))
    lo, hi = -0.5 * np.ones(192), 0.5 * np.ones(192)
    got = _batch([(A, b, None, ss.BoxProjOp(192, lo, hi))], [1e-7], 3000)
    assert isinstance(got[0].solution, np.ndarray)
    o = cpu_ref.solve_mprgp(A, b, 1e-7, 3000, P=cpu_ref.Box(192, lo, hi))
    assert o.solution_num_matrix_vector_multiplications == got[0].solution_num_matrix_vector_multiplications
    assert o.executed_mv == got[0].executed_matrix_vector_multiplications
    assert np.linalg.norm(got[0].solution - o.solution) <= 1e-10 * np.linalg.norm(o.solution)


def test_under_a_matvec_budget():
    n, count = 300, 8
    probs = _problems(n, count, 11, 0.1)
    tols = [1e-7] * count
    full = _reference(probs, tols, 400)
    longest = max(r.solution_num_matrix_vector_multiplications for r in full)
    shortest = min(r.solution_num_matrix_vector_multiplications for r in full)
    assert longest < 400
    around = sorted(set(range(max(shortest - 3, 8), shortest + 3)) | set(range(longest - 3, longest + 3)) |
                    set(np.linspace(8, longest, 10).astype(int).tolist()))
    for cap in list(range(1, 8)) + around:
        ref = _reference(probs, tols, cap)
        got = _batch(probs, tols, cap)
        for a, c in zip(ref, got):
            _same(a, c)


@pytest.mark.parametrize("n,count", [(70, 150), (100, 15), (257, 12), (1000, 6), (2048, 4)])
def test_one_workgroup_per_problem_equals_lock_step_and_individual_solves(n, count):
    """MPRGP's single-launch form, forced on and off, against solve(): tight boxes (multi-pass feasible-step
    searches), warm starts, mixed tolerances, budgets."""
    probs = _problems(n, count, 5300 + n, 0.1)
    tols = [[1e-7, 1e-5, 1e30][p % 3] for p in range(count)]
    ref = _reference(probs, tols, 3000)
    for single in (True, False):
        got = _batch(probs, tols, 3000, single_launch=single)
        for a, c in zip(ref, got):
            _same(a, c)
    if count > 20:
        return
    longest = max(r.solution_num_matrix_vector_multiplications for r in ref)
    for p in range(0, count, 2):
        A, b, _, P = probs[p]
        probs[p] = (A, b, ref[p].solution.clone(), P)
    for cap in (2, 3, 4, 5, 9, longest - 1, longest, longest + 1):
        ref2 = _reference(probs, tols, cap)
        got = _batch(probs, tols, cap, single_launch=True)
        for a, c in zip(ref2, got):
            _same(a, c)


def test_nan_data_take_the_reference_exits():
    """A NaN residual at the start fails `res >= tol` and skips the loop (BBPGD, MPRGP); APGD's line search never
    accepts and runs into the budget.  Same exits in lock-step, next to healthy problems of the same batch."""
    import torch
    from ccqppy_amd import batch, solvers
    n = 160
    probs = _problems(n, 4, 90)
    A, b, x0, P = probs[1]
    bad = b.clone()
    bad[7] = float("nan")
    probs[1] = (A, bad, x0, P)
    for cls, cap in ((solvers.CCQPSolverMPRGP, 400), (solvers.CCQPSolverBBPGD, 400), (solvers.CCQPSolverAPGD, 14)):
        ref = []
        for (A_, b_, x0_, P_) in probs:
            with contextlib.redirect_stdout(io.StringIO()):
                ref.append(cls(1e-7, cap).solve(A_, b_, x0=x0_, convex_proj_op=P_))
        got = batch.solve_batch_pipelined(lambda: cls(1e-7, cap), probs)
        for i, (a, c) in enumerate(zip(ref, got)):
            assert a.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
            assert bool(a.solution_converged) == bool(c.solution_converged)
            assert bool(np.isnan(a.solution_residual)) == bool(np.isnan(c.solution_residual))
            assert torch.equal(torch.isnan(a.solution), torch.isnan(c.solution))
            if i != 1:
                _same(a, c)


def test_single_solve_pipeline_option():
    """CCQPSolverMPRGP(..., pipeline=True) = the lock-step scheduler with a batch of one; where it does not apply
    (scalar bounds raise as in the reference, spheres, budget reached) the host-driven loop runs."""
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    for n in (100, 1024, 3000):
        for p, (A, b, x0, P) in enumerate(_problems(n, 5, 40 + n, 0.1)):
            for cap in (2000, 9, 21):
                from ccqppy_amd import lockstep
                before = lockstep.stats["fallback_solves"]
                with contextlib.redirect_stdout(io.StringIO()):
                    a = solvers.CCQPSolverMPRGP(1e-7, cap, pipeline=False).solve(A, b, x0=x0, convex_proj_op=P)
                    c = solvers.CCQPSolverMPRGP(1e-7, cap, pipeline=True).solve(A, b, x0=x0, convex_proj_op=P)
                    d = solvers.CCQPSolverMPRGP(1e-7, cap).solve(A, b, x0=x0, convex_proj_op=P)      # the default: None
                _same(a, c)
                _same(a, d)
                assert cap < 2000 or lockstep.stats["fallback_solves"] == before   # really solved on the device
    # the default form gives the device 1000 matvecs: a solve that stalls short of its tolerance goes to the
    # host-driven loop after them, with the budget's own exit
    from ccqppy_amd import lockstep
    A, b, x0, P = _problems(256, 1, 77, 0.1)[0]
    before = lockstep.stats["fallback_solves"]
    with contextlib.redirect_stdout(io.StringIO()):
        a = solvers.CCQPSolverMPRGP(1e-13, 1300, pipeline=False).solve(A, b, convex_proj_op=P)
        d = solvers.CCQPSolverMPRGP(1e-13, 1300).solve(A, b, convex_proj_op=P)
    _same(a, d)
    assert not a.solution_converged and lockstep.stats["fallback_solves"] == before + 1
    op, b, _ = problems.random_problem(256, 3)
    with contextlib.redirect_stdout(io.StringIO()):
        a = solvers.CCQPSolverMPRGP(1e-7, 500).solve(op, b, convex_proj_op=ss.SphereProjOp(256, 2.0))
        c = solvers.CCQPSolverMPRGP(1e-7, 500, pipeline=True).solve(op, b, convex_proj_op=ss.SphereProjOp(256, 2.0))
    _same(a, c)
    with pytest.raises(TypeError):
        with contextlib.redirect_stdout(io.StringIO()):
            solvers.CCQPSolverMPRGP(1e-7, 500, pipeline=True).solve(op, b, convex_proj_op=ss.BoxProjOp(256, -0.5, 0.5))


def test_rejects_scalar_bounds_and_takes_tiny_problems():
    """Scalar bounds raise (the reference's normal_vector indexes them).  Tiny problems (n <= 64, where the host-driven
    loop takes the literal kernels and BoxProjOp.normal_vector sums |x - P(x)|^2 sequentially) are taken since round 4:
    lock-step rounds, the one-launch form and pipeline=True give the bits and counts of solve()."""
    from ccqppy_amd import batch, lockstep, problems, solvers, solution_spaces as ss
    op, b, _ = problems.random_problem(128, 1)
    with pytest.raises(TypeError):
        batch.solve_batch_pipelined(lambda: solvers.CCQPSolverMPRGP(1e-8), [(op, b, None, ss.BoxProjOp(128, -0.5, 0.5))])
    for n in (48, 64, 17):
        items = []
        for p in range(5):
            op, b, _ = problems.random_problem(n, 40 + p)
            items.append((op, b, None, ss.BoxProjOp(n, -0.3 * np.ones(n) - 0.01 * p, 0.35 * np.ones(n))))
        for single in (True, False):
            before = lockstep.stats["fallback_solves"]
            got = batch.solve_batch_pipelined(lambda: solvers.CCQPSolverMPRGP(1e-6, 3000), items, single_launch=single)
            assert lockstep.stats["fallback_solves"] == before
            for it, c in zip(items, got):
                with contextlib.redirect_stdout(io.StringIO()):
                    one = solvers.CCQPSolverMPRGP(1e-6, 3000).solve(it[0], it[1], convex_proj_op=it[3])
                    two = solvers.CCQPSolverMPRGP(1e-6, 3000, pipeline=True).solve(it[0], it[1], convex_proj_op=it[3])
                _same(one, c)
                _same(one, two)


@pytest.mark.parametrize("n,count", [(4096, 9), (4098, 8), (6000, 4)])
def test_default_batched_matvec_is_the_persistent_lds_form_and_changes_no_bit(n, count):
    """From n = 4096 and 1 GiB of matrices on, the batched matvec of lock-step MPRGP is ONE persistent LDS-DMA launch whose
    waves walk the live problems' matrices and read their 4 rows about 1 MiB apart (bmprgp_gemv_pers_kernel,
    gemv_lds_row_stride) -- and the run LOOKS AHEAD: the A.dx launch of an iteration also computes A.x_{k+1} + b and
    A.p_{k+1} (one pass over A with three right-hand sides instead of three passes per CG step; the start's A.x_0 + b
    serves the first loop top).  An explicit rows-x-steps field in the tuning word keeps the VALU form, one matvec per
    launch.  Same bits, same reported and executed MV counts, and those of each problem's own solve()."""
    from ccqppy_amd._engine import get_engine
    probs = _problems(n, count, 6100 + n)
    tols = [1e-6] * count
    eng = get_engine(0)
    import ctypes
    lib = eng.lib

    def timed(**kw):
        lib.ccqp_mv_timing_begin(4096)
        out = _batch(probs, tols, 3000, single_launch=False, **kw)
        buf, cnt = (ctypes.c_double * 4096)(), ctypes.c_int()
        lib.ccqp_mv_timing_end(buf, 4096, ctypes.byref(cnt))
        return out, cnt.value

    lds, launches_lds = timed()
    valu, launches_valu = timed(tune=3 << 4)
    assert launches_lds > 5 and launches_valu > 5
    for a, c in zip(valu, lds):
        assert c.solution_converged
        _same(a, c)
    ref = _reference(probs[:2], tols[:2], 3000)
    for a, c in zip(ref, lds[:2]):
        _same(a, c)


@pytest.mark.parametrize("n,count", [(100, 9), (256, 6), (777, 3), (1500, 4), (2048, 1)])
def test_fused_rounds_of_small_batches_change_no_bit(n, count):
    """A small batch of one-chunk problems (<= 256 MiB of matrices): a launch of the stage kernel passes THROUGH the rounds
    whose product is already there (A.x_k + b and A.p computed ahead by the A.dx launch as grid slices of one VALU launch; a
    feasibility search that repeats) -- two launches per CG step instead of six; the device enforces the budget (slab slot
    55).  Against the unfused rounds (context option SMALL_SHARING = 1) and each problem's own host-driven solve: x bit
    for bit, reported and executed counts, residual, converged flag -- wide and tight boxes (expansion steps, repeated
    feasibility searches), rows 16-byte aligned and not (n = 777), budgets from a handful of matvecs to past convergence."""
    from ccqppy_amd import _lib, lockstep, solvers
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    fl = int(n % 2 == 0) | _lib.LOCKSTEP_FUSED_ROUNDS
    for width in (0.3, 0.04):
        probs = _problems(n, count, 6100 + n, width)
        tols = [1e-7] * count
        with contextlib.redirect_stdout(io.StringIO()):
            full = [solvers.CCQPSolverMPRGP(1e-7, 3000, pipeline=False).solve(A, b, x0=x0, convex_proj_op=P)
                    for A, b, x0, P in probs]
        longest = max(r.solution_num_matrix_vector_multiplications for r in full)
        for cap in (3000, 4, 7, 12, longest - 1, longest, longest + 2):
            with contextlib.redirect_stdout(io.StringIO()):
                ref = [solvers.CCQPSolverMPRGP(1e-7, cap, pipeline=False).solve(A, b, x0=x0, convex_proj_op=P)
                       for A, b, x0, P in probs]
            got = {}
            try:
                for off in (1, 0):
                    eng.set_option("SMALL_SHARING", off)
                    assert eng.lib.ccqp_bmprgp_fused_form(eng.ctx, count, n, fl) == (0 if off else 2)
                    got[off] = _batch(probs, tols, cap, single_launch=False)
            finally:
                eng.set_option("SMALL_SHARING", 0)
            for a, c, d in zip(ref, got[0], got[1]):
                _same(a, c)
                _same(a, d)
        A, b, x0, P = probs[0]
        with contextlib.redirect_stdout(io.StringIO()):
            one = solvers.CCQPSolverMPRGP(1e-7, 3000, pipeline=True).solve(A, b, x0=x0, convex_proj_op=P)
        _same(full[0], one)
    assert max(r.executed_matrix_vector_multiplications - r.solution_num_matrix_vector_multiplications for r in full) > 0
    assert eng.lib.ccqp_bmprgp_fused_form(eng.ctx, 4, 4096, 1 | _lib.LOCKSTEP_FUSED_ROUNDS) == 0      # two chunks
    assert eng.lib.ccqp_bmprgp_fused_form(eng.ctx, 100, 1024, 1 | _lib.LOCKSTEP_FUSED_ROUNDS) == 2    # (no byte limit ...)
    assert eng.lib.ccqp_bmprgp_fused_form(eng.ctx, 128, 1024, 1 | _lib.LOCKSTEP_FUSED_ROUNDS) == 0    # ... up to the persistent form
    assert eng.lib.ccqp_bmprgp_fused_form(eng.ctx, 4, 1024, 1) == 0                                   # not asked for
