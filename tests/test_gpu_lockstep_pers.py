"""The batched matvec of every lock-step solver as ONE persistent LDS-DMA launch (bp_gemv_pers_kernel, the tuned default
from n = 4096 and 512 MiB of matrices on) against the batched VALU kernels (an explicit rows-x-steps field in the tuning
word keeps them): the same bits, MV counts and residuals per problem, and those of each problem's own solve()."""
import contextlib
import ctypes
import io

import numpy as np
import pytest

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


def _form(count, n, flags):
    from ccqppy_amd._engine import get_engine
    buf = ctypes.create_string_buffer(64)
    assert get_engine(0).lib.ccqp_bpipe_matvec_form(get_engine(0).ctx, count, n, flags, buf, 64) == 0
    return buf.value.decode()


def test_which_batches_take_the_persistent_form():
    assert _form(8, 4096, 1) == "bp_gemv_pers_kernel"
    assert _form(128, 8192, 1) == "bp_gemv_pers_kernel"
    assert _form(128, 8192, 1 | 8) == "bp_gemv_pers_kernel"          # (bits 3 / 7: solver variants, not tuning)
    assert _form(128, 8192, 1 | (3 << 4)) == "batched VALU kernel"    # an explicit rows-x-steps field
    assert _form(128, 8192, 0) == "batched VALU kernel"               # unaligned operands
    assert _form(3, 4096, 1) == "batched VALU kernel"                 # below 512 MiB
    assert _form(4, 4096, 1) == "bp_gemv_pers_kernel" and _form(1, 8192, 1) == "bp_gemv_pers_kernel"
    assert _form(512, 2048, 1) == "batched VALU kernel"               # n < 4096
    assert _form(8, 4097, 1) == "batched VALU kernel"                 # odd n
    assert _form(600, 4096, 1) == "bp_gemv_pers_kernel"               # (more than 512 problems: one launch per 512)
    assert _form(8, 8192, 1 | (3 << 1) | (4 << 4)) == "bp_gemv_pers_kernel"     # MPRGP's explicit ring codes
    # the runs that share passes between products (lock-step APGD / MPRGP, bit 16 of the query): from n = 1024 on
    assert _form(512, 2048, 1 | (1 << 16)) == "bp_gemv_pers_kernel"
    assert _form(128, 1024, 1 | (1 << 16)) == "bp_gemv_pers_kernel"
    assert _form(127, 1024, 1 | (1 << 16)) == "batched VALU kernel"   # below 1 GiB
    assert _form(2048, 512, 1 | (1 << 16)) == "batched VALU kernel"   # n < 1024
    assert _form(128, 1024, 1 | (1 << 16) | (3 << 4)) == "batched VALU kernel"


@pytest.mark.parametrize("kind,segmented", [("BBPGD", False), ("BBPGDf", False), ("PGD", False), ("APGD", False),
                                            ("APGDAntiRelaxation", False), ("MPRGP", False), ("SPG", False),
                                            ("BBPGD", True), ("APGD", True), ("MPRGP", True), ("SPG", True)])
def test_persistent_batched_matvec_changes_no_bit(kind, segmented):
    from ccqppy_amd import lockstep, problems, solvers, solution_spaces as ss
    n, count = 4096, 8                                    # 8 n^2 count = 1 GiB (the form is taken from 512 MiB on)
    assert _form(count, n, 1) == "bp_gemv_pers_kernel"
    probs = []
    for p in range(count):
        op, b, _ = problems.random_problem(n, 7300 + p)
        if segmented:
            h = n // 2
            first = ss.SphereProjOp(h, 0.8 + 0.01 * p)
            P = ss.DisjointProjOp(first, ss.BoxProjOp(n - h, -0.4 * np.ones(n - h), 0.5 * np.ones(n - h)))
        else:
            P = ss.BoxProjOp(n, -0.5 * np.ones(n), (0.4 + 0.005 * p) * np.ones(n))
        probs.append((op, b, None, P))
    cls = getattr(solvers, "CCQPSolver" + kind)
    budget = 400 if kind in ("APGD", "APGDAntiRelaxation", "MPRGP") else 2000
    tol = 1e-6

    def new_solver(p):
        if kind == "PGD":
            return cls(1e-4, 600, 0.3)
        s = cls(tol, budget)
        if kind == "SPG":
            s.rng = np.random.RandomState(900 + p)        # (lock-step SPG: one generator per solver)
        return s

    def run(tune):
        order = iter(range(count))
        fb0 = lockstep.stats["fallback_solves"]
        out = lockstep.solve_batch_pipelined(lambda: new_solver(next(order)), probs, device=0, tune=tune)
        assert lockstep.stats["fallback_solves"] == fb0
        return out

    pers = run(0)
    valu = run(1 << 4)
    for a, c in zip(valu, pers):
        _same(a, c)
    assert max(r.solution_num_matrix_vector_multiplications for r in pers) > 5
    # and the plain per-problem solve of the first two
    for p in range(2):
        A, b, x0, P = probs[p]
        s = new_solver(p)
        with contextlib.redirect_stdout(io.StringIO()):
            s.solve(A, b, x0=x0, convex_proj_op=P)
        _same(s, pers[p])


def test_more_problems_than_one_persistent_launch_takes():
    """The persistent launch holds the operands of at most 512 problems in LDS; a larger batch is multiplied by one launch
    per 512 problems.  520 problems of n = 4096 (65 GiB): the bits of the VALU kernels, and of the first / last
    problem's own solve()."""
    import torch
    from ccqppy_amd import lockstep, problems, solvers, solution_spaces as ss
    if torch.cuda.mem_get_info(0)[0] < 90 * 2 ** 30:
        pytest.skip("needs 90 GiB of free device memory")
    n, count = 4096, 520
    P = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    probs = []
    for p in range(count):
        op, b, _ = problems.random_problem(n, 8800 + p)
        probs.append((op, b, None, P))
    make = lambda: solvers.CCQPSolverBBPGD(1e-8, 2000)        # noqa: E731
    pers = lockstep.solve_batch_pipelined(make, probs, device=0)
    valu = lockstep.solve_batch_pipelined(make, probs, device=0, tune=3 << 4)
    for a, c in zip(valu, pers):
        assert c.solution_converged
        _same(a, c)
    for p in (0, 511, 512, count - 1):
        A, b, x0, Pp = probs[p]
        with contextlib.redirect_stdout(io.StringIO()):
            s = make().solve(A, b, x0=x0, convex_proj_op=Pp)
        _same(s, pers[p])


@pytest.mark.parametrize("kind", ["APGD", "APGDAntiRelaxation", "MPRGP"])
@pytest.mark.parametrize("segmented", [False, True])
@pytest.mark.parametrize("n,count", [(4096, 12), (2048, 36), (1024, 130), (1024, 300)])     # (300: two three-vector launches)
def test_look_ahead_with_problems_far_out_of_step(kind, segmented, n, count):
    """Problems whose boxes are 0.02 ... 8 wide (and start points partly outside them): they take very different numbers
    of iterations, CG and expansion steps, line-search rejections -- so in most rounds some problems find their product
    computed ahead, some multiply, some have finished.  The look-ahead run (default from n = 1024 and 1 GiB of matrices
    on) against one matvec per launch (the VALU kernels), bit for bit, and against each problem's own solve()."""
    import torch
    from ccqppy_amd import lockstep, problems, solvers, solution_spaces as ss
    assert _form(count, n, 1 | (1 << 16)) == "bp_gemv_pers_kernel"
    widths = np.geomspace(0.02, 8.0, count)
    if count > 256:
        widths = widths[(np.arange(count) * 7) % count]       # (fast and slow problems on both sides of the launch split)
    probs = []
    for p in range(count):
        op, b, _ = problems.random_problem(n, 9100 + p)
        w = float(widths[p])
        if segmented:
            h = n // 2
            P = ss.DisjointProjOp(ss.SphereProjOp(h, 4.0 * w), ss.BoxProjOp(n - h, -w * np.ones(n - h), 1.5 * w * np.ones(n - h)))
        else:
            P = ss.BoxProjOp(n, -w * np.ones(n), 1.5 * w * np.ones(n))
        x0 = None if p % 3 else torch.full((n,), 0.7, dtype=torch.float64, device="cuda")
        probs.append((op, b, x0, P))
    cls = getattr(solvers, "CCQPSolver" + kind)
    make = (lambda: cls(1e-6, 400)) if kind == "MPRGP" else (lambda: cls(1e-8, 600))
    fb0 = lockstep.stats["fallback_solves"]
    look = lockstep.solve_batch_pipelined(make, probs, device=0)
    plain = lockstep.solve_batch_pipelined(make, probs, device=0, tune=3 << 4)
    assert lockstep.stats["fallback_solves"] == fb0
    for a, c in zip(plain, look):
        _same(a, c)
    counts = [r.solution_num_matrix_vector_multiplications for r in look]
    assert max(counts) - min(counts) >= 4, counts                      # (the case is meant to be heterogeneous)
    for p in (0, count // 2, count - 1):
        A, b, x0, P = probs[p]
        with contextlib.redirect_stdout(io.StringIO()):
            s = make().solve(A, b, x0=x0, convex_proj_op=P)
        _same(s, look[p])


@pytest.mark.parametrize("kind", ["APGD", "APGDAntiRelaxation"])
def test_fused_rounds_enforce_the_budget_on_the_device(kind):
    """Lock-step APGD with FUSED rounds (an accepting round also takes A.y_{k+1} + b over: a round may count two matvecs,
    so the host's round count no longer bounds the matvec count and the device enforces max_mv itself, slab slot 48).
    Problems of n = 256 through the persistent look-ahead rounds (context option PERS1_MIN_N = 128), budgets from 3 up
    to past the matvec count of convergence, three tolerances: every result -- x bit for bit, reported and executed
    counts, converged flag, residual -- equals the problem's own host-driven solve() (whose budget exits the reference
    goldens pin), whether the device finished the problem or stopped it one short of the budget for the host loop;
    with two matvecs of room past convergence the device finishes every problem itself."""
    from ccqppy_amd import lockstep, problems, solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    cls = getattr(solvers, "CCQPSolver" + kind)
    n = 256
    eng = get_engine(0)
    items = []
    for p in range(4):
        op, b, _ = problems.random_problem(n, 9100 + p)
        items.append((op, b, None, ss.BoxProjOp(n, -0.3 * np.ones(n) - 0.02 * p, 0.4 * np.ones(n))))
    eng.set_option("PERS1_MIN_N", 128)
    try:
        assert _form(len(items), n, 1 | (1 << 16)) == "bp_gemv_pers_kernel"
        for tol in (1e-3, 1e-6, 1e-9):
            conv = []
            for it in items:
                with contextlib.redirect_stdout(io.StringIO()):
                    conv.append(cls(tol, np.inf).solve(it[0], it[1], convex_proj_op=it[3]).solution_num_matrix_vector_multiplications)
            budgets = sorted(set(list(range(3, 9)) + [m + d for m in conv for d in (-2, -1, 0, 1, 2, 3)]))
            for max_mv in budgets:
                before = lockstep.stats["fallback_solves"]
                got = lockstep.solve_batch_pipelined(lambda: cls(tol, max_mv), items, device=0, single_launch=False)
                fell = lockstep.stats["fallback_solves"] - before
                for it, c, m in zip(items, got, conv):
                    with contextlib.redirect_stdout(io.StringIO()):
                        try:
                            one = cls(tol, max_mv).solve(it[0], it[1], convex_proj_op=it[3])
                        except UnboundLocalError:
                            one = None
                    if one is None:
                        continue
                    _same(one, c)
                if max_mv >= max(conv) + 3:
                    assert fell == 0, (tol, max_mv, conv, fell)
    finally:
        eng.set_option("PERS1_MIN_N", 0)


@pytest.mark.parametrize("kind", ["APGD", "APGDAntiRelaxation"])
def test_fused_rounds_below_the_persistent_sizes_change_no_bit(kind):
    """A handful of problems below the persistent matvec's sizes (n < 4096): fused rounds whose two products come from ONE
    launch of the batched VALU kernel (bp_gemv3_valu_kernel, blockIdx.z = the product) -- one round per accepted
    iteration instead of two -- against the unfused rounds (context option SMALL_SHARING = 1) and the host-driven solve(): x
    bit for bit, counts, residual.  Sizes with one and two reduction chunks, rows 16-byte aligned and not (n = 777),
    vector and scalar bounds, a budget that stops the device one short (AP_DONE = 2) and one that lets it finish; single
    solves (pipeline=True) and a batch of five problems out of step with each other; and the same single solves at
    n >= 4096, where the fused rounds are the persistent form's."""
    from ccqppy_amd import _lib, lockstep, problems, solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    cls = getattr(solvers, "CCQPSolver" + kind)
    eng = get_engine(0)
    fl = 1 | _lib.LOCKSTEP_LOOKAHEAD | _lib.LOCKSTEP_FUSED_ROUNDS
    form = lambda count, n, f=fl: eng.lib.ccqp_bapgd_fused_form(eng.ctx, count, n, f)       # noqa: E731
    try:
        for n, scalar_bounds in ((130, False), (300, True), (777, False), (1500, False), (2500, True), (3072, False),
                                 (5000, False), (8192, True)):
            op, b, _ = problems.random_problem(n, 4400 + n)
            P = ss.BoxProjOp(n, -0.35, 0.45) if scalar_bounds else \
                ss.BoxProjOp(n, -0.3 * np.ones(n) - 1e-5 * np.arange(n), 0.4 * np.ones(n))
            with contextlib.redirect_stdout(io.StringIO()):
                ref = cls(1e-7, np.inf, pipeline=False)
                ref.solve(op, b, convex_proj_op=P)
            conv = ref.solution_num_matrix_vector_multiplications
            assert conv > 8
            for max_mv in (np.inf, conv + 3, conv - 3):
                with contextlib.redirect_stdout(io.StringIO()):
                    host = cls(1e-7, max_mv, pipeline=False)
                    host.solve(op, b, convex_proj_op=P)
                for pair in (0, 1):
                    eng.set_option("SMALL_SHARING", pair)
                    aligned = int(n % 2 == 0)
                    want = 1 if (n >= 4096 and aligned) else (2 if pair == 0 else 0)
                    assert form(1, n, (fl & ~1) | aligned) == want, (n, pair)
                    before = dict(lockstep.stats)
                    with contextlib.redirect_stdout(io.StringIO()):
                        dev = cls(1e-7, max_mv, pipeline=True)
                        dev.solve(op, b, convex_proj_op=P)
                    _same(host, dev)
                    if max_mv >= conv + 3:
                        assert lockstep.stats["fallback_solves"] == before["fallback_solves"], (n, max_mv, pair)
            del op
        n = 512
        items = []
        for p in range(5):
            op, b, _ = problems.random_problem(n, 4600 + p)
            items.append((op, b, None, ss.BoxProjOp(n, (-0.3 - 0.1 * p) * np.ones(n), (0.4 - 0.05 * p) * np.ones(n))))
        got = {}
        for pair in (1, 0):
            eng.set_option("SMALL_SHARING", pair)
            assert form(5, n) == (0 if pair == 1 else 2)
            got[pair] = lockstep.solve_batch_pipelined(lambda: cls(1e-8, 500), items, device=0, single_launch=False)
        for a, c in zip(got[1], got[0]):
            _same(a, c)
        with contextlib.redirect_stdout(io.StringIO()):
            one = cls(1e-8, 500, pipeline=False)
            one.solve(items[2][0], items[2][1], convex_proj_op=items[2][3])
        _same(one, got[0][2])
    finally:
        eng.set_option("SMALL_SHARING", 0)
    # the defaults
    assert form(1, 4096) == 1 and form(128, 4096) == 1 and form(1, 2048) == 2 and form(32, 1024) == 2
    assert form(33, 1024) == 0                       # more than 256 MiB of small matrices: unfused rounds
    assert form(512, 2048) == 1                      # ... until the persistent form takes it
    assert form(1, 4096, 1) == 0                     # fused rounds not asked for
    eng.set_option("SMALL_SHARING", 2)
    try:
        assert form(64, 1024) == 2
    finally:
        eng.set_option("SMALL_SHARING", 0)
