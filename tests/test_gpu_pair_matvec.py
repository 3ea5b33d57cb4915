"""Two matvecs in one pass over A (ccqp_gemv2_f64 / gemv_lds2_kernel) and the APGD loop built on it: the second product
of an iteration, A.y_{k+1}, is formed in the pass that computes A.x_{k+1} (y_{k+1} does not depend on the line search,
solvers.py:313-316).  Everything must come out bit for bit as from single matvecs: products, iterates, residuals, reported
and executed MV counts -- also when the line search rejects a step or the matvec budget ends the solve."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
LDS = 0x51001800          # the LDS-staged form forced (the default only from 8 GiB on)


def _bits(t):
    return t.view(torch.int64)


@pytest.mark.parametrize("rows,cols", [(512, 512), (300, 1000), (1030, 258), (64, 130), (2048, 4096), (777, 2050), (4096, 4098)])
@pytest.mark.parametrize("with_g", [(True, True), (False, True), (True, False), (False, False)])
def test_pair_product_has_the_bits_of_two_single_products(rows, cols, with_g):
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    torch.manual_seed(rows * 7 + cols)
    A = torch.randn(rows, cols, dtype=torch.float64, device="cuda")
    x1, x2 = (torch.randn(cols, dtype=torch.float64, device="cuda") for _ in range(2))
    b = torch.randn(rows, dtype=torch.float64, device="cuda")
    ref = []
    for x, wg in ((x1, with_g[0]), (x2, with_g[1])):
        y, g = torch.empty_like(b), torch.empty_like(b)
        eng.gemv(A, x, y, b if wg else None, g if wg else None, variant=0xF)       # the VALU form
        ref.append((y, g if wg else None))
    for variant in (LDS, 0x11001800, 0):
        one_pass = variant != 0 or rows * cols >= 2 ** 24       # (the default: from 2^24 elements on)
        assert eng.gemv2_form(rows, cols, variant) == ("gemv_lds2_kernel" if one_pass else "two single matvecs")
        y1, y2 = torch.full_like(b, float("nan")), torch.full_like(b, float("nan"))
        g1 = torch.full_like(b, float("nan")) if with_g[0] else None
        g2 = torch.full_like(b, float("nan")) if with_g[1] else None
        mv0 = eng.mv_executed
        eng.gemv2(A, x1, x2, y1, y2, b if (with_g[0] or with_g[1]) else None, g1, g2, variant=variant)
        assert eng.mv_executed == mv0 + 2
        assert torch.equal(_bits(y1), _bits(ref[0][0])) and torch.equal(_bits(y2), _bits(ref[1][0]))
        if g1 is not None:
            assert torch.equal(_bits(g1), _bits(ref[0][1]))
        if g2 is not None:
            assert torch.equal(_bits(g2), _bits(ref[1][1]))


def test_pair_product_on_a_row_block_of_a_larger_matrix():
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    big = torch.randn(900, 1536, dtype=torch.float64, device="cuda")
    A = big[100:700, :1400]                     # lda 1536 > cols 1400, rows 600
    x1, x2 = (torch.randn(1400, dtype=torch.float64, device="cuda") for _ in range(2))
    y = [torch.empty(600, dtype=torch.float64, device="cuda") for _ in range(4)]
    eng.gemv(A, x1, y[0], variant=0xF)
    eng.gemv(A, x2, y[1], variant=0xF)
    eng.gemv2(A, x1, x2, y[2], y[3], variant=LDS)
    assert torch.equal(_bits(y[0]), _bits(y[2])) and torch.equal(_bits(y[1]), _bits(y[3]))


def _solve(n, seed, pair, tol=1e-8, max_mv=5000, width=0.5, variant=LDS, x0=None):
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    op, b, _ = problems.random_problem(n, seed)
    old = eng.gemv_variant
    eng.gemv_variant = variant
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            return solvers.CCQPSolverAPGD(tol, max_mv, pair_matvecs=pair).solve(
                op, b, x0=x0, convex_proj_op=ss.BoxProjOp(n, -width * np.ones(n), width * np.ones(n)))
    finally:
        eng.gemv_variant = old


def _same(a, c):
    assert a.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
    assert a.executed_matrix_vector_multiplications == c.executed_matrix_vector_multiplications
    assert bool(a.solution_converged) == bool(c.solution_converged)
    assert float(a.solution_residual) == float(c.solution_residual)
    assert torch.equal(_bits(a.solution), _bits(c.solution))


@pytest.mark.parametrize("n", [256, 1000, 2050, 4096])
def test_apgd_with_paired_matvecs_equals_apgd(n):
    for seed in (11, 12):
        one = _solve(n, seed, False)
        two = _solve(n, seed, True)
        auto = _solve(n, seed, None)            # (the default pairs from n = 4096 on)
        assert one.solution_converged and one.solution_num_matrix_vector_multiplications > 20
        _same(one, two)
        _same(one, auto)


@pytest.mark.parametrize("max_mv", [3, 4, 5, 6, 7, 12, 13])
def test_paired_matvecs_keep_the_budget_semantics(max_mv):
    """The matvec budget is checked where the reference checks it (:277, :292): the product computed ahead is counted when
    the next iteration takes it, not when the pass computes it."""
    one = _solve(777, 5, False, max_mv=max_mv)
    two = _solve(777, 5, True, max_mv=max_mv)
    assert not one.solution_converged
    _same(one, two)


def test_paired_matvecs_when_the_line_search_rejects_steps():
    """A start far outside the box and a tiny first Lipschitz estimate make the line search double L several times
    (:306-310): every rejected trial discards the product computed ahead."""
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    n = 1500
    op, b, _ = problems.random_problem(n, 77)
    A = op.A.clone()
    A[0, 0] += 400.0                             # one stiff direction the |A d| / |d| estimate at x0 - 1 underrates
    x0 = torch.ones(n, dtype=torch.float64, device="cuda")
    x0[0] = 1.0 + 1e-9
    P = ss.BoxProjOp(n, -5.0 * np.ones(n), 5.0 * np.ones(n))
    out = []
    old = eng.gemv_variant
    eng.gemv_variant = LDS
    try:
        for pair in (False, True):
            s = solvers.CCQPSolverAPGD(1e-8, 5000, pair_matvecs=pair)
            s.trace = []
            with contextlib.redirect_stdout(io.StringIO()):
                s.solve(A, b, x0=x0, convex_proj_op=P)
            out.append(s)
    finally:
        eng.gemv_variant = old
    assert sum(t["nback"] for t in out[0].trace) >= 2, "the case is meant to backtrack"
    assert [t["nback"] for t in out[0].trace] == [t["nback"] for t in out[1].trace]
    _same(out[0], out[1])


# ------------------------------------------------------------------ three products in one pass (ccqp_gemv3_f64)
@pytest.mark.parametrize("n", [128, 130, 1000, 1502, 4096])
@pytest.mark.parametrize("mask", [(1, 1, 1), (1, 1, 0), (1, 0, 1), (0, 1, 0)])
def test_triple_product_has_the_bits_of_single_products(n, mask):
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    torch.manual_seed(n)
    A = torch.randn(n, n, dtype=torch.float64, device="cuda")
    xs = [torch.randn(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    b = torch.randn(n, dtype=torch.float64, device="cuda")
    want_g = (False, False, True)
    ref = []
    for j in range(3):
        y, g = torch.empty_like(b), torch.empty_like(b)
        eng.gemv(A, xs[j], y, b if want_g[j] else None, g if want_g[j] else None, variant=0xF)
        ref.append((y, g))
    for variant in (LDS, 0):
        one_pass = variant != 0 or n * n >= 2 ** 24            # (the default: from 4096^2 on)
        assert eng.gemv3_form(n, variant) == ("bp_gemv3_pers_kernel" if one_pass else "single matvecs")
        ys = [torch.full_like(b, float("nan")) if mask[j] else None for j in range(3)]
        gs = [torch.full_like(b, float("nan")) if (mask[j] and want_g[j]) else None for j in range(3)]
        mv0 = eng.mv_executed
        eng.gemv3(A, [xs[j] if mask[j] else None for j in range(3)], ys, b, gs, variant=variant)
        assert eng.mv_executed == mv0 + sum(mask)
        for j in range(3):
            if mask[j]:
                assert torch.equal(_bits(ys[j]), _bits(ref[j][0]))
                if gs[j] is not None:
                    assert torch.equal(_bits(gs[j]), _bits(ref[j][1]))


def _solve_mprgp(n, seed, look, tol=1e-6, max_mv=300, width=0.5, variant=LDS):
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    op, b, _ = problems.random_problem(n, seed)
    old = eng.gemv_variant
    eng.gemv_variant = variant
    try:
        s = solvers.CCQPSolverMPRGP(tol, max_mv, look_ahead=look)
        s.trace = []
        with contextlib.redirect_stdout(io.StringIO()):
            s.solve(op, b, convex_proj_op=ss.BoxProjOp(n, -width * np.ones(n), width * np.ones(n)))
        return s
    finally:
        eng.gemv_variant = old


@pytest.mark.parametrize("n,width", [(256, 0.5), (1000, 0.5), (2050, 0.3), (4096, 0.5), (1500, 5.0)])
def test_mprgp_with_look_ahead_equals_mprgp(n, width):
    """CG steps (wide box: few active bounds) and expansion steps (narrow box) both occur over these cases."""
    branches = set()
    for seed in (21, 22):
        one = _solve_mprgp(n, seed, False, width=width)
        two = _solve_mprgp(n, seed, True, width=width)
        auto = _solve_mprgp(n, seed, None, width=width)
        assert one.solution_converged
        _same(one, two)
        _same(one, auto)
        assert [t["branch"] for t in one.trace] == [t["branch"] for t in two.trace]
        branches |= {t["branch"] for t in one.trace}
    assert branches & {"cg", "expand"}


@pytest.mark.parametrize("max_mv", [2, 3, 4, 5, 6, 7, 8, 9, 11, 14])
def test_look_ahead_keeps_the_budget_semantics(max_mv):
    for width in (0.5, 5.0):
        one = _solve_mprgp(900, 3, False, max_mv=max_mv, width=width)
        two = _solve_mprgp(900, 3, True, max_mv=max_mv, width=width)
        _same(one, two)
