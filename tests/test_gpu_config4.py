"""GPU: BASELINE config 4 (n = 262144, SPG + DisjointUnion(Box, Cone), row-block sharded over 8 GPUs)
checked on ONE GPU:
  * the last rank's slab (rows 229376..262143, 64 GiB): generator bits vs the CPU statement at offsets
    beyond 2^35 elements, GEMV vs an independent fp64 matvec;
  * the whole solve with the 8 row blocks visited one after the other -- each block is REGENERATED in
    place (64 GiB in ~20 ms) and multiplied by the production GEMV, which is exactly the arithmetic 8
    ranks perform (the row-sum bits do not depend on the sharding); only the RCCL exchange is absent.
    512 GiB never has to exist at once.  Size-independent properties are asserted."""
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import gen_ref
from tests import helpers as H

pytestmark = pytest.mark.gpu

N = 262144
G = 8


def test_last_rank_slab_generator_and_gemv():
    from ccqppy_amd import problems
    from ccqppy_amd._engine import get_engine
    from ccqppy_amd.operators import row_partition
    eng = get_engine(0)
    torch.cuda.empty_cache()
    row0, rows, _ = row_partition(N, G, G - 1)
    assert (row0, rows) == (229376, 32768)
    A = problems.spd_rows(N, 1234, row0, rows)
    for r in (0, 1, 12345, rows - 1):               # spot rows against the CPU statement (offsets > 2^35)
        want = gen_ref.spd_rows(N, 1234, row0 + r, 1)[0]
        assert H.same_bits(A[r].cpu().numpy(), want)
    x = problems.xstar(N, 1234)
    y = eng.empty(rows)
    eng.gemv(A, x, y)
    worst = 0.0
    for r0 in range(0, rows, 4096):
        ref = A[r0:r0 + 4096] @ x
        worst = max(worst, float((ref - y[r0:r0 + 4096]).abs().max()))
    assert worst < 1e-11
    del A
    torch.cuda.empty_cache()


def test_config4_solve_by_regenerated_row_blocks():
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    torch.cuda.empty_cache()
    from ccqppy_amd.operators import RegeneratedRowBlocksOperator
    op = RegeneratedRowBlocksOperator(N, 1234, G, eng)
    xs = problems.xstar(N, 1234)
    Ax = eng.empty(N)
    op.matvec(xs, Ax)
    b = eng.empty(N)
    eng.scale(-1.0, Ax, b)
    half = N // 2
    P = ss.DisjointProjOp(ss.BoxProjOp(half, -0.5, 0.5), ss.ConeProjOp(N - half))
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        r = solvers.CCQPSolverSPG(1e-8, 64).solve(op, b, convex_proj_op=P)
    assert r.solution_num_matrix_vector_multiplications == 64 and not r.solution_converged   # fixed budget
    assert r.executed_matrix_vector_multiplications == 64
    x = r.solution
    assert x.shape == (N,) and bool(torch.isfinite(x).all())
    # SPG iterates are convex combinations of feasible points of the Box half
    assert float(x[:half].max()) <= 0.5 + 1e-15 and float(x[:half].min()) >= -0.5 - 1e-15
    # monotone-ish progress: the objective at x is far below the objective at x0 = 0 (which is 0)
    Axk = eng.empty(N)
    op.matvec(x, Axk)
    f = 0.5 * float(torch.dot(x, Axk)) + float(torch.dot(b, x))
    assert f < -1000.0
    # same run again -> bitwise identical (deterministic reductions, seeded RNG)
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        r2 = solvers.CCQPSolverSPG(1e-8, 64).solve(op, b, convex_proj_op=P)
    assert torch.equal(r2.solution, x)
    del op
    torch.cuda.empty_cache()
