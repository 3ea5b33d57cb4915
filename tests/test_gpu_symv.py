"""GPU: the opt-in symmetric matvec (ccqp_symv_upper_f64 / operators.SymmetricOperator).
Non-parity by design (different summation order), so the bar is an fp64 error bound against the general
matvec and solver-level agreement (same MV counts, x within 1e-10) on the generator's SPD problems."""
import contextlib
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [128, 256, 1024, 8192, 8320, 16384 + 128])
def test_symv_matches_general_matvec(n):
    from ccqppy_amd import problems
    from ccqppy_amd._engine import get_engine
    from ccqppy_amd.operators import SymmetricOperator
    eng = get_engine(0)
    A = problems.spd_rows(n, 5)
    x = torch.randn(n, dtype=torch.float64, device="cuda")
    b = torch.randn(n, dtype=torch.float64, device="cuda")
    y0, g0 = eng.empty(n), eng.empty(n)
    eng.gemv(A, x, y0, b, g0)
    op = SymmetricOperator(A, use_general=False)       # force the triangle kernel at every size
    y1, g1 = eng.empty(n), eng.empty(n)
    op.matvec(x, y1, b, g1)
    bound = 8 * np.finfo(float).eps * np.sqrt(n) * (A.abs() @ x.abs())
    assert bool(((y1 - y0).abs() <= bound + 1e-300).all())
    assert torch.equal(g1, y1 + b)
    y2 = eng.empty(n)
    op.matvec(x, y2)
    assert torch.equal(y2, y1)                     # deterministic, with or without the epilogue
    # only the upper triangle is read: garbage below the diagonal must not matter
    Abad = A.clone()
    Abad[np.tril_indices(n, -1)] = float("nan") if n <= 1024 else Abad[np.tril_indices(n, -1)]
    if n <= 1024:
        op2 = SymmetricOperator(Abad, check=False, use_general=False)
        y3 = eng.empty(n)
        op2.matvec(x, y3)
        assert torch.equal(y3, y1)


def test_symmetric_operator_small_n_uses_general_kernel():
    from ccqppy_amd import problems
    from ccqppy_amd.operators import SymmetricOperator
    assert SymmetricOperator(problems.spd_rows(1024, 1)).use_general
    assert not SymmetricOperator(problems.spd_rows(16384, 1)).use_general


def test_symmetric_operator_rejects_bad_input():
    from ccqppy_amd import problems
    from ccqppy_amd.operators import SymmetricOperator
    with pytest.raises(ValueError):
        SymmetricOperator(problems.spd_rows(100, 1))
    A = problems.spd_rows(256, 1)
    A[3, 200] += 1.0
    with pytest.raises(ValueError):
        SymmetricOperator(A)


@pytest.mark.parametrize("sname,args", [("APGD", (1e-8, 5000)), ("BBPGD", (1e-8, 5000)), ("MPRGP", (1e-6, 300))])
def test_solvers_on_symmetric_operator(sname, args):
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd.operators import SymmetricOperator
    n = 4096
    op, b, _ = problems.random_problem(n, 1234)
    P = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    cls = getattr(solvers, "CCQPSolver" + sname)
    with contextlib.redirect_stdout(io.StringIO()):
        dense = cls(*args).solve(op, b, convex_proj_op=P)
        sym = cls(*args).solve(SymmetricOperator(op.A, use_general=False), b, convex_proj_op=P)
    assert sym.solution_num_matrix_vector_multiplications == dense.solution_num_matrix_vector_multiplications
    assert sym.solution_converged
    xd, xs = dense.solution.cpu().numpy(), sym.solution.cpu().numpy()
    assert np.linalg.norm(xd - xs) <= 1e-10 * np.linalg.norm(xd)
