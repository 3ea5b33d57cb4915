"""GPU parity of the sparse operator (ccqp_spmv_csr_f64 behind operators.SparseOperator).

  * the matvec against scipy's `A.dot(x)` (what the reference would call) and, bit for bit, against the
    summation order the header promises -- for every subgroup width, on irregular matrices with empty,
    short and very long rows;
  * the seven solvers on sparse A against the golden results of the REFERENCE run with scipy CSR input;
  * row-sharded sparse solves (2 ranks, one device, gloo) bitwise equal to the single-GPU solve;
  * index validation (the kernel does not bounds-check: bad input must be refused on the host)."""
import contextlib
import io
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import cases as C
from oracle import gen_ref
from tests import helpers as H

pytestmark = pytest.mark.gpu

CASES, ARR = H.load_sparse()
RTOL = 1e-10


def _ragged(rows, cols, seed, long_rows=()):
    """CSR with empty rows, rows of 1..90 entries in UNSORTED storage order, duplicates-free, plus a few
    very long rows."""
    rng = np.random.default_rng(seed)
    indptr, indices, data = [0], [], []
    for r in range(rows):
        if r in long_rows:
            m = min(cols, long_rows[r])
        else:
            m = 0 if r % 5 == 0 else int(rng.integers(1, min(cols, 90) + 1))
        c = rng.choice(cols, size=m, replace=False)
        indices.append(c)
        data.append(rng.normal(size=m))
        indptr.append(indptr[-1] + m)
    return (np.array(indptr, dtype=np.int64), np.concatenate(indices).astype(np.int32), np.concatenate(data))


@pytest.mark.parametrize("rows,cols,seed,long_rows", [(37, 50, 1, {}), (300, 211, 2, {7: 200, 150: 211}),
                                                     (1, 1, 3, {}), (513, 4096, 4, {100: 4000, 101: 129, 102: 64, 103: 65}),
                                                     (64, 3000, 5, {0: 1024, 1: 1025, 2: 2047, 3: 2049, 62: 3000, 63: 1500}),
                                                     (4000, 64, 6, {})])
def test_spmv_bits_follow_the_canonical_order_for_every_width(rows, cols, seed, long_rows):
    from ccqppy_amd.operators import SparseOperator
    indptr, indices, data = _ragged(rows, cols, seed, long_rows)
    x = np.random.default_rng(seed + 100).normal(size=cols)
    want = H.spmv_canonical(indptr, indices, data, x)
    ref = sp.csr_matrix((data, indices, indptr), shape=(rows, cols)).dot(x)
    np.testing.assert_allclose(want, ref, rtol=1e-12, atol=1e-13)
    b = np.random.default_rng(seed + 200).normal(size=rows)
    for width in (0, 4, 8, 16, 32, 64):
        op = SparseOperator((indptr, indices, data), shape=(rows, cols), width=width)
        eng = op.engine
        eng.bind()
        xd = eng.to_device(x)
        y = torch.full((rows,), 7.0, dtype=torch.float64, device=eng.device)
        g = torch.full((rows,), 7.0, dtype=torch.float64, device=eng.device)
        op.matvec(xd, y)
        assert H.same_bits(y.cpu().numpy(), want), width
        y.fill_(3.0)
        op.matvec(xd, y, eng.to_device(b), g)
        assert H.same_bits(y.cpu().numpy(), want), width
        assert H.same_bits(g.cpu().numpy(), want + b), width


def test_unknown_plan_summary_and_forced_kernels_give_the_same_bits():
    """C-ABI callers may pass summary = NULL (both extra code paths are then launched); the plan itself and
    the lean / transposed-capable kernel choice must not change a bit either."""
    from ccqppy_amd.operators import SparseOperator
    for A in (gen_ref.sparse_spd(4096, 64, 3), gen_ref.irregular_spd(3000, 4)):
        # one row far beyond the LDS path
        A = A.tolil()
        A[5, :] = np.random.default_rng(1).normal(size=A.shape[1])
        A = A.tocsr()
        x = np.random.default_rng(2).normal(size=A.shape[1])
        op = SparseOperator(A)
        assert op.summary[0] >= 1
        y0 = op.dot(x).cpu().numpy()
        np.testing.assert_allclose(y0, A.dot(x), rtol=1e-12, atol=1e-12)
        keep = op.summary
        op.summary = None
        assert H.same_bits(op.dot(x).cpu().numpy(), y0)
        import ctypes
        for fake in ((keep[0], 0, 0), (keep[0], 10 ** 9, 0)):
            op.summary = (ctypes.c_int64 * 3)(*fake)
            assert H.same_bits(op.dot(x).cpu().numpy(), y0)


def test_local_x_windows_are_used_where_columns_repeat_and_change_no_bit():
    """Banded / stencil-like matrices reuse their columns inside a 1792-entry window: the plan then switches
    those windows to the local-x mode (distinct x values through LDS, 16-bit local columns).  Same bits as the
    global-gather modes and as the canonical order; matrices without reuse keep their mode."""
    from ccqppy_amd import problems
    from ccqppy_amd.operators import SparseOperator
    n, side = 32768, 32
    offs = sorted({dz * side * side + dy * side + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)})
    i = np.arange(n)[:, None]
    cols = np.sort((i + np.array(offs)[None, :]) % n, axis=1)
    rng = np.random.default_rng(3)
    data = rng.normal(size=cols.shape)
    indptr = np.arange(n + 1, dtype=np.int64) * len(offs)
    x = rng.normal(size=n)
    on = SparseOperator((indptr, cols.reshape(-1).astype(np.int32), data.reshape(-1)), shape=(n, n))
    off = SparseOperator((indptr, cols.reshape(-1).astype(np.int32), data.reshape(-1)), shape=(n, n), local_x=False)
    assert on.summary[2] > 0.9 * (on.nnz // 1792) and off.summary[2] == 0
    y_on, y_off = on.dot(x).cpu().numpy(), off.dot(x).cpu().numpy()
    assert H.same_bits(y_on, y_off)
    rows = slice(1000, 1200)
    want = H.spmv_canonical(indptr[rows.start:rows.stop + 1] - indptr[rows.start], cols[rows].reshape(-1),
                            data[rows].reshape(-1), x)
    assert H.same_bits(y_on[rows], want)
    for width in (4, 8, 16, 32, 64):
        assert H.same_bits(SparseOperator((indptr, cols.reshape(-1).astype(np.int32), data.reshape(-1)), shape=(n, n),
                                          width=width).dot(x).cpu().numpy(), y_on)
    # no reuse (hashed circulant offsets): nothing switches
    circ = SparseOperator(problems.sparse_spd_csr(65536, 32, 1234), shape=(65536, 65536))
    assert circ.summary[2] == 0
    # transposes and row blocks re-plan for themselves
    assert H.same_bits(on.T.dot(x).cpu().numpy(), off.T.dot(x).cpu().numpy())
    assert H.same_bits(on.row_block(5000, 3000).dot(x).cpu().numpy(), y_on[5000:8000])


@pytest.mark.parametrize("k", [8, 16, 32, 100])
def test_diagonal_windows_are_used_for_repeated_offset_patterns_and_change_no_bit(k):
    """Windows whose rows all have one length and, position by position, one offset col - row (banded / circulant /
    regular-grid matrices without column reuse inside a window) keep their L offsets in the plan; the matvec then reads
    no column index for them.  Same products in the same order: the bits of the generic paths and of the canonical
    order.  Rows whose wrapped columns sort differently (near the ends of a circulant matrix) break the pattern: their
    windows stay generic."""
    from ccqppy_amd import problems
    from ccqppy_amd.operators import SparseOperator
    n = 65536
    csr = problems.sparse_spd_csr(n, k, 1234)
    on = SparseOperator(csr, shape=(n, n))
    off = SparseOperator(csr, shape=(n, n), diagonal_windows=False)
    nwin = on.nnz // 1792 + 1
    assert on.summary[2] == 0 and 0.8 * nwin < on.summary[3] < nwin      # most windows, not the wrapped ones
    assert off.summary[3] == 0
    rng = np.random.default_rng(k)
    x = rng.normal(size=n)
    y_on, y_off = on.dot(x).cpu().numpy(), off.dot(x).cpu().numpy()
    assert H.same_bits(y_on, y_off)
    rowptr, col, val = (t.cpu().numpy() for t in csr)
    for rows in (slice(0, 70), slice(30000, 30100), slice(n - 70, n)):
        want = H.spmv_canonical(rowptr[rows.start:rows.stop + 1] - rowptr[rows.start],
                                col[rowptr[rows.start]:rowptr[rows.stop]], val[rowptr[rows.start]:rowptr[rows.stop]], x)
        assert H.same_bits(y_on[rows], want)
    for width in (4, 8, 16, 32, 64):
        assert H.same_bits(SparseOperator(csr, shape=(n, n), width=width).dot(x).cpu().numpy(), y_on)
    # with the epilogue, through row blocks and the transpose (each re-plans for itself)
    b = rng.normal(size=n)
    eng = on.engine
    yd, gd = eng.empty(n), eng.empty(n)
    on.apply(eng, eng.to_device(x), yd, eng.to_device(b), gd)
    assert H.same_bits(gd.cpu().numpy(), y_on + b)
    assert H.same_bits(on.row_block(777, 5000).dot(x).cpu().numpy(), y_on[777:5777])
    assert H.same_bits(on.T.dot(x).cpu().numpy(), off.T.dot(x).cpu().numpy())
    # a banded matrix with an irregular row in the middle: that window must not be taken for a diagonal one
    m = 20000
    offs = np.array([-7, -3, -1, 0, 2, 5, 11])
    rws = np.arange(20, m - 20)
    cols = (rws[:, None] + offs[None, :]).astype(np.int32)
    data = rng.normal(size=cols.shape)
    cols[9000, 3] = 8990                                   # one entry off its diagonal (row stays sorted: 8990 < 8999 ...)
    cols[9000] = np.sort(cols[9000])
    ip = np.arange(len(rws) + 1, dtype=np.int64) * len(offs)
    band = SparseOperator((ip, cols.reshape(-1), data.reshape(-1)), shape=(len(rws), m))
    ref = SparseOperator((ip, cols.reshape(-1), data.reshape(-1)), shape=(len(rws), m), diagonal_windows=False,
                         local_x=False)
    xb = rng.normal(size=m)
    # (7 entries per row: below the 8 entries from which the diagonal form pays; not used)
    assert band.summary[3] == 0
    wide = np.arange(-20, 21)                              # 41 entries per row: diagonal windows, one irregular row
    cw = (rws[:, None] + wide[None, :]).astype(np.int32)
    dw = rng.normal(size=cw.shape)
    cw[9000, 5] = cw[9000, 4]                              # a duplicated column inside one row
    ipw = np.arange(len(rws) + 1, dtype=np.int64) * len(wide)
    on_w = SparseOperator((ipw, cw.reshape(-1), dw.reshape(-1)), shape=(len(rws), m))
    off_w = SparseOperator((ipw, cw.reshape(-1), dw.reshape(-1)), shape=(len(rws), m), diagonal_windows=False, local_x=False)
    nwin_w = on_w.nnz // 1792 + 1
    assert 0 < on_w.summary[3] + on_w.summary[2] <= nwin_w
    assert H.same_bits(on_w.dot(xb).cpu().numpy(), off_w.dot(xb).cpu().numpy())
    assert H.same_bits(band.dot(xb).cpu().numpy(), ref.dot(xb).cpu().numpy())
    want = H.spmv_canonical(ip[8990:9011] - ip[8990], cols[8990:9010].reshape(-1), data[8990:9010].reshape(-1), xb)
    assert H.same_bits(band.dot(xb).cpu().numpy()[8990:9010], want)


@pytest.mark.parametrize("L", [2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 20, 31, 32, 33, 40, 63, 64, 65])
def test_short_row_diagonal_windows_take_the_threads_per_row_path_and_change_no_bit(L):
    """Diagonal windows whose rows hold at most 64 entries are multiplied by 1 / 2 / 4 THREADS per row (up to 16 / 32 /
    64 entries; ccqp_spmv.hip, spmv_short_rows): values transposed through LDS, coalesced x gathers, the canonical
    order as additions in registers plus at most two DPP steps.  Same bits as the generic paths and as the canonical
    order, with and without the epilogue, for full and ragged last windows; full windows of 2, 4, 6 entries per row do
    not fit the padded LDS layout and stay generic; 65 entries per row take round 2's diagonal path."""
    from ccqppy_amd.operators import SparseOperator
    rng = np.random.default_rng(100 + L)
    rows = 40000 + 7 * L
    offs = np.sort(rng.choice(np.arange(1, 400), size=L, replace=False)) * 5000      # no column shared inside a window
    offs[L // 3] += 1                                          # (not all offsets multiples of one stride)
    m = rows + int(offs.max()) + 1
    cols = (np.arange(rows)[:, None] + offs[None, :]).astype(np.int32)
    data = rng.normal(size=cols.shape)
    data[17, L // 2] = -0.0                                    # a product of -0.0: the lane sum starts from +0.0
    ip = np.arange(rows + 1, dtype=np.int64) * L
    x = rng.normal(size=m)
    x[cols[23, 0]] = 0.0
    on = SparseOperator((ip, cols.reshape(-1), data.reshape(-1)), shape=(rows, m))
    off = SparseOperator((ip, cols.reshape(-1), data.reshape(-1)), shape=(rows, m), diagonal_windows=False, local_x=False)
    nwin = on.nnz // 1792 + 1
    assert on.summary[2] == 0 and off.summary[3] == 0
    if L in (2, 4, 6):
        assert on.summary[3] <= 1                              # (only a ragged last window can fit)
    else:
        assert on.summary[3] >= nwin - 1
    y_on, y_off = on.dot(x).cpu().numpy(), off.dot(x).cpu().numpy()
    assert H.same_bits(y_on, y_off)
    for lo in (0, 10, rows // 2, rows - 40):
        want = H.spmv_canonical(ip[lo:lo + 41] - ip[lo], cols[lo:lo + 40].reshape(-1), data[lo:lo + 40].reshape(-1), x)
        assert H.same_bits(y_on[lo:lo + 40], want)
    b = rng.normal(size=rows)
    eng = on.engine
    yd, gd = eng.empty(rows), eng.empty(rows)
    on.apply(eng, eng.to_device(x), yd, eng.to_device(b), gd)
    assert H.same_bits(yd.cpu().numpy(), y_on) and H.same_bits(gd.cpu().numpy(), y_on + b)
    # row blocks re-plan for themselves (other window boundaries, a ragged last window)
    assert H.same_bits(on.row_block(1234, 7001).dot(x).cpu().numpy(), y_on[1234:8235])
    assert H.same_bits(on.row_block(rows - 50, 50).dot(x).cpu().numpy(), y_on[rows - 50:])


def test_duplicate_and_unsorted_columns_inside_rows():
    """CSR does not forbid repeated or unsorted columns within a row: every stored entry contributes, in storage
    order, on every path (entry order, transposed, local-x)."""
    from ccqppy_amd.operators import SparseOperator
    rng = np.random.default_rng(11)
    for rows, cols, per_row in ((3000, 40, 7), (500, 5000, 40), (64, 64, 64)):
        indptr = np.arange(rows + 1, dtype=np.int64) * per_row
        indices = rng.integers(0, cols, size=rows * per_row).astype(np.int32)     # repeats, any order
        data = rng.normal(size=rows * per_row)
        x = rng.normal(size=cols)
        want = H.spmv_canonical(indptr, indices, data, x)
        for local_x in (True, False):
            op = SparseOperator((indptr, indices, data), shape=(rows, cols), local_x=local_x)
            assert H.same_bits(op.dot(x).cpu().numpy(), want), (rows, cols, local_x)
        dense = np.zeros((rows, cols))
        np.add.at(dense, (np.repeat(np.arange(rows), per_row), indices), data)
        np.testing.assert_allclose(want, dense.dot(x), rtol=1e-11, atol=1e-12)


def test_empty_matrix_and_empty_rows():
    from ccqppy_amd.operators import SparseOperator
    x = np.arange(1.0, 6.0)
    op = SparseOperator((np.zeros(4, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0)), shape=(3, 5))
    assert op.dot(x).cpu().numpy().tolist() == [0.0, 0.0, 0.0]
    # 5000 rows, only three of them non-empty (many more rows than entries in every window)
    indptr = np.zeros(5001, dtype=np.int64)
    indptr[11:] = 2
    indptr[2500:] = 3
    indptr[5000:] = 5
    op = SparseOperator((indptr, np.array([0, 4, 2, 1, 3], dtype=np.int32), np.array([1.0, 2.0, 3.0, 4.0, 5.0])),
                        shape=(5000, 5))
    y = op.dot(x).cpu().numpy()
    assert y[10] == 1.0 * 1 + 2.0 * 5 and y[2499] == 9.0 and y[4999] == 4.0 * 2 + 5.0 * 4
    assert np.count_nonzero(y) == 3


def test_input_formats_and_row_blocks():
    from ccqppy_amd.operators import SparseOperator, as_operator
    A = gen_ref.irregular_spd(700, 11)
    x = np.random.default_rng(5).normal(size=700)
    base = as_operator(A)                                   # scipy CSR
    assert isinstance(base, SparseOperator) and base.shape == (700, 700) and base.nnz == A.nnz
    y0 = base.dot(x).cpu().numpy()
    np.testing.assert_allclose(y0, A.dot(x), rtol=1e-13, atol=1e-13)
    for other in (A.tocoo(), A.tocsc(), sp.csr_array(A)):   # converted through .tocsr(): same sorted CSR
        assert H.same_bits(as_operator(other).dot(x).cpu().numpy(), y0)
    t = torch.sparse_csr_tensor(torch.from_numpy(A.indptr.astype(np.int64)), torch.from_numpy(A.indices.astype(np.int64)),
                                torch.from_numpy(A.data), size=(700, 700))
    assert H.same_bits(as_operator(t).dot(x).cpu().numpy(), y0)
    assert H.same_bits(as_operator(t.cuda()).dot(x).cpu().numpy(), y0)
    assert H.same_bits(as_operator(t.to_sparse_coo()).dot(x).cpu().numpy(), y0)
    # any row block gives the bits of those rows of the full product (what row-sharding relies on)
    for row0, rows in ((0, 700), (0, 1), (123, 400), (699, 1), (350, 0)):
        blk = base.row_block(row0, rows)
        assert blk.shape == (rows, 700)
        assert H.same_bits(blk.dot(x).cpu().numpy(), y0[row0:row0 + rows])
    with pytest.raises(ValueError):
        SparseOperator((A.indptr, A.indices, A.data))       # triple without shape


def test_invalid_csr_is_refused_before_any_matvec():
    from ccqppy_amd.operators import SparseOperator
    A = gen_ref.irregular_spd(100, 2)
    ip, ix, dv = A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data
    bad_col = ix.copy()
    bad_col[17] = 100
    neg_col = ix.copy()
    neg_col[3] = -1
    not_mono = ip.copy()
    not_mono[10], not_mono[11] = not_mono[11], not_mono[10] - 1
    short = ip.copy()
    short[-1] -= 1
    shifted = ip + 1
    for rowptr, col in ((ip, bad_col), (ip, neg_col), (not_mono, ix), (short, ix), (shifted, ix)):
        with pytest.raises(ValueError, match="CSR invariants"):
            SparseOperator((rowptr, col, dv), shape=(100, 100))
    with pytest.raises(ValueError, match="inconsistent"):
        SparseOperator((ip[:-1], ix, dv), shape=(100, 100))
    SparseOperator((ip, ix, dv), shape=(100, 100))          # the untouched arrays pass


def test_device_generator_matches_oracle_generator_bits():
    from ccqppy_amd import problems
    for n, k, seed, row0, rows in ((256, 8, 1234, 0, None), (4096, 100, 1234, 1000, 2000)):
        K = gen_ref.sparse_spd(n, k, seed, row0, rows)
        rowptr, col, val = problems.sparse_spd_csr(n, k, seed, row0, rows)
        assert val.is_cuda
        assert np.array_equal(K.indptr, rowptr.cpu().numpy())
        assert np.array_equal(K.indices, col.cpu().numpy())
        assert H.same_bits(K.data, val.cpu().numpy())
    op, b, xs = problems.random_sparse_problem(4096, 32, 1234)
    A, b_ref, xs_ref = gen_ref.sparse_problem(4096, 32, 1234)
    assert H.same_bits(xs.cpu().numpy(), xs_ref)
    np.testing.assert_allclose(b.cpu().numpy(), b_ref, rtol=1e-13, atol=1e-14)


_OPS = {}


def _operator(pid):
    from ccqppy_amd import problems
    from ccqppy_amd.operators import SparseOperator
    if pid not in _OPS:
        if pid.startswith("sgen:"):
            _, n, k, seed = pid.split(":")
            n = int(n)
            _OPS[pid] = SparseOperator(problems.sparse_spd_csr(n, int(k), int(seed)), shape=(n, n))
        elif pid.startswith("sstencil:"):
            _, n, seed = pid.split(":")
            n = int(n)
            _OPS[pid] = SparseOperator(problems.sparse_spd_csr(n, 26, int(seed), pattern="stencil27"), shape=(n, n))
            assert _OPS[pid].summary[2] > 0                      # local-x windows in play
        else:
            _OPS[pid] = H.sparse_fixture_matrix(pid, ARR)        # scipy input, converted by every solve()
    return _OPS[pid]


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_solver_golden_sparse(case):
    from ccqppy_amd import solvers
    A = _operator(case["problem"])
    b = ARR[case["problem"] + ".b"]
    n = b.shape[0]
    P = None if case["op"] is None else C.build_op(case["op"], C.product_factory(), dict(C.SUITE_ARRAYS))
    max_mv = np.inf if case["max_mv"] is None else case["max_mv"]
    np.random.seed(case["rng_seed"])
    s = getattr(solvers, "CCQPSolver" + case["solver"])(case["tol"], max_mv, **case["kwargs"])
    s.trace = []
    with contextlib.redirect_stdout(io.StringIO()):
        s.solve(A, b, x0=C.case_x0(case, n), convex_proj_op=P)
    want = ARR[case["id"] + ".x"]
    lo, hi = case["mv_band"]
    if lo == hi:
        assert s.solution_num_matrix_vector_multiplications == case["mv"]
        assert s.executed_matrix_vector_multiplications == case["executed_mv"]
        rtol = RTOL
        H.check_trace(case, s, ARR, n)               # per-iteration steering scalars (first 32 iterations)
    else:       # the reference's own count moves under 1-ulp perturbations of b (band in the fixture)
        assert lo - (hi - lo) <= s.solution_num_matrix_vector_multiplications <= hi + (hi - lo)
        rtol = max(RTOL, 8 * case["x_spread"])
    assert bool(s.solution_converged) == case["converged"]
    assert np.linalg.norm(s.solution - want) <= rtol * np.linalg.norm(want) + 1e-14
    if case["converged"] and case["solver"] != "SPG":
        assert float(s.solution_residual) < case["tol"]


def test_sparse_in_both_batch_schedulers_matches_plain_solves():
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd.batch import solve_batch, solve_batch_fused
    n = 2048
    probs = []
    for p in range(6):
        op, b, _ = problems.random_sparse_problem(n, 16, 100 + p)
        probs.append((op, b, None, ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))))
    for make in (lambda: solvers.CCQPSolverMPRGP(1e-6, 300), lambda: solvers.CCQPSolverAPGD(1e-8, 5000)):
        plain = []
        for op, b, x0, P in probs:
            with contextlib.redirect_stdout(io.StringIO()):
                plain.append(make().solve(op, b, convex_proj_op=P))
        for sched in (lambda: solve_batch(make, probs, lanes=3), lambda: solve_batch_fused(make, probs, lanes=4, groups=2)):
            got = sched()
            for a, c in zip(plain, got):
                assert a.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
                assert H.same_bits(a.solution.cpu().numpy(), c.solution.cpu().numpy())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _solve_set(op, b, n):
    from ccqppy_amd import solvers, solution_spaces as ss
    out = {}
    box = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    for name, cls, args in (("APGD", solvers.CCQPSolverAPGD, (1e-8, 5000)), ("BBPGD", solvers.CCQPSolverBBPGD, (1e-8, 5000)),
                            ("SPG", solvers.CCQPSolverSPG, (1e-8, 5000)), ("MPRGP", solvers.CCQPSolverMPRGP, (1e-6, 300))):
        np.random.seed(1234)
        with contextlib.redirect_stdout(io.StringIO()):
            r = cls(*args).solve(op, b, convex_proj_op=box)
        out[name] = (r.solution.cpu().numpy().copy(), int(r.solution_num_matrix_vector_multiplications))
    return out


def _worker(rank, world, port, n, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from ccqppy_amd import problems
        op, b, _ = problems.random_sparse_problem(n, k, 1234, device=0, group=dist.group.WORLD)
        assert op.sparse and op.rows < n
        q.put((rank, _solve_set(op, b, n)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,k", [(4096, 32), (1001, 8)])
def test_row_sharded_sparse_equals_single_gpu_bitwise(n, k):
    import torch.multiprocessing as mp
    from ccqppy_amd import problems
    op, b, _ = problems.random_sparse_problem(n, k, 1234, device=0)
    single = _solve_set(op, b, n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, k, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, res in results:
        for name, (x, mv) in res.items():
            assert mv == single[name][1], (name, rank)
            assert H.same_bits(x, single[name][0]), (name, rank)
