"""Synthetic random-SPD problems generated ON the GPU.

Counterpart of `BenchmarkRandomCCQP.generate_random_convex_quadratic_func`
(/root/reference/benchmarks/benchmark_random_ccqp.py:40-63), whose Wishart generator is O(n^3)
and unusable beyond n ~ 1e4.  Here A is a counter-hashed symmetric matrix with spectrum
~[0.845, 3.155]; any row range can be produced independently, so each rank of a row-sharded
run fills only its own block, in HBM, with no host involvement.
"""
import torch

from ._engine import get_engine
from .operators import DenseOperator, RowShardedOperator, row_partition


def spd_rows(n, seed, row0=0, rows=None, device=None):
    """Rows [row0, row0+rows) of the n x n matrix as a (rows, n) fp64 device tensor."""
    eng = get_engine(device)
    rows = n - row0 if rows is None else rows
    A = torch.empty((rows, n), dtype=torch.float64, device=eng.device)
    eng.gen_spd_rows(A, row0, n, seed)
    return A


def xstar(n, seed, device=None):
    """x* = 1 - 2u, the unconstrained minimiser used to build b (benchmark_random_ccqp.py:61)."""
    eng = get_engine(device)
    x = eng.empty(n)
    eng.gen_vec(x, seed)
    return x


def random_problem(n, seed, device=None, group=None, mode="allgather"):
    """(operator, b, xstar): b = -A.xstar computed with the engine's own matvec.

    With `group` (a torch.distributed process group with > 1 rank, e.g. `dist.group.WORLD`) the matrix is
    row-sharded: this rank generates and keeps only rows row_partition(n, world, rank).  Without it the
    problem is local to this process even if torch.distributed is initialised (batch-sharded mode)."""
    eng = get_engine(device)
    xs = xstar(n, seed, eng.device)
    world = 1
    if group is not None:                      # sharding is opt-in: batch-sharded ranks build LOCAL problems
        world = torch.distributed.get_world_size(group)
    if world > 1:
        rank = torch.distributed.get_rank(group)
        row0, rows, _ = row_partition(n, world, rank)
        op = RowShardedOperator(spd_rows(n, seed, row0, rows, eng.device), n, group, mode, eng)
    else:
        op = DenseOperator(spd_rows(n, seed, 0, n, eng.device), eng)
    Ax = eng.empty(n)
    op.matvec(xs, Ax)
    b = eng.empty(n)
    eng.scale(-1.0, Ax, b)
    return op, b, xs


# ------------------------------------------------------------------ sparse problems
_M64 = (1 << 64) - 1
_GOLDEN, _M1, _M2 = 0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB
_OFF_STREAM = 0xA0761D6478BD642F


def _mix_int(z):
    """SplitMix64 output step on a Python int (host side: keys and the few pattern offsets)."""
    z = (z + _GOLDEN) & _M64
    z = ((z ^ (z >> 30)) * _M1) & _M64
    z = ((z ^ (z >> 27)) * _M2) & _M64
    return z ^ (z >> 31)


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _mix_tensor(z):
    """The same step on an int64 tensor (wrapping arithmetic; logical shifts by masking)."""
    z = z + _signed(_GOLDEN)
    z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * _signed(_M1)
    z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * _signed(_M2)
    return z ^ ((z >> 31) & ((1 << 33) - 1))


def sparse_offsets(n, k, seed):
    """The k/2 distinct positive offsets d_t in [1, n/2 - 1] of the circulant sparsity pattern."""
    m, span = k // 2, n // 2 - 1
    if k % 2 or m < 1 or m > span:
        raise ValueError("sparse pattern needs an even k with 2 <= k <= 2*(n/2 - 1)")
    key = _mix_int((seed & _M64) ^ _OFF_STREAM)
    used, out = set(), []
    for t in range(m):
        d = 1 + _mix_int((key + t) & _M64) % span
        while d in used:
            d = d % span + 1
        used.add(d)
        out.append(d)
    return out


def stencil_offsets(n):
    """The 13 positive offsets of a periodic 27-point stencil on a side^3 grid, side = round(n^(1/3))."""
    side = int(round(n ** (1.0 / 3.0)))
    if side < 4 or 2 * (side * side + side + 1) >= n:
        raise ValueError("stencil27 pattern needs n ~ side^3 with side >= 4")
    offs = sorted({dz * side * side + dy * side + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)})
    return [d for d in offs if d > 0]


def sparse_spd_csr(n, k, seed, row0=0, rows=None, device=None, pattern="hashed"):
    """Rows [row0, row0+rows) of the sparse SPD test matrix as device CSR arrays (rowptr int64, col int32,
    val float64): the entries of the dense generator matrix on k hashed symmetric off-diagonals
    (columns (i +- d_t) mod n), rescaled to (2u - 1)/k, diagonal 2 => spectrum inside [1, 3] (Gershgorin).
    Columns are sorted within a row; every row holds k + 1 entries.  pattern="stencil27" (k = 26) takes the
    offsets of a periodic 27-point stencil on a side^3 grid instead of hashed ones: neighbouring rows then share
    columns, the case the matvec's local-x windows are made for."""
    rows = n - row0 if rows is None else rows
    dev = torch.device("cpu") if device == "cpu" else get_engine(device).device
    if pattern == "stencil27":
        if k != 26:
            raise ValueError("stencil27 has k = 26 off-diagonals")
        offs = stencil_offsets(n)
    else:
        offs = sparse_offsets(n, k, seed)
    delta = torch.tensor(sorted([-d for d in offs] + [0] + offs), dtype=torch.int64, device=dev)
    i = torch.arange(row0, row0 + rows, dtype=torch.int64, device=dev)[:, None]
    cols = torch.sort(torch.remainder(i + delta[None, :], n), dim=1).values
    kk = torch.minimum(i, cols) * n + torch.maximum(i, cols)
    h = _mix_tensor(kk + _signed(_mix_int(seed & _M64)))
    u = ((h >> 11) & ((1 << 53) - 1)).to(torch.float64) * (2.0 ** -53)
    # divide by a TENSOR: torch's GPU division by a Python scalar multiplies by the rounded reciprocal
    vals = (2.0 * u - 1.0) / torch.full((1, 1), float(k), dtype=torch.float64, device=dev)
    vals[cols == i] = 2.0
    rowptr = torch.arange(rows + 1, dtype=torch.int64, device=dev) * delta.numel()
    return rowptr, cols.reshape(-1).to(torch.int32), vals.reshape(-1)


def random_sparse_problem(n, k, seed, device=None, group=None, mode="allgather", pattern="hashed"):
    """(operator, b, xstar) for the sparse SPD test matrix with k off-diagonal entries per row;
    b = -A.xstar by the engine's own sparse matvec.  With a process group of > 1 rank the matrix is
    row-sharded (each rank builds only its own rows)."""
    from .operators import SparseOperator
    eng = get_engine(device)
    xs = xstar(n, seed, eng.device)
    world = 1
    if group is not None:                      # sharding is opt-in: batch-sharded ranks build LOCAL problems
        world = torch.distributed.get_world_size(group)
    if world > 1:
        rank = torch.distributed.get_rank(group)
        row0, rows, _ = row_partition(n, world, rank)
        local = SparseOperator(sparse_spd_csr(n, k, seed, row0, rows, eng.device, pattern), shape=(rows, n), engine=eng)
        op = RowShardedOperator(local, n, group, mode, eng)
    else:
        op = SparseOperator(sparse_spd_csr(n, k, seed, 0, n, eng.device, pattern), shape=(n, n), engine=eng)
    Ax = eng.empty(n)
    op.matvec(xs, Ax)
    b = eng.empty(n)
    eng.scale(-1.0, Ax, b)
    return op, b, xs
