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

    With `group` (a torch.distributed process group with > 1 rank) the matrix is row-sharded:
    this rank generates and keeps only rows row_partition(n, world, rank)."""
    eng = get_engine(device)
    xs = xstar(n, seed, eng.device)
    world = 1
    if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
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
