"""GPU, 2 ranks on ONE device: the row-sharded operator end to end (local HIP GEMV on the rank's
row block -> vector exchange -> replicated host loop).  NCCL/RCCL refuses two ranks on one GPU, so
the exchange runs over gloo with host staging here (`VectorExchange.stage_cpu`); everything else is
the production path.  Asserts the property the design promises: results are BITWISE identical to
the single-GPU solve (same x, same MV count), for even and ragged row splits and both modes."""
import contextlib
import io
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _solve_all(op, b, n):
    from ccqppy_amd import solvers, solution_spaces as ss
    out = {}
    box = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    mixed = ss.DisjointProjOp(ss.BoxProjOp(n // 2, -0.5, 0.5), ss.ConeProjOp(n - n // 2))
    for name, cls, args, P in (("APGD", solvers.CCQPSolverAPGD, (1e-8, 5000), box),
                               ("BBPGD", solvers.CCQPSolverBBPGD, (1e-8, 5000), box),
                               ("SPG", solvers.CCQPSolverSPG, (1e-8, 64), mixed),
                               ("MPRGP", solvers.CCQPSolverMPRGP, (1e-6, 300), box)):
        np.random.seed(1234)
        with contextlib.redirect_stdout(io.StringIO()):
            r = cls(*args).solve(op, b, convex_proj_op=P)
        out[name] = (r.solution.cpu().numpy().copy(), int(r.solution_num_matrix_vector_multiplications),
                     float(r.solution_residual))
    return out


def _worker(rank, world, port, n, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from ccqppy_amd import problems
        op, b, _ = problems.random_problem(n, 1234, device=0, group=dist.group.WORLD, mode=mode)
        assert op.rows < n
        q.put((rank, _solve_all(op, b, n)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,mode", [(1024, "allgather"), (1001, "allgather"), (1024, "allreduce")])
def test_row_sharded_equals_single_gpu_bitwise(n, mode):
    import torch.multiprocessing as mp
    from ccqppy_amd import problems
    op, b, _ = problems.random_problem(n, 1234, device=0)
    single = _solve_all(op, b, n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, res in results:
        for name, (x, mv, resid) in res.items():
            x1, mv1, res1 = single[name]
            assert mv == mv1, (name, rank)
            if mode == "allgather":
                assert np.array_equal(x.view(np.uint64), x1.view(np.uint64)), (name, rank)
                assert resid == res1
            else:   # all-reduce adds +0.0 to every entry: only the sign of an exact zero may differ
                assert np.array_equal(x, x1), (name, rank)


def _rccl_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from ccqppy_amd import problems, solvers, solution_spaces as ss
        from ccqppy_amd.operators import VectorExchange
        n = 4096
        op, b, _ = problems.random_problem(n, 1234, device=0, group=dist.group.WORLD, mode="allgather")
        ex = VectorExchange(n, dist.group.WORLD, "allgather")
        assert not ex.stage_cpu
        y = torch.arange(n, dtype=torch.float64, device="cuda") * 0.37
        want = y.clone()
        # the RCCL calls of the N>1 path, issued on torch's current stream right behind a HIP kernel of ours
        op.engine.bind()
        op.engine.scale(1.0, want, y)
        ex._gather_into(y, y[ex.row0:ex.row0 + ex.rows], op.engine)
        dist.all_reduce(y, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        same = bool(torch.equal(y, want))
        with contextlib.redirect_stdout(io.StringIO()):
            r = solvers.CCQPSolverAPGD(1e-8, 5000).solve(op, b, convex_proj_op=ss.BoxProjOp(n, -0.5, 0.5))
        q.put((same, bool(r.solution_converged), int(r.solution_num_matrix_vector_multiplications)))
    finally:
        dist.destroy_process_group()


def test_rccl_backend_single_rank_collectives():
    """RCCL itself (backend "nccl") on the box's one GPU: process-group creation on the device, the
    all_gather_into_tensor / all_reduce calls of the vector exchange ordered behind our own kernels on
    the same stream, and a solve through a group-bound operator."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    same, conv, mv = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0
    assert same and conv and mv > 0


def _batch_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from ccqppy_amd import problems, solvers, solution_spaces as ss
        from ccqppy_amd.batch import solve_batch_sharded
        n = 1024

        def make(p):
            op, b, _ = problems.random_problem(n, 300 + p)
            return op, b, None, ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
        res = solve_batch_sharded(lambda: solvers.CCQPSolverMPRGP(1e-6, 300), make, 7, group=dist.group.WORLD,
                                  lanes=4, gather=True)
        q.put((rank, {p: (r.solution, r.solution_num_matrix_vector_multiplications, r.solution_converged, r.name)
                      for p, r in res.items()}))
    finally:
        dist.destroy_process_group()


def test_batch_sharded_two_ranks_with_final_gather():
    """Batch-sharded mode over 2 ranks (problem p -> rank p % 2, no collective while solving) + the one
    end-of-job gather: every rank ends up with all 7 results, equal to plain single-process solves."""
    import torch.multiprocessing as mp
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    n = 1024
    want = {}
    for p in range(7):
        op, b, _ = problems.random_problem(n, 300 + p)
        with contextlib.redirect_stdout(io.StringIO()):
            r = solvers.CCQPSolverMPRGP(1e-6, 300).solve(op, b, convex_proj_op=ss.BoxProjOp(n, -0.5 * np.ones(n),
                                                                                           0.5 * np.ones(n)))
        want[p] = (r.solution.cpu().numpy(), int(r.solution_num_matrix_vector_multiplications))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, res in results:
        assert sorted(res) == list(range(7))
        for p, (x, mv, conv, name) in res.items():
            assert mv == want[p][1] and conv and name == "MPRGP"
            assert np.array_equal(x.view(np.uint64), want[p][0].view(np.uint64)), (rank, p)
