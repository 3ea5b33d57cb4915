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
    # ("APGD paired": the two matvecs of an iteration from one two-vector matvec -- on a row-sharded operator one pass
    # over the local row block and two exchanges; single GPU: no pair form asked for, so the comparison is with the
    # plain solve)
    paired = (lambda *a: solvers.CCQPSolverAPGD(*a, pair_matvecs=hasattr(op, "exchange")))
    for name, cls, args, P in (("APGD", solvers.CCQPSolverAPGD, (1e-8, 5000), box),
                               ("APGD paired", paired, (1e-8, 5000), box),
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


def _solve_sharing(op, b, n):
    """The solvers whose products share passes -- and, on a row-sharded operator, ONE exchange per pass: APGD and its
    anti-relaxation variant with paired matvecs, MPRGP with its look-ahead (forced on; single GPU: the plain loops)."""
    from ccqppy_amd import solvers, solution_spaces as ss
    sharded = hasattr(op, "exchange")
    box = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    out = {}
    for name, make in (("APGD", lambda: solvers.CCQPSolverAPGD(1e-8, 5000, pair_matvecs=sharded)),
                       ("AR", lambda: solvers.CCQPSolverAPGDAntiRelaxation(1e-8, 5000, pair_matvecs=sharded)),
                       ("APGD budget 9", lambda: solvers.CCQPSolverAPGD(1e-12, 9, pair_matvecs=sharded)),
                       ("MPRGP", lambda: solvers.CCQPSolverMPRGP(1e-6, 300, look_ahead=sharded)),
                       ("MPRGP budget 7", lambda: solvers.CCQPSolverMPRGP(1e-12, 7, look_ahead=sharded))):
        eng = op.engine
        calls0 = op.exchange.calls if sharded else 0
        launched0 = eng.products_launched
        with contextlib.redirect_stdout(io.StringIO()):
            r = make().solve(op, b, convex_proj_op=box)
        out[name] = (r.solution.cpu().numpy().copy(), int(r.solution_num_matrix_vector_multiplications),
                     int(r.executed_matrix_vector_multiplications), float(r.solution_residual),
                     (op.exchange.calls - calls0) if sharded else None, eng.products_launched - launched0)
    return out


def _sharing_worker(rank, world, port, n, mode, variant, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from ccqppy_amd import problems
        from ccqppy_amd._engine import get_engine
        get_engine(0).gemv_variant = variant
        op, b, _ = problems.random_problem(n, 1234, device=0, group=dist.group.WORLD, mode=mode)
        forms = (op.pair_in_one_pass(), op.triple_in_one_pass())
        q.put((rank, forms, _solve_sharing(op, b, n)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,mode,variant", [(1024, "allgather", 0x51001800), (1001, "allgather", 0x51001800),
                                            (1024, "allreduce", 0x51001800), (1026, "allgather", 0)])
def test_shared_passes_share_one_exchange_and_change_no_bit(n, mode, variant):
    """Row-sharded APGD / anti-relaxation APGD with paired matvecs and MPRGP with its look-ahead, 2 ranks: per pass ONE
    local multi-vector launch (variant 0x51001800 forces the LDS-staged forms gemv_lds2_kernel / gemv_lds3_kernel onto
    these small blocks; n = 1001 and variant 0 fall back to single local launches) + ONE exchange + one unpack launch.
    x, residual, reported and executed counts are bitwise those of the plain single-GPU solves; the number of exchanges
    is the number of PASSES (fewer than the matvecs executed)."""
    import torch.multiprocessing as mp
    from ccqppy_amd import problems
    op, b, _ = problems.random_problem(n, 1234, device=0)
    single = _solve_sharing(op, b, n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharing_worker, args=(r, 2, port, n, mode, variant, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, forms, res in results:
        assert forms == ((True, True) if (variant and n % 2 == 0) else (False, False)), (rank, forms)
        for name, (x, mv, execd, resid, exchanges, launched) in res.items():
            x1, mv1, ex1, res1, _, _ = single[name]
            assert (mv, execd) == (mv1, ex1), (name, rank)
            if mode == "allgather":
                assert np.array_equal(x.view(np.uint64), x1.view(np.uint64)), (name, rank)
                assert resid == res1
            else:
                assert np.array_equal(x, x1), (name, rank)
            # one exchange per pass: the products computed come in pairs / triples, each pass exchanged once
            assert exchanges < launched and exchanges <= execd, (name, rank, exchanges, launched, execd)
            if name == "APGD":
                assert 2 * exchanges == launched, (name, rank, exchanges, launched)


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


def _rccl_direct_worker(port, q):
    """RCCL through the C ABI (ccqp_comm_*, ccqp_shard_gemv_allgather_f64) with the one rank a 1-GPU box allows: the
    bootstrap (unique id -> communicator on the context's device), the in-place all-gather and the padded all-reduce
    on our own stream behind our own kernels, the sharded matvec entry, and VectorExchange(mode="direct")."""
    import ctypes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    out = {}
    try:
        from ccqppy_amd import _lib, problems
        from ccqppy_amd._engine import get_engine
        from ccqppy_amd.operators import VectorExchange
        eng = get_engine(0)
        lib = eng.lib
        out["version"] = int(lib.ccqp_comm_rccl_version())
        idbuf = (ctypes.c_ubyte * 128)()
        _lib.check(lib.ccqp_comm_unique_id(idbuf))
        comm = ctypes.c_void_p()
        _lib.check(lib.ccqp_comm_create(eng.ctx, 1, 0, idbuf, ctypes.byref(comm)))
        nr, rk = ctypes.c_int(), ctypes.c_int()
        _lib.check(lib.ccqp_comm_info(comm, ctypes.byref(nr), ctypes.byref(rk)))
        out["info"] = (nr.value, rk.value)
        n = 4096
        A = problems.spd_rows(n, 5)
        x = problems.xstar(n, 5)
        want = eng.empty(n)
        eng.gemv(A, x, want)
        for mode in (0, 1):
            y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
            _lib.check(lib.ccqp_shard_gemv_allgather_f64(eng.ctx, comm, A.data_ptr(), A.stride(0), x.data_ptr(),
                                                         y.data_ptr(), n, n, n, 0, mode, eng.stream))
            torch.cuda.synchronize()
            out["shard_mode%d" % mode] = bool(torch.equal(y.view(torch.int64), want.view(torch.int64)))
        # k products, one exchange (ccqp_shard_gemv2_allgather_f64 / ccqp_shard_gemvk_allgather_f64): local pass into
        # `work`, in-place all-gather, unpack with the + b epilogue -- each product bitwise its own matvec
        xs = [x, problems.xstar(n, 6), problems.xstar(n, 7)]
        bvec = problems.xstar(n, 8)
        wants = []
        for xj in xs:
            yj, gj = eng.empty(n), eng.empty(n)
            eng.gemv(A, xj, yj, bvec, gj)
            wants.append((yj, gj))
        for variant in (0, 0x51001800):
            work = torch.full((3 * n,), float("nan"), dtype=torch.float64, device="cuda")
            ys = [torch.full((n,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(3)]
            gs = [torch.full((n,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(3)]
            _lib.check(lib.ccqp_shard_gemv2_allgather_f64(eng.ctx, comm, A.data_ptr(), A.stride(0), xs[0].data_ptr(),
                                                          xs[1].data_ptr(), bvec.data_ptr(), ys[0].data_ptr(), gs[0].data_ptr(),
                                                          ys[1].data_ptr(), None, n, n, n, variant, work.data_ptr(), eng.stream))
            torch.cuda.synchronize()
            ok = all(torch.equal(ys[j].view(torch.int64), wants[j][0].view(torch.int64)) for j in range(2))
            ok = ok and torch.equal(gs[0].view(torch.int64), wants[0][1].view(torch.int64)) and bool(torch.isnan(gs[1]).all())
            arr = ctypes.c_void_p * 3
            for t in ys + gs:
                t.fill_(float("nan"))
            _lib.check(lib.ccqp_shard_gemvk_allgather_f64(eng.ctx, comm, A.data_ptr(), A.stride(0), 3,
                                                          arr(*[t.data_ptr() for t in xs]), bvec.data_ptr(),
                                                          arr(*[t.data_ptr() for t in ys]),
                                                          arr(None, gs[1].data_ptr(), gs[2].data_ptr()), n, n, n, variant,
                                                          work.data_ptr(), eng.stream))
            torch.cuda.synchronize()
            ok = ok and all(torch.equal(ys[j].view(torch.int64), wants[j][0].view(torch.int64)) for j in range(3))
            ok = ok and all(torch.equal(gs[j].view(torch.int64), wants[j][1].view(torch.int64)) for j in (1, 2))
            out["shard_many_%x" % variant] = bool(ok)
        # the collectives themselves, on our stream right behind one of our kernels (1 rank: identity / x1 sum)
        z = eng.empty(n)
        eng.scale(2.0, want, z)
        _lib.check(lib.ccqp_comm_allgather_f64(comm, z.data_ptr(), n, eng.stream))
        _lib.check(lib.ccqp_comm_allreduce_sum_f64(comm, z.data_ptr(), n, eng.stream))
        torch.cuda.synchronize()
        out["collectives"] = bool(torch.equal(z, 2.0 * want))
        rc = lib.ccqp_comm_create(eng.ctx, 2, 5, idbuf, ctypes.byref(ctypes.c_void_p()))
        out["bad_rank_rejected"] = rc != 0
        _lib.check(lib.ccqp_comm_destroy(comm))
        ex = VectorExchange(n, dist.group.WORLD, "direct")
        out["exchange_mode"] = (ex.direct, ex.mode)
        ex(z, eng)                                  # world == 1: nothing to exchange, must not touch RCCL
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_rccl_through_the_c_abi_single_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_direct_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=180)
    p.join(60)
    assert p.exitcode == 0
    assert out["version"] > 0 and out["info"] == (1, 0)
    assert out["shard_mode0"] and out["shard_mode1"] and out["collectives"] and out["bad_rank_rejected"]
    assert out["shard_many_0"] and out["shard_many_51001800"]
    assert out["exchange_mode"] == (True, "allgather")
