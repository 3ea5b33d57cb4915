"""CPU, world_size 2 (gloo): the N>1 path of the row-sharded operator -- partition + vector
exchange -- with the local row-block matvec injected (numpy) in place of the HIP kernel.
Checks that every rank ends with the full, bitwise identical y for even and ragged n, in both
exchange modes (all-gather, zero-padded all-reduce)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ccqppy_amd.operators import VectorExchange, row_partition
        rng = np.random.default_rng(7)
        A = rng.normal(size=(n, n))
        x = rng.normal(size=n)
        # row sums in a fixed order that does not depend on the partition
        full = np.array([float(np.sum(A[i] * x)) for i in range(n)])
        ex = VectorExchange(n, None, mode)
        r0, rows, _ = row_partition(n, world, rank)
        assert (ex.row0, ex.rows) == (r0, rows)
        y = torch.full((n,), float("nan"), dtype=torch.float64)
        for i in range(r0, r0 + rows):
            y[i] = float(np.sum(A[i] * x))
        ex(y)
        got = y.numpy()
        ok = np.array_equal(got.view(np.uint64), full.view(np.uint64)) if mode == "allgather" \
            else np.array_equal(got, full)
        # a second exchange on the same object must work too (buffers are reused)
        y2 = torch.zeros(n, dtype=torch.float64)
        y2[r0:r0 + rows] = torch.from_numpy(full[r0:r0 + rows])
        ex(y2)
        ok = ok and np.array_equal(y2.numpy(), full)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,mode", [(64, "allgather"), (64, "allreduce"), (37, "allgather"),
                                    (37, "allreduce"), (1, "allgather")])
def test_vector_exchange_two_ranks(n, mode):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(0, True), (1, True)]


def _many_worker(rank, world, port, n, mode, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ccqppy_amd.operators import VectorExchange, row_partition
        rng = np.random.default_rng(11)
        full = [rng.normal(size=n) for _ in range(k)]          # the k assembled products, as one GPU would have them
        b = rng.normal(size=n)
        ex = VectorExchange(n, None, mode)
        r0, rows, per = row_partition(n, world, rank)
        ok = True
        for turn in range(2):                                  # (the buffers are reused by the second call)
            ys = [torch.full((n,), float("nan"), dtype=torch.float64) for _ in range(k)]
            gs = [torch.full((n,), float("nan"), dtype=torch.float64) if j != 1 else None for j in range(k)]
            slabs = ex.slabs(k, ys[0])
            assert len(slabs) == k and all(s.numel() == rows for s in slabs)
            for j in range(k):                                 # the local k-vector matvec writes the slabs in place
                slabs[j].copy_(torch.from_numpy(full[j][r0:r0 + rows] + turn))
            calls = ex.calls
            ex.exchange_many(ys, torch.from_numpy(b), gs)
            assert ex.calls == calls + 1                       # ONE exchange for the k products
            for j in range(k):
                want = full[j] + turn
                eq = np.array_equal(ys[j].numpy().view(np.uint64), want.view(np.uint64)) if mode == "allgather" \
                    else np.array_equal(ys[j].numpy(), want)
                ok = ok and eq
                if gs[j] is not None:
                    ok = ok and np.array_equal(gs[j].numpy(), want + b)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,mode,k", [(64, "allgather", 2), (37, "allgather", 3), (64, "allreduce", 2), (37, "allreduce", 3),
                                      (2, "allgather", 2), (1, "allgather", 3)])
def test_k_products_one_exchange_two_ranks(n, mode, k):
    """VectorExchange.slabs / exchange_many (APGD's pair, MPRGP's look-ahead triple on a row-sharded matrix): the k
    local slabs travel in ONE collective and are scattered into the k replicated vectors, g_j = y_j + b formed on the
    way -- even and ragged splits (a rank may hold no row at all), both collectives."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_many_worker, args=(r, world, port, n, mode, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(0, True), (1, True)]
