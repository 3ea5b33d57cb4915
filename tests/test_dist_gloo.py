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
