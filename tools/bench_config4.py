#!/usr/bin/env python3
"""BASELINE config 4: n = 262144 dense SPD, row-block sharded, DisjointUnion(Box, Cone), SPG with a 64-MV budget.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        tools/bench_config4.py [--size 262144] [--bw1-gbs 6900]

One rank per GPU (RCCL); rank 0 prints ONE JSON line.  The problem cannot reside on one MI355X, so the
1-GPU comparator is the model SURVEY.md 8(d) prescribes, stated in the output:
    T1_model = MV_executed * 8 n (n + 2) / BW1,     speedup_N = T1_model / T_N
with BW1 = the 1-GPU achieved matvec rate at n = 131072 (bench.py roofline.achieved; default 6900 GB/s, the
round-1 measurement).  Without torchrun (1 process) the same solve runs on ONE GPU through the matrix-free
RegeneratedRowBlocksOperator (every matvec regenerates the 8 row blocks) -- a functional check of the
workload, not a timing to compare.  CCQP_DIST_BACKEND=gloo rehearses the N-rank path on a 1-GPU box."""
import argparse
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", dest="n", type=int, default=262144)
    ap.add_argument("--max-mv", type=int, default=64)
    ap.add_argument("--bw1-gbs", type=float, default=6900.0)
    ap.add_argument("--exchange", dest="mode", default="allgather", choices=["allgather", "allreduce", "direct"])
    ap.add_argument("--repeats", type=int, default=2)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("CCQP_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    from ccqppy_amd.operators import RegeneratedRowBlocksOperator
    n = args.n
    eng = get_engine(local_rank)
    if world > 1:
        op, b, _ = problems.random_problem(n, 1234, device=local_rank, group=group, mode=args.mode)
    else:
        op = RegeneratedRowBlocksOperator(n, 1234, 8, eng)
        xs = problems.xstar(n, 1234, eng.device)
        Ax = eng.empty(n)
        op.matvec(xs, Ax)
        b = eng.empty(n)
        eng.scale(-1.0, Ax, b)
    half = n // 2
    P = ss.DisjointProjOp(ss.BoxProjOp(half, -0.5, 0.5), ss.ConeProjOp(n - half))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    best = None
    for rep in range(args.repeats + 1):          # first pass = warm-up
        np.random.seed(1234)
        sync()
        mv0 = eng.mv_executed
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            r = solvers.CCQPSolverSPG(1e-8, args.max_mv).solve(op, b, convex_proj_op=P)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        if rep > 0 and (best is None or dt < best[0]):
            best = (dt, eng.mv_executed - mv0, r)
    dt, mv, r = best
    if rank == 0:
        mv_bytes = 8.0 * n * (n + 2)
        t1 = mv * mv_bytes / (args.bw1_gbs * 1e9)
        line = {"config": "config4: n=%d dense SPD row-block x%d, DisjointUnion(Box,Cone), SPG max_mv=%d" % (n, world, args.max_mv),
                "n_gpus": world, "wall_s": dt, "executed_mv": int(mv),
                "reported_mv": int(r.solution_num_matrix_vector_multiplications),
                "converged": bool(r.solution_converged), "residual": float(r.solution_residual),
                "matvecs_per_s": mv / dt, "aggregate_GBps": mv * mv_bytes / dt / 1e9,
                "frac_of_aggregate_hbm_peak": mv * mv_bytes / dt / 1e9 / (8000.0 * max(world, 1)),
                "T1_model_s": t1,
                "T1_model_formula": "MV_executed * 8*n*(n+2) / BW1, BW1 = %.0f GB/s (1-GPU achieved rate at n=131072)" % args.bw1_gbs,
                "speedup_vs_T1_model": (t1 / dt) if world > 1 else None,
                "operator": "RowShardedOperator (%s %s)" % ("RCCL" if backend == "nccl" else backend, args.mode) if world > 1 else
                            "RegeneratedRowBlocksOperator (matrix-free, 1 GPU: functional check only)"}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
