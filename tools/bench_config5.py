#!/usr/bin/env python3
"""BASELINE config 5: batch = 1024 independent n = 8192 box-QPs, MPRGP(tol 1e-6, max 300 MVs),
batch-sharded (problem p -> rank p % world, NO collective on the data path).

    python tools/bench_config5.py                    # one GPU: its 1/8 share (128 problems, 64 GiB)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        tools/bench_config5.py --problems 1024        # the whole config

Prints one JSON line (rank 0): problems/s over the whole job, matvecs/s, solve-level HBM fraction."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools._opts import add_option_args, apply_options  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=None, help="total problems (default 128 per rank)")
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--lanes", type=int, default=32)
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--scheduler", default="pipelined", choices=["fused", "lanes", "sequential", "pipelined"])
    add_option_args(ap)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("CCQP_DIST_BACKEND", "nccl")   # gloo: rehearsal of N ranks on a 1-GPU box
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    options = apply_options(args, local)           # (per-context tuning options: the A/B switches, tools/README.md)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group(backend)
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd.batch import solve_batch, solve_batch_fused, solve_batch_pipelined
    from ccqppy_amd.operators import batch_partition
    n = args.size
    total = args.problems if args.problems is not None else 128 * world
    mine = batch_partition(total, world, rank)
    lb, ub = -0.5 * np.ones(n), 0.5 * np.ones(n)
    probs = []
    for p in mine:
        op, b, _ = problems.random_problem(n, 1234 + p, device=local)     # problem p uses seed 1234 + p
        probs.append((op, b, None, ss.BoxProjOp(n, lb, ub)))
    make = lambda: solvers.CCQPSolverMPRGP(1e-6, 300)       # noqa: E731
    best = None
    for rep in range(args.reps + 1):
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t0 = time.perf_counter()
        if args.scheduler == "pipelined":
            res = solve_batch_pipelined(make, probs, device=local)
        elif args.scheduler == "fused":
            res = solve_batch_fused(make, probs, lanes=args.lanes, groups=args.groups, device=local)
        elif args.scheduler == "lanes":
            res = solve_batch(make, probs, lanes=4, device=local)
        else:
            res = solve_batch(make, probs, lanes=1, device=local)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        if rep and (best is None or dt < best):
            best = dt
    mv = sum(r.executed_matrix_vector_multiplications for r in res)
    conv = sum(bool(r.solution_converged) for r in res)
    if world > 1:
        t = torch.tensor([mv, conv], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t)
        mv, conv = int(t[0].item()), int(t[1].item())
    if rank == 0:
        print(json.dumps({
            "options": options,
            "config": "BASELINE config 5: %d x (n=%d MPRGP+Box, tol 1e-6), batch-sharded over %d GPU(s), scheduler=%s"
                      % (total, n, world, args.scheduler),
            "n_gpus": world, "problems": total, "wall_s": best, "problems_per_s": total / best,
            "executed_mv": mv, "mv_per_s": mv / best, "converged": conv,
            "solve_level_hbm_frac_per_gpu": mv * 8.0 * (n * n + 2 * n) / best / 8e12 / world,
            "reported_mv_first8": [int(r.solution_num_matrix_vector_multiplications) for r in res[:8]],
        }), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
