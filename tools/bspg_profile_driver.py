#!/usr/bin/env python3
"""Minimal program to put behind `rocprofv3 --kernel-trace --stats --`: B Box problems of size n solved by lock-step
SPG-QP only (batch.solve_batch_pipelined -> ccqp_bspg_run_f64), every solver with its own RandomState; prints
problems/s, the matrix bytes/s, and the wall-clock split of the run (staging / driver / result objects).
    python tools/bspg_profile_driver.py [n] [B]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from ccqppy_amd import lockstep, problems, solvers, solution_spaces as ss
    from ccqppy_amd.batch import solve_batch_pipelined
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    P = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    probs = []
    for p in range(B):
        op, b, _ = problems.random_problem(n, 1234 + p)
        probs.append((op, b, None, P))
    out = {"n": n, "problems": B}
    inner = {}
    run0 = lockstep._bspg_run

    def timed_run(*a, **k):
        t = time.perf_counter()
        r = run0(*a, **k)
        inner["driver_s"] = time.perf_counter() - t
        return r
    lockstep._bspg_run = timed_run
    for rep in range(2):
        pool = [np.random.RandomState(1000 + p) for p in range(B)][::-1]

        def make():
            s = solvers.CCQPSolverSPG(1e-8, 5000)
            s.rng = pool.pop()
            return s
        torch.cuda.synchronize()
        timing = {}
        t0 = time.perf_counter()
        res = solve_batch_pipelined(make, probs, timing=timing)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    mv = sum(r.executed_matrix_vector_multiplications for r in res)
    rounds = max(r.solution_num_matrix_vector_multiplications for r in res) - 2
    out.update(problems_per_s=B / dt, matvecs_per_problem=mv / B, rounds=rounds, matrix_TBps=mv * 8.0 * n * n / dt / 1e12,
               wall_ms=dt * 1e3, staging_ms=timing["staging_s"] * 1e3, run_ms=timing["run_s"] * 1e3,
               driver_ms=inner["driver_s"] * 1e3, us_per_round=inner["driver_s"] * 1e6 / rounds,
               all_converged=all(r.solution_converged for r in res))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
