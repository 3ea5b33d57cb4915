#!/usr/bin/env python3
"""Minimal program to put behind `rocprofv3 --kernel-trace --stats --`: B Box problems of size n solved by one of the
device-resident lock-step schedulers only (batch.solve_batch_pipelined, single_launch off); prints problems/s and the
matrix bytes/s.     python tools/bpipe_profile_driver.py [BBPGD|APGD|MPRGP|PGD|BBPGDf] [n] [B]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd.batch import solve_batch_pipelined
    name = sys.argv[1] if len(sys.argv) > 1 else "BBPGD"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    cls = getattr(solvers, "CCQPSolver" + name)
    args = (1e-6, 300) if name == "MPRGP" else (1e-8, 5000, 0.3) if name == "PGD" else (1e-8, 5000)
    P = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    probs = []
    for p in range(B):
        op, b, _ = problems.random_problem(n, 1234 + p)
        probs.append((op, b, None, P))
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = solve_batch_pipelined(lambda: cls(*args), probs, single_launch=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    mv = sum(r.executed_matrix_vector_multiplications for r in res)
    print(json.dumps({"solver": name, "n": n, "problems": B, "problems_per_s": B / dt, "matvecs_per_problem": mv / B,
                      "max_matvecs": max(r.solution_num_matrix_vector_multiplications for r in res),
                      "matrix_TBps": mv * 8.0 * n * n / dt / 1e12, "wall_ms": dt * 1e3,
                      "all_converged": all(r.solution_converged for r in res)}))


if __name__ == "__main__":
    main()
