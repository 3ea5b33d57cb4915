#!/usr/bin/env python3
"""Minimal program to put behind `rocprofv3 --kernel-trace --stats --`: B problems of size n, BBPGD (tol 1e-8) and APGD
on DisjointProjOp(Sphere(3) x n/3), solved by the segmented lock-step scheduler only (batch.solve_batch_pipelined);
prints problems/s and the matrix bytes/s of both runs.     python tools/bsegs_profile_driver.py [n] [B]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd.batch import solve_batch_pipelined
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1023
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    P = ss.DisjointProjOp(*([ss.SphereProjOp(3, 0.3)] * (n // 3) + ([ss.IdentityProjOp(n % 3)] if n % 3 else [])))
    probs = []
    for p in range(B):
        op, b, _ = problems.random_problem(n, 1234 + p)
        probs.append((op, b, None, P))
    out = {"n": n, "problems": B, "set": "DisjointProjOp(%d x SphereProjOp(3))" % (n // 3)}
    for name, cls in (("BBPGD", solvers.CCQPSolverBBPGD), ("APGD", solvers.CCQPSolverAPGD)):
        solve_batch_pipelined(lambda: cls(1e-8, 5000), probs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = solve_batch_pipelined(lambda: cls(1e-8, 5000), probs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        mv = sum(r.executed_matrix_vector_multiplications for r in res)
        out[name] = {"problems_per_s": B / dt, "matvecs_per_problem": mv / B, "matrix_TBps": mv * 8.0 * n * n / dt / 1e12,
                     "all_converged": all(r.solution_converged for r in res)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
