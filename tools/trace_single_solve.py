#!/usr/bin/env python3
"""Driver for `rocprofv3 --kernel-trace`: a few device-resident single solves (pipeline=True) of one size, so that the
trace shows the kernels of a lock-step round of ONE problem and the gaps between them.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/trace_single_solve.py --n 4096 [--grid 0|1]"""
import argparse
import contextlib
import io
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--solver", default="APGD")
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from ccqppy_amd import lockstep, problems, solvers, solution_spaces as ss
    lockstep._APGD_GRID_SOLVE = bool(args.grid)
    n = args.n
    op, b, _ = problems.random_problem(n, 1234)
    P = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
    cls = getattr(solvers, "CCQPSolver" + args.solver)
    tolmv = (1e-6, 300) if args.solver == "MPRGP" else (1e-8, 5000)
    for _ in range(args.reps):
        with contextlib.redirect_stdout(io.StringIO()):
            r = cls(*tolmv, pipeline=True).solve(op, b, convex_proj_op=P)
        torch.cuda.synchronize()
    print(r.solution_num_matrix_vector_multiplications, lockstep.stats)


if __name__ == "__main__":
    main()
