#!/usr/bin/env python3
"""Where the HOST time of one solve_batch_pipelined call goes (cProfile; the device queue runs by itself meanwhile):
    python3 tools/host_profile_lockstep.py [--problems 128] [--size 8192] [--solver MPRGP] [--calls 3]
Prints the wall-clock of each call, the `timing` split and the top functions by own and cumulative time."""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=128)
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--solver", default="MPRGP")
    ap.add_argument("--calls", type=int, default=3)
    ap.add_argument("--shared-set", action="store_true", help="one BoxProjOp object for the whole batch")
    args = ap.parse_args()
    from ccqppy_amd import lockstep, problems, solvers, solution_spaces as ss
    n, B = args.size, args.problems
    lb, ub = -0.5 * np.ones(n), 0.5 * np.ones(n)
    shared = ss.BoxProjOp(n, lb, ub)
    probs = []
    for p in range(B):
        op, b, _ = problems.random_problem(n, 1234 + p, device=0)
        probs.append((op, b, None, shared if args.shared_set else ss.BoxProjOp(n, lb, ub)))
    kind = getattr(solvers, "CCQPSolver" + args.solver)
    make = (lambda: kind(1e-6, 300)) if args.solver in ("MPRGP", "APGD") else (lambda: kind(1e-6, 2000))
    lockstep.solve_batch_pipelined(make, probs, device=0)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    for c in range(args.calls):
        tm = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if c == args.calls - 1:
            pr.enable()
        lockstep.solve_batch_pipelined(make, probs, device=0, timing=tm)
        pr.disable()
        torch.cuda.synchronize()
        print("call %d: %.3f ms   %s" % (c, 1e3 * (time.perf_counter() - t0), tm), flush=True)
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
        print(s.getvalue())


if __name__ == "__main__":
    main()
