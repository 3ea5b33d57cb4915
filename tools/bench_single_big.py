#!/usr/bin/env python3
"""One big dense problem, host-driven solvers, with and without the one-pass multi-vector matvecs:
APGD (pair_matvecs: A.x_{k+1} and A.y_{k+1} in one pass) and MPRGP (look_ahead: A.dx, A.x_{k+1} + b, A.p_{k+1} in one
pass).  Same bits and counts asserted; wall-clock per solve, best of a few repetitions.
    python tools/bench_single_big.py [--sizes 32768,65536]"""
import argparse
import contextlib
import io
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
    ap.add_argument("--sizes", default="32768,65536")
    ap.add_argument("--reps", type=int, default=3)
    add_option_args(ap)
    args = ap.parse_args()
    options = apply_options(args)                  # (per-context tuning options: the A/B switches, tools/README.md)
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    for n in (int(v) for v in args.sizes.split(",")):
        op, b, _ = problems.random_problem(n, 1234)
        P = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
        for name, make in (("APGD", lambda on: solvers.CCQPSolverAPGD(1e-8, 5000, pair_matvecs=on)),
                           ("MPRGP", lambda on: solvers.CCQPSolverMPRGP(1e-6, 300, look_ahead=on))):
            out = {}
            for on in (False, True):
                best, r = None, None
                for rep in range(args.reps + 1):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    with contextlib.redirect_stdout(io.StringIO()):
                        r = make(on).solve(op, b, convex_proj_op=P)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    if rep and (best is None or dt < best):
                        best = dt
                out[on] = (best, r)
            a, c = out[False][1], out[True][1]
            same = bool(torch.equal(a.solution.view(torch.int64), c.solution.view(torch.int64))) and \
                a.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications and \
                a.executed_matrix_vector_multiplications == c.executed_matrix_vector_multiplications
            print(json.dumps({"n": n, "options": options, "solver": name, "reported_mv": int(a.solution_num_matrix_vector_multiplications),
                              "executed_mv": int(a.executed_matrix_vector_multiplications),
                              "wall_ms_one_pass_per_matvec": 1e3 * out[False][0], "wall_ms_multi_vector_passes": 1e3 * out[True][0],
                              "speedup": out[False][0] / out[True][0], "same_bits_and_counts": same,
                              "converged": bool(a.solution_converged)}), flush=True)
        del op
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
