#!/usr/bin/env python3
"""A/B of ONE context option on whole lock-step solves: problems/s of `solve_batch_pipelined` for each shape n:B under
every value given, best of a few repetitions, same bits and counts asserted across the values.
    python tools/ab_lockstep_option.py --solver MPRGP --option MPRGP_FUSED_TAIL --values 0,2 --shapes 256:1,256:8,1024:4
    python tools/ab_lockstep_option.py --solver BBPGD --option SINGLE_LAUNCH --values 0,1 --shapes 128:4,256:16,512:64
(SINGLE_LAUNCH is not a context option but the `single_launch` argument: rounds against one workgroup per problem)"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--solver", default="MPRGP")
    ap.add_argument("--option", required=True)
    ap.add_argument("--values", default="0,2")
    ap.add_argument("--shapes", default="256:1,256:8,256:32,1024:1,1024:8,2048:1,2048:4")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from bench import box_id
    from ccqppy_amd import lockstep, problems, solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    box = box_id(0)
    cls = getattr(solvers, "CCQPSolver" + args.solver)
    tolmv = (1e-6, 300) if args.solver == "MPRGP" else (1e-8, 5000)
    values = [int(v) for v in args.values.split(",")]
    host = lambda x: x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)       # noqa: E731
    for shape in args.shapes.split(","):
        n, B = (int(v) for v in shape.split(":"))
        probs = []
        for p in range(B):
            op, b, _ = problems.random_problem(n, 1234 + p, device=0)
            probs.append((op, b, None, ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))))
        out, res = {}, {}
        try:
            for v in values:
                # (the pseudo-option SINGLE_LAUNCH: 0 = lock-step rounds, 1 = one workgroup per problem, one launch)
                # (DEFAULT: nothing set, `single_launch=None` -- the library's own choice of form)
                single = bool(v) if args.option == "SINGLE_LAUNCH" else None if args.option == "DEFAULT" else False
                if args.option not in ("SINGLE_LAUNCH", "DEFAULT"):
                    eng.set_option(args.option, v)
                best = np.inf
                for _ in range(args.reps + 1):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    r = lockstep.solve_batch_pipelined(lambda: cls(*tolmv), probs, device=0, single_launch=single)
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                out[v], res[v] = best, r
        finally:
            if args.option not in ("SINGLE_LAUNCH", "DEFAULT"):
                eng.set_option(args.option, 0)
        same = all(np.array_equal(host(a.solution).view(np.uint64), host(c.solution).view(np.uint64)) and
                   a.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
                   for v in values[1:] for a, c in zip(res[values[0]], res[v]))
        print(json.dumps({"solver": args.solver, "option": args.option, "n": n, "problems": B,
                          "problems_per_s": {str(v): round(B / out[v], 1) for v in values},
                          "ms_per_batch": {str(v): round(1e3 * out[v], 4) for v in values},
                          "same_bits_and_counts": bool(same), "box": box}), flush=True)
        del probs


if __name__ == "__main__":
    main()
