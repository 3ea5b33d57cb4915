#!/usr/bin/env python3
"""Config 5's 1-GPU share (128 x n = 8192 MPRGP + Box) under the lock-step scheduler's knobs: the batch as one run or
as two alternating halves, rounds queued ahead, and the batched matvec's tuning word (rows per wave, workgroups per
problem) -- one JSON line per variant, problems generated once.  Same bits for every variant (asserted).
    python tools/sweep_config5.py [--problems 128] [--size 8192] [--reps 3]"""
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
    ap.add_argument("--problems", type=int, default=128)
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=3)
    add_option_args(ap)
    args = ap.parse_args()
    options = apply_options(args)                  # (per-context tuning options: the A/B switches, tools/README.md)
    from ccqppy_amd import lockstep, problems, solvers, solution_spaces as ss
    n, B = args.size, args.problems
    lb, ub = -0.5 * np.ones(n), 0.5 * np.ones(n)
    probs = []
    for p in range(B):
        op, b, _ = problems.random_problem(n, 1234 + p, device=0)
        probs.append((op, b, None, ss.BoxProjOp(n, lb, ub)))
    make = lambda: solvers.CCQPSolverMPRGP(1e-6, 300)       # noqa: E731
    variants = [("default (persistent LDS-DMA matvec, 4 waves per CU, rows 1 MiB apart; two chained halves)", dict(halves=True, ahead=4, tune=0)),
                ("VALU matvec, two rows x two steps per wave (the default until the row-spacing probe)", dict(halves=False, ahead=4, tune=3 << 4)),
                ("round 2's matvec (two rows x one step per wave)", dict(halves=False, ahead=4, tune=7 << 4)),
                ("persistent LDS-DMA matvec, 2 waves per CU", dict(halves=False, ahead=4, tune=(3 << 1) | (4 << 4))),
                ("persistent LDS-DMA matvec, R 2, S 8, 4 waves per CU", dict(halves=False, ahead=4, tune=(3 << 1) | (6 << 4))),
                ("one run instead of two chained halves", dict(halves=False, ahead=4, tune=0)),
                ("default (repeat)", dict(halves=True, ahead=4, tune=0)),
                ("VALU matvec (repeat)", dict(halves=False, ahead=4, tune=3 << 4))]
    ref = None
    for label, kw in variants:
        lockstep._HALVES_DEFAULT = bool(kw["halves"])
        lockstep._HALVES_CHAINED = bool(kw.get("chained", True))
        best = None
        for rep in range(args.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = lockstep.solve_batch_pipelined(make, probs, device=0, ahead=kw["ahead"], tune=kw["tune"])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep and (best is None or dt < best):
                best = dt
        x = torch.stack([r.solution for r in res])
        if ref is None:
            ref = x
        mv = sum(r.executed_matrix_vector_multiplications for r in res)
        print(json.dumps({"options": options, "variant": label, **kw, "n": n, "problems": B, "wall_s": best, "problems_per_s": B / best,
                          "executed_mv": int(mv), "solve_level_hbm_frac": mv * 8.0 * (n * n + 2 * n) / best / 8e12,
                          "same_bits_as_first": bool(torch.equal(x.view(torch.int64), ref.view(torch.int64))),
                          "converged": int(sum(bool(r.solution_converged) for r in res))}), flush=True)


if __name__ == "__main__":
    main()
