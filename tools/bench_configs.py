#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configs that are parity-test cases rather than the
bench line (configs 2, 3, 5 on one GPU).  Prints one JSON line per config; copy into profiles/."""
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


def mv_bytes(n):
    return 8.0 * (n * n + 2 * n)


def run(label, n, make_solver, make_op, seed=1234, repeats=2, problems_n=1):
    from ccqppy_amd import problems
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    probs = [problems.random_problem(n, seed + p) for p in range(problems_n)]
    P = make_op(n)
    best = None
    for rep in range(repeats + 1):
        torch.cuda.synchronize()
        mv0 = eng.mv_executed
        t0 = time.perf_counter()
        res = []
        for op, b, _ in probs:
            np.random.seed(1234)
            with contextlib.redirect_stdout(io.StringIO()):
                res.append(make_solver().solve(op, b, convex_proj_op=P))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        mv = eng.mv_executed - mv0
        if rep > 0 and (best is None or dt < best[0]):
            best = (dt, mv, res)
    dt, mv, res = best
    line = dict(config=label, n=n, problems=problems_n, wall_s=dt, executed_mv=mv, mv_per_s=mv / dt,
                solve_level_hbm_frac=mv * mv_bytes(n) / dt / 8e12,
                reported_mv=[int(r.solution_num_matrix_vector_multiplications) for r in res][:8],
                converged=[bool(r.solution_converged) for r in res][:8],
                residual=[float(r.solution_residual) for r in res][:4])
    print(json.dumps(line), flush=True)
    del probs
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="2,3,5")
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    from ccqppy_amd import solvers, solution_spaces as ss
    which = args.which.split(",")
    if "2" in which:
        run("config2: n=32768 APGD+Box tol=1e-8", 32768, lambda: solvers.CCQPSolverAPGD(1e-8, 5000),
            lambda n: ss.BoxProjOp(n, -0.5, 0.5))
    if "3" in which:
        run("config3: n=131072 BBPGD+Cone max_mv=64", 131072, lambda: solvers.CCQPSolverBBPGD(1e-8, 64),
            lambda n: ss.ConeProjOp(n), repeats=1)
    if "4s" in which:
        run("config4 (1-GPU stand-in): n=65536 SPG+Disjoint(Box,Cone) max_mv=64", 65536,
            lambda: solvers.CCQPSolverSPG(1e-8, 64),
            lambda n: ss.DisjointProjOp(ss.BoxProjOp(n // 2, -0.5, 0.5), ss.ConeProjOp(n // 2)))
    if "5" in which:
        run("config5 slice: %d x (n=8192 MPRGP+Box tol=1e-6), sequential on 1 GPU" % args.batch, 8192,
            lambda: solvers.CCQPSolverMPRGP(1e-6, 300),
            lambda n: ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n)), problems_n=args.batch)
    if "5l" in which:
        from ccqppy_amd import problems
        from ccqppy_amd.batch import solve_batch
        from ccqppy_amd._engine import get_engine
        n = 8192
        probs = []
        for p in range(args.batch):
            op, b, _ = problems.random_problem(n, 1234 + p)
            probs.append((op, b, None, ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))))
        for lanes in (1, 2, 4, 8):
            best = None
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = solve_batch(lambda: solvers.CCQPSolverMPRGP(1e-6, 300), probs, lanes=lanes)
                dt = time.perf_counter() - t0
                if rep and (best is None or dt < best):
                    best = dt
            mv = sum(r.executed_matrix_vector_multiplications for r in res)
            print(json.dumps(dict(config="config5 slice via solve_batch: %d x n=8192 MPRGP" % args.batch, lanes=lanes,
                                  wall_s=best, problems_per_s=args.batch / best, executed_mv=mv, mv_per_s=mv / best,
                                  solve_level_hbm_frac=mv * mv_bytes(n) / best / 8e12,
                                  reported_mv=[int(r.solution_num_matrix_vector_multiplications) for r in res][:8])),
                  flush=True)
    if "5f" in which:
        from ccqppy_amd import problems
        from ccqppy_amd.batch import solve_batch_fused
        n = 8192
        probs = []
        for p in range(args.batch):
            op, b, _ = problems.random_problem(n, 1234 + p)
            probs.append((op, b, None, ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))))
        for lanes, groups in ((8, 1), (8, 2), (16, 2), (32, 2), (32, 4)):
            if lanes > args.batch:
                continue
            best = None
            stats = {}
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = solve_batch_fused(lambda: solvers.CCQPSolverMPRGP(1e-6, 300), probs, lanes=lanes,
                                        groups=groups, stats=stats)
                dt = time.perf_counter() - t0
                if rep and (best is None or dt < best):
                    best = dt
            mv = sum(r.executed_matrix_vector_multiplications for r in res)
            print(json.dumps(dict(config="config5 slice via solve_batch_fused: %d x n=8192 MPRGP" % args.batch,
                                  lanes=lanes, groups=groups, wall_s=best, problems_per_s=args.batch / best,
                                  executed_mv=mv, mv_per_s=mv / best,
                                  solve_level_hbm_frac=mv * mv_bytes(n) / best / 8e12, stats=stats,
                                  reported_mv=[int(r.solution_num_matrix_vector_multiplications) for r in res][:8])),
                  flush=True)
    if "5b" in which:
        for name, mk in (("BBPGD", lambda: solvers.CCQPSolverBBPGD(1e-8, 5000)),
                         ("APGD", lambda: solvers.CCQPSolverAPGD(1e-8, 5000)),
                         ("SPG", lambda: solvers.CCQPSolverSPG(1e-8, 5000))):
            run("n=8192 %s+Box x%d" % (name, args.batch), 8192, mk,
                lambda n: ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n)), problems_n=args.batch)


if __name__ == "__main__":
    main()
