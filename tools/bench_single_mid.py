#!/usr/bin/env python3
"""One dense problem of 2048 <= n <= 16384 (where the reference's users live: benchmarks/benchmark_random_ccqp.py:156,187,
tests/test_module.py): the host-driven loop (one blocking scalar read per iteration, ~25-40 us of host turn-around around
~27-90 us of matvec) against the device-resident form `pipeline=True` (the lock-step scheduler with a batch of ONE: every
scalar and branch on the device, rounds queued ahead; from n = 4096 on its matvec is the persistent multi-vector launch).
Same bits and counts asserted; wall-clock per solve: median and best of `reps`.  Prints JSON lines.
    python tools/bench_single_mid.py [--sizes 2048,4096,8192,16384] [--solvers APGD,APGDAntiRelaxation,MPRGP]"""
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
    ap.add_argument("--sizes", default="2048,4096,8192,16384")
    ap.add_argument("--solvers", default="APGD,APGDAntiRelaxation,MPRGP")
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--grid", type=int, default=0, help="0: lock-step rounds instead of the one-launch cooperative solve")
    ap.add_argument("--ahead", type=int, default=0, help="A/B: rounds queued beyond the device's last report (4)")
    ap.add_argument("--cap", type=int, default=0, help="A/B: largest n of the device-resident form of PGD / BBPGD(f) / MPRGP")
    add_option_args(ap)
    args = ap.parse_args()
    options = apply_options(args)
    from bench import box_id
    from ccqppy_amd import lockstep, problems, solvers, solution_spaces as ss
    box = box_id(0)
    lockstep._APGD_GRID_SOLVE = bool(args.grid)
    if args.ahead:
        lockstep._ONE_AHEAD = args.ahead
    if args.cap:
        solvers.CCQPSolverBase._LOCKSTEP_ONE_MAX_N = args.cap
        solvers.CCQPSolverMPRGP._PIPELINE_MAX_N = args.cap
        solvers.CCQPSolverSPG._PIPELINE_MAX_N = args.cap
    for n in (int(v) for v in args.sizes.split(",")):
        op, b, _ = problems.random_problem(n, 1234)
        P = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
        for name in args.solvers.split(","):
            cls = getattr(solvers, "CCQPSolver" + name)
            tolmv = (1e-6, 300) if name == "MPRGP" else (1e-8, 100000, 0.3) if name == "PGD" else (1e-8, 5000)
            out = {}
            for label, kw in (("host_loop", {"pipeline": False}), ("device_resident", {"pipeline": True})):
                ts, r = [], None
                fb0 = lockstep.stats["fallback_solves"]
                for rep in range(args.reps + 2):
                    np.random.seed(1234)                 # (SPG draws from numpy's global generator)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    with contextlib.redirect_stdout(io.StringIO()):
                        r = cls(*tolmv, **kw).solve(op, b, convex_proj_op=P)
                    torch.cuda.synchronize()
                    if rep >= 2:
                        ts.append(time.perf_counter() - t0)
                out[label] = (ts, r, lockstep.stats["fallback_solves"] - fb0)
            a, c = out["host_loop"][1], out["device_resident"][1]
            same = bool(torch.equal(a.solution.view(torch.int64), c.solution.view(torch.int64))) and \
                a.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications and \
                a.executed_matrix_vector_multiplications == c.executed_matrix_vector_multiplications and \
                float(a.solution_residual) == float(c.solution_residual)
            h, d = out["host_loop"][0], out["device_resident"][0]
            print(json.dumps({"n": n, "solver": name, "reported_mv": int(a.solution_num_matrix_vector_multiplications),
                              "executed_mv": int(a.executed_matrix_vector_multiplications),
                              "host_loop_ms": round(1e3 * float(np.median(h)), 4), "host_loop_best_ms": round(1e3 * min(h), 4),
                              "device_resident_ms": round(1e3 * float(np.median(d)), 4),
                              "device_resident_best_ms": round(1e3 * min(d), 4),
                              "speedup": round(float(np.median(h)) / float(np.median(d)), 3),
                              "same_bits_and_counts": same, "device_resident_fell_back": out["device_resident"][2],
                              "converged": bool(a.solution_converged), "options": options,
                              "device_resident_host_split_us": {k: round(1e6 * v, 1) for k, v in lockstep.last_timing.items()}
                              if name.startswith("APGD") else None, "grid_solves": lockstep.stats.get("grid_solves"),
                              "box": box.get("host"), "gpu_uuid": box.get("gpu_uuid")}), flush=True)
        del op
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
