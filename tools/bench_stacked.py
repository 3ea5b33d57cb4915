#!/usr/bin/env python3
"""Throughput of batch.solve_stacked: B dense box-QPs of size n as stacked tensors (random SPD matrices from the
device generator, Box +-0.5, tol 1e-8; MPRGP 1e-6 / 300 matvecs), no per-problem host work.  One JSON line per
(solver, n): problems/s, matvecs per problem, bytes of matrix the matvecs read per second.
   python tools/bench_stacked.py [--solvers BBPGD,APGD,MPRGP,SPG] [--set box|spheres3|sphere] [n:B ...]
--set spheres3 / sphere: one shared DisjointUnion of Sphere(3) blocks / whole-vector Sphere instead of the Box
(solve_stacked(feasible_set=...): the segmented lock-step forms; no MPRGP, no single-launch form).
SPG (lock-step only, ccqp_bspg.hip): problem p draws from its own RandomState, created outside the timed region."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from ccqppy_amd import batch
    from ccqppy_amd._engine import get_engine
    args = sys.argv[1:]
    names = ["BBPGD", "APGD", "MPRGP"]
    if args and args[0] == "--solvers":
        names, args = args[1].split(","), args[2:]
    setname = "box"
    if args and args[0] == "--set":
        setname, args = args[1], args[2:]
    sizes = [tuple(int(v) for v in a.split(":")) for a in args] or \
        [(32, 32768), (64, 16384), (128, 8192), (256, 4096), (512, 2048), (1024, 512), (2048, 256)]
    eng = get_engine(0)
    eng.bind()
    for n, B in sizes:
        A = torch.empty((B, n, n), dtype=torch.float64, device="cuda")
        xs = torch.empty((B, n), dtype=torch.float64, device="cuda")
        for p in range(B):                               # problem p = the generator's problem (n, seed 1234 + p)
            eng.gen_spd_rows(A[p], 0, n, 1234 + p)
            eng.gen_vec(xs[p], 1234 + p)
        b = -torch.bmm(A, xs.unsqueeze(2)).squeeze(2)
        lo, up = -0.5 * np.ones(n), 0.5 * np.ones(n)
        setkw = {"lower": lo, "upper": up}
        if setname != "box":
            from ccqppy_amd import solution_spaces as ss
            P = ss.SphereProjOp(n, 2.0) if setname == "sphere" else ss.DisjointProjOp(
                *([ss.SphereProjOp(3, 0.3)] * (n // 3) + ([ss.IdentityProjOp(n % 3)] if n % 3 else [])))
            setkw = {"feasible_set": P}
        for name in names:
            if name == "MPRGP" and (n <= 64 or setname != "box"):
                continue
            tol, cap = (1e-6, 300) if name == "MPRGP" else (1e-8, 5000)
            forms = [False, True] if n <= 2048 and name != "SPG" and setname == "box" else [False]
            for single in forms:
                kw = {} if single is None else {"single_launch": single}
                if name == "SPG":
                    kw["rngs"] = [np.random.RandomState(p) for p in range(B)]
                r = batch.solve_stacked(name, A, b, tol=tol, max_mv=cap, **setkw, **kw)
                torch.cuda.synchronize()
                best = 0.0
                for _ in range(3):
                    if name == "SPG":
                        kw["rngs"] = [np.random.RandomState(p) for p in range(B)]
                    t0 = time.perf_counter()
                    r = batch.solve_stacked(name, A, b, tol=tol, max_mv=cap, **setkw, **kw)
                    torch.cuda.synchronize()
                    best = max(best, B / (time.perf_counter() - t0))
                ex = r.executed_matrix_vector_multiplications
                print(json.dumps({"solver": name, "set": setname, "form": {None: "lock-step", False: "lock-step", True: "single launch"}[single],
                                  "n": n, "problems": B, "problems_per_s": round(best, 1),
                                  "converged": int(r.converged.sum()), "executed_mv_mean": float(ex.mean()),
                                  "matrix_GBps": round(float(ex.sum()) * 8.0 * n * n / B * best / 1e9, 1)}), flush=True)
        del A, b, xs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
