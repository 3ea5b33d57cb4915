#!/usr/bin/env python3
"""Throughput of MANY SMALL dense BBPGD (--apgd: APGD, --mprgp: MPRGP) solves (Box, tol 1e-8; MPRGP: 1e-6 / 300 matvecs) under every scheduler of ccqppy_amd:
sequential solve(), sequential with the pipelined iteration, the stream-lane and fused batch schedulers
(host-driven scalars) and solve_batch_pipelined (lock-step, scalars on the device) and, for BBPGD up to n = 2048, its single-launch form (one workgroup per problem).  One JSON line per size.
   python tools/bench_small_batch.py [--apgd | --mprgp | --spg] [--set box|sphere|spheres3] [n:B ...]
--spg: SPG-QP, every solver with its own RandomState(1000 + p) (what the lock-step form needs; the other schedulers get
the same generators, so all columns solve the same thing bit for bit).
--set sphere / spheres3 (BBPGD): a whole-vector Sphere / a DisjointUnion of Sphere(3) blocks instead of the Box -- the
segmented lock-step form (ccqp_bsegs.hip); no single-launch column there."""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from ccqppy_amd.batch import solve_batch, solve_batch_fused, solve_batch_pipelined
    argv = list(sys.argv[1:])
    setname = "box"
    if "--set" in argv:
        i = argv.index("--set")
        setname = argv[i + 1]
        del argv[i:i + 2]
    args = [a for a in argv if a not in ("--apgd", "--mprgp", "--spg")]
    mprgp = "--mprgp" in sys.argv[1:]
    spg = "--spg" in sys.argv[1:]
    apgd = "--apgd" in sys.argv[1:] or mprgp or spg   # (no single-solve pipeline column for these)
    cls = solvers.CCQPSolverSPG if spg else solvers.CCQPSolverMPRGP if mprgp else \
        solvers.CCQPSolverAPGD if apgd else solvers.CCQPSolverBBPGD
    sizes = [tuple(int(v) for v in a.split(":")) for a in args] or \
        [(64, 1024), (256, 1024), (1024, 512), (2048, 256), (4096, 128)]
    for n, B in sizes:
        probs = []
        for p in range(B):
            op, b, _ = problems.random_problem(n, 1234 + p)
            if setname == "sphere":
                P = ss.SphereProjOp(n, 2.0)
            elif setname == "spheres3":
                P = ss.DisjointProjOp(*([ss.SphereProjOp(3, 0.3)] * (n // 3) + ([ss.IdentityProjOp(n % 3)] if n % 3 else [])))
            else:
                P = ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
            probs.append((op, b, None, P))
        tol, cap = (1e-6, 300) if mprgp else (1e-8, 5000)          # MPRGP: BASELINE config 5's settings
        pool = []                          # generators made OUTSIDE the timed region (seeding one costs ~110 us)

        def refill():
            if spg:
                pool[:] = [np.random.RandomState(1000 + p) for p in range(B)][::-1]

        class mk_factory:
            """() -> a fresh solver; the p-th SPG solver of a run draws from RandomState(1000 + p)"""
            def __call__(self):
                s = cls(tol, cap)
                if spg:
                    s.rng = pool.pop()
                return s

        def seq(pipe):
            mk = mk_factory()
            for op, b, x0, P in probs:
                with contextlib.redirect_stdout(io.StringIO()):
                    solver = mk()
                    if pipe:
                        solver.pipeline = True
                    solver.solve(op, b, convex_proj_op=P)

        out = {}
        res = {}
        for name, fn in (("sequential", lambda: seq(False)),) + \
                ((() if apgd and not spg else (("sequential_pipelined", lambda: seq(True)),))) + (
                         ("stream_lanes_8", lambda: solve_batch(mk_factory(), probs, lanes=8)),
                         ("fused_32x2", lambda: solve_batch_fused(mk_factory(), probs, lanes=32, groups=2)),
                         ("batch_pipelined", lambda: solve_batch_pipelined(mk_factory(), probs, single_launch=False))) + \
                (() if apgd or n > 2048 or setname != "box" else
                 (("batch_single_launch", lambda: solve_batch_pipelined(mk_factory(), probs, single_launch=True)),)):
            refill()
            fn()
            torch.cuda.synchronize()
            best = 0.0
            for _ in range(3):
                refill()
                t0 = time.perf_counter()
                r = fn()
                torch.cuda.synchronize()
                best = max(best, B / (time.perf_counter() - t0))
            out[name] = round(best, 1)
            if r is not None:
                res[name] = r
        mvs = [s.solution_num_matrix_vector_multiplications for s in res["batch_pipelined"]]
        same = all(torch.equal(a.solution, c.solution) for a, c in zip(res["fused_32x2"], res["batch_pipelined"]))
        if "batch_single_launch" in res:
            same = same and all(torch.equal(a.solution, c.solution)
                                for a, c in zip(res["fused_32x2"], res["batch_single_launch"]))
        # bytes the matvecs of the batch must read: executed matvecs per problem, 8 n^2 bytes each
        mv_bytes = sum(s.executed_matrix_vector_multiplications * 8.0 * n * n for s in res["batch_pipelined"])
        print(json.dumps({"solver": cls.__name__, "set": setname, "n": n, "problems": B, "problems_per_s": out, "mv_per_problem_mean": float(np.mean(mvs)),
                          "mv_per_problem_max": int(max(mvs)), "bitwise_equal_to_fused": bool(same),
                          "batch_pipelined_matrix_GBps": round(mv_bytes * out["batch_pipelined"] / B / 1e9, 1)}),
              flush=True)


if __name__ == "__main__":
    main()
