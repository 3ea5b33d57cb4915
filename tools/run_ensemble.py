#!/usr/bin/env python3
"""Measured run of the ensemble harness (counterpart of
benchmarks/benchmark_random_ccqp.py::benchmark_single_constraint / benchmark_disjoint_constraints at
sizes the reference's Wishart generator cannot reach).  Writes profiles/<tag>_ensemble.json."""
import io
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from ccqppy_amd import solvers, solution_spaces as ss
    from ccqppy_amd.ensemble import BenchmarkRandomCCQP
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    sizes = [1024, 4096, 8192]
    tol, max_mv, trials = 1e-5, 5000, 8          # tol / max_mv of the reference's benchmark (:156-158)
    slv = [solvers.CCQPSolverPGD(tol, max_mv, 0.3), solvers.CCQPSolverAPGD(tol, max_mv),
           solvers.CCQPSolverAPGDAntiRelaxation(tol, max_mv), solvers.CCQPSolverBBPGD(tol, max_mv),
           solvers.CCQPSolverBBPGDf(tol, max_mv), solvers.CCQPSolverSPG(tol, max_mv),
           solvers.CCQPSolverMPRGP(tol, max_mv)]
    fams = [[ss.IdentityProjOp(d) for d in sizes], [ss.LowerBoundProjOp(d) for d in sizes],
            [ss.UpperBoundProjOp(d) for d in sizes], [ss.SphereProjOp(d) for d in sizes],
            [ss.BoxProjOp(d) for d in sizes],
            [ss.DisjointProjOp(*[ss.SphereProjOp(3)] * (d // 3), ss.IdentityProjOp(d % 3)) for d in sizes]]
    np.random.seed(1234)
    t0 = time.perf_counter()
    # (spg_seed: SPG trial t draws from its own RandomState(1234 + t) -- the reference never seeds its harness)
    bench = BenchmarkRandomCCQP(trials, slv, fams, lanes=4, spg_seed=1234).run()
    wall = time.perf_counter() - t0
    buf = io.StringIO()
    bench.process_results(stream=buf)
    print(buf.getvalue())
    doc = bench.to_json(os.path.join("profiles", "%s_ensemble.json" % tag))
    cells = int(np.prod(bench._problem_time.shape))
    print(json.dumps({"cells": cells, "wall_s": wall, "solves_per_s": cells / wall,
                      "all_converged": bool(np.all(np.array(doc["converged"]) == 1))}))


if __name__ == "__main__":
    main()
