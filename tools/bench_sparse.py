#!/usr/bin/env python3
"""Sparse-operator extension of the headline workload: box-constrained QP with the sparse
SPD test matrix (k hashed symmetric off-diagonals), APGD tol 1e-8 -- one JSON line per shape with the
solve time, the matvec rate and the CSR matvec's share of algorithmic HBM bandwidth
(12 B per stored entry + 24 B per row + 8 B per column)."""
import argparse
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="1048576:8,4194304:32,16777216:8,2097152:64,2097152:26s,16777216:26s",
                    help="n:k with k off-diagonals per row; a trailing s = 27-point-stencil pattern (k = 26)")
    ap.add_argument("--cpu-n", type=int, default=1048576, help="largest n also solved by the scipy-based oracle")
    args = ap.parse_args()
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    for shape in args.shapes.split(","):
        pattern = "stencil27" if shape.endswith("s") else "hashed"
        n, k = (int(v) for v in shape.rstrip("s").split(":"))
        op, b, _ = problems.random_sparse_problem(n, k, 1234, pattern=pattern)
        P = ss.BoxProjOp(n, -0.5, 0.5)
        eng = op.engine

        def solve():
            with contextlib.redirect_stdout(io.StringIO()):
                return solvers.CCQPSolverAPGD(1e-8, 5000).solve(op, b, convex_proj_op=P)
        solve()
        torch.cuda.synchronize()
        eng.gemv_events = []
        mv0 = eng.mv_executed
        t0 = time.perf_counter()
        r = solve()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        mv = eng.mv_executed - mv0
        ms = [a.elapsed_time(b_) for a, b_ in eng.gemv_events]
        eng.gemv_events = None
        nbytes = 12 * op.nnz + 24 * n + 8 * n
        row = {"workload": "sparse box-QP, APGD tol 1e-8", "n": n, "offdiag_per_row": k, "pattern": pattern,
               "local_x_windows": int(op.summary[2]), "nnz": op.nnz,
               "wall_s": wall, "executed_mv": mv, "reported_mv": int(r.solution_num_matrix_vector_multiplications),
               "converged": bool(r.solution_converged), "matvecs_per_s": mv / wall,
               "spmv_ms": float(np.mean(ms)), "spmv_algorithmic_GBps": nbytes / np.mean(ms) / 1e6,
               "spmv_frac_of_8TBps": nbytes / np.mean(ms) / 1e6 / 8000.0,
               "solve_level_GBps": nbytes * mv / wall / 1e9}
        if n <= args.cpu_n:
            from oracle import cpu_ref, gen_ref
            A, bb, _ = gen_ref.sparse_problem(n, k, 1234, pattern=pattern)
            t0 = time.perf_counter()
            o = cpu_ref.SOLVERS["APGD"](A, bb, 1e-8, 5000, P=cpu_ref.Box(n, -0.5, 0.5))
            row["cpu_oracle_wall_s"] = time.perf_counter() - t0
            row["cpu_oracle_mv"] = int(o.solution_num_matrix_vector_multiplications)
            xg = r.solution.cpu().numpy() if isinstance(r.solution, torch.Tensor) else r.solution
            row["rel_diff_vs_oracle"] = float(np.linalg.norm(xg - o.solution) / np.linalg.norm(o.solution))
        print(json.dumps(row), flush=True)
        del op, b


if __name__ == "__main__":
    main()
