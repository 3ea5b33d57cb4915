#!/usr/bin/env python3
"""Matrix-free contact-type QP: A = D diag(minv) D^T + eps I with D a sparse m x n
constraint Jacobian, never formed (operators.ComposedOperator), x >= 0, BBPGD (or --solver APGD / MPRGP) tol 1e-8,
host-driven loop vs pipeline=True (scalars and branch decisions on the device).
One JSON line per shape m:n; the composed matvec moves 2 CSR matvecs + 4 vector passes."""
import argparse
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def device_jacobian(m, n, per_row, seed, dev):
    """m x n CSR with exactly per_row entries per row at hashed columns (duplicates allowed: they just add)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    col = torch.randint(0, n, (m, per_row), device=dev, generator=g, dtype=torch.int64)
    # a constraint touches nearby bodies: keep the columns within a window around m-proportional position
    centre = (torch.arange(m, device=dev, dtype=torch.int64) * n // m)[:, None]
    col = torch.remainder(centre + torch.remainder(col, 64) - 32, n)
    col = torch.sort(col, dim=1).values
    val = torch.rand((m, per_row), device=dev, generator=g, dtype=torch.float64) * 2 - 1
    rowptr = torch.arange(m + 1, device=dev, dtype=torch.int64) * per_row
    return rowptr, col.reshape(-1).to(torch.int32), val.reshape(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="20000:30000,200000:300000,2000000:3000000,8000000:12000000")
    ap.add_argument("--per-row", type=int, default=6)
    ap.add_argument("--solver", default="BBPGD", choices=["BBPGD", "APGD", "MPRGP"])
    ap.add_argument("--tol", type=float, default=1e-8)
    args = ap.parse_args()
    from ccqppy_amd import solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    from ccqppy_amd.operators import ComposedOperator, DiagonalOperator, SparseOperator
    eng = get_engine(0)
    eng.bind()
    for shape in args.shapes.split(","):
        m, n = (int(v) for v in shape.split(":"))
        D = SparseOperator(device_jacobian(m, n, args.per_row, 1, eng.device), shape=(m, n), engine=eng)
        gen = torch.Generator(device=eng.device)
        gen.manual_seed(7)
        minv = 0.5 + torch.rand(n, device=eng.device, dtype=torch.float64, generator=gen)
        op = ComposedOperator(D, DiagonalOperator(minv), D.T, shift=0.05)
        xs = torch.rand(m, device=eng.device, dtype=torch.float64, generator=gen) * 2 - 1
        b = -op.dot(xs)
        # (MPRGP indexes the bounds: array bounds, as the reference requires)
        P = ss.LowerBoundProjOp(m, torch.zeros(m, dtype=torch.float64).numpy() if args.solver == "MPRGP" else 0.0)
        cls = getattr(solvers, "CCQPSolver" + args.solver)

        def solve(pipe=False):
            with contextlib.redirect_stdout(io.StringIO()):
                return cls(args.tol, 20000, pipeline=pipe).solve(op, b, convex_proj_op=P)
        walls = {}
        for pipe in (False, True):
            solve(pipe)
            torch.cuda.synchronize()
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                r = solve(pipe)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            walls[pipe] = best
        wall = walls[False]
        mv = int(r.executed_matrix_vector_multiplications)
        nnz = D.nnz
        nbytes = 2 * (12 * nnz + 16 * max(m, n) + 8 * min(m, n)) + 8 * (3 * n + 3 * m)
        print(json.dumps({"workload": "contact QP: A = D diag D^T + 0.05 I (matrix-free), x >= 0, %s tol %g" % (args.solver, args.tol),
                          "constraints_m": m, "coordinates_n": n, "nnz_D": nnz, "wall_s": wall, "executed_mv": mv,
                          "reported_mv": int(r.solution_num_matrix_vector_multiplications),
                          "converged": bool(r.solution_converged), "residual": float(r.solution_residual),
                          "wall_s_pipelined": walls[True], "speedup_pipelined": walls[False] / walls[True],
                          "ms_per_composed_matvec_incl_vector_work": 1e3 * wall / mv,
                          "solve_level_GBps_of_operator_bytes": nbytes * mv / wall / 1e9}), flush=True)
        del op, D


if __name__ == "__main__":
    main()
