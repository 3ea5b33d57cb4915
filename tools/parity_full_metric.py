#!/usr/bin/env python3
"""One-off evidence run: the bench workload itself (n = 131072, APGD + Box, tol 1e-8) solved by the
HIP engine AND by the CPU oracle on the same 128 GiB matrix (copied device -> host), on the GPU box.
Reports MV-count equality, the relative difference of x and both wall-clocks.  Needs ~130 GiB of host
RAM and ~2-3 minutes; not part of the test-suite for that reason (result kept under profiles/)."""
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
    from oracle import cpu_ref
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    op, b, _ = problems.random_problem(n, 1234)
    P = ss.BoxProjOp(n, -0.5, 0.5)
    with contextlib.redirect_stdout(io.StringIO()):
        solvers.CCQPSolverAPGD(1e-8, 5000).solve(op, b, convex_proj_op=P)        # warm
        t0 = time.perf_counter()
        r = solvers.CCQPSolverAPGD(1e-8, 5000).solve(op, b, convex_proj_op=P)
        t_gpu = time.perf_counter() - t0
    print("gpu done %.3f s, copying matrix to host ..." % t_gpu, flush=True)
    t0 = time.perf_counter()
    A = np.empty((n, n), dtype=np.float64)
    step = 8192
    for r0 in range(0, n, step):                     # chunked copy keeps pinned staging small
        A[r0:r0 + step] = op.A[r0:r0 + step].cpu().numpy()
        if (r0 // step) % 4 == 0:
            print("  copied rows %d / %d" % (r0 + step, n), flush=True)
    t_copy = time.perf_counter() - t0
    bb = b.cpu().numpy()
    x_gpu = r.solution.cpu().numpy()
    del op
    torch.cuda.empty_cache()
    print("copy %.1f s; running the oracle ..." % t_copy, flush=True)
    lb, ub = -0.5 * np.ones(n), 0.5 * np.ones(n)
    t0 = time.perf_counter()
    o = cpu_ref.solve_apgd(A, bb, 1e-8, 5000, P=cpu_ref.Box(n, lb, ub))
    t_cpu = time.perf_counter() - t0
    rel = float(np.linalg.norm(x_gpu - o.solution) / np.linalg.norm(o.solution))
    try:
        from threadpoolctl import threadpool_info
        thr = max([d.get("num_threads", 1) for d in threadpool_info() if d.get("user_api") == "blas"] + [1])
    except Exception:  # noqa: BLE001
        thr = None
    print(json.dumps({
        "workload": "n=%d APGD+Box(+-0.5) tol=1e-8, same matrix on both sides" % n,
        "gpu": {"reported_mv": int(r.solution_num_matrix_vector_multiplications), "wall_s": t_gpu,
                "residual": float(r.solution_residual)},
        "cpu_oracle": {"reported_mv": int(o.solution_num_matrix_vector_multiplications), "wall_s": t_cpu,
                       "residual": float(o.solution_residual), "blas_threads": thr, "host_cpus": os.cpu_count()},
        "mv_count_equal": int(r.solution_num_matrix_vector_multiplications) == int(o.solution_num_matrix_vector_multiplications),
        "rel_diff_x": rel, "within_1e-10": rel <= 1e-10, "speedup_wall": t_cpu / t_gpu,
        "matrix_copy_to_host_s": t_copy,
    }), flush=True)


if __name__ == "__main__":
    main()
