#!/usr/bin/env python3
"""One-off evidence run: full-size workloads (n = 131072) solved by the HIP engine AND by the CPU oracle on
the same 128 GiB matrix (copied device -> host once), on the GPU box:
  metric   n=131072 APGD + Box(+-0.5), tol 1e-8                 (the bench workload)
  config3  n=131072 BBPGD + Cone, 64-MV budget                  (BASELINE config 3)
Reports MV-count equality, the relative difference of x and both wall-clocks, one JSON line per case.
Needs ~130 GiB of host RAM and ~4 minutes; not part of the test-suite for that reason (results kept under
profiles/).   python tools/parity_full_metric.py [n] [metric,config3]"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = {"metric": ("APGD", "box", 1e-8, 5000), "config3": ("BBPGD", "cone", 1e-8, 64)}


def main():
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from oracle import cpu_ref
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    which = (sys.argv[2] if len(sys.argv) > 2 else "metric,config3").split(",")
    op, b, _ = problems.random_problem(n, 1234)
    gpu = {}
    for name in which:
        sname, oname, tol, mx = CASES[name]
        P = ss.BoxProjOp(n, -0.5, 0.5) if oname == "box" else ss.ConeProjOp(n)
        cls = getattr(solvers, "CCQPSolver" + sname)
        with contextlib.redirect_stdout(io.StringIO()):
            cls(tol, mx).solve(op, b, convex_proj_op=P)        # warm
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = cls(tol, mx).solve(op, b, convex_proj_op=P)
            torch.cuda.synchronize()
            gpu[name] = (r, time.perf_counter() - t0)
        print("gpu %s done %.3f s" % (name, gpu[name][1]), flush=True)
    print("copying matrix to host ...", flush=True)
    t0 = time.perf_counter()
    A = np.empty((n, n), dtype=np.float64)
    step = 8192
    for r0 in range(0, n, step):                     # chunked copy keeps pinned staging small
        A[r0:r0 + step] = op.A[r0:r0 + step].cpu().numpy()
        if (r0 // step) % 4 == 0:
            print("  copied rows %d / %d" % (r0 + step, n), flush=True)
    t_copy = time.perf_counter() - t0
    bb = b.cpu().numpy()
    xg = {k: v[0].solution.cpu().numpy() for k, v in gpu.items()}
    del op
    torch.cuda.empty_cache()
    try:
        from threadpoolctl import threadpool_info
        thr = max([d.get("num_threads", 1) for d in threadpool_info() if d.get("user_api") == "blas"] + [1])
    except Exception:  # noqa: BLE001
        thr = None
    for name in which:
        sname, oname, tol, mx = CASES[name]
        print("copy %.1f s; running the oracle for %s ..." % (t_copy, name), flush=True)
        P = cpu_ref.Box(n, -0.5 * np.ones(n), 0.5 * np.ones(n)) if oname == "box" else cpu_ref.Cone(n)
        t0 = time.perf_counter()
        o = cpu_ref.SOLVERS[sname](A, bb, tol, mx, P=P)
        t_cpu = time.perf_counter() - t0
        r, t_gpu = gpu[name]
        rel = float(np.linalg.norm(xg[name] - o.solution) / np.linalg.norm(o.solution))
        print(json.dumps({
            "workload": "n=%d %s+%s tol=%g max_mv=%d, same matrix on both sides" % (n, sname, oname, tol, mx),
            "gpu": {"reported_mv": int(r.solution_num_matrix_vector_multiplications), "wall_s": t_gpu,
                    "executed_mv": int(r.executed_matrix_vector_multiplications),
                    "residual": float(r.solution_residual), "converged": bool(r.solution_converged)},
            "cpu_oracle": {"reported_mv": int(o.solution_num_matrix_vector_multiplications), "wall_s": t_cpu,
                           "executed_mv": int(o.executed_mv), "residual": float(o.solution_residual),
                           "converged": bool(o.solution_converged), "blas_threads": thr, "host_cpus": os.cpu_count()},
            "mv_count_equal": int(r.solution_num_matrix_vector_multiplications) == int(o.solution_num_matrix_vector_multiplications),
            "rel_diff_x": rel, "within_1e-10": rel <= 1e-10, "speedup_wall": t_cpu / t_gpu,
            "matrix_copy_to_host_s": t_copy,
        }), flush=True)


if __name__ == "__main__":
    main()
