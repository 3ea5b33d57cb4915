#!/usr/bin/env python3
"""Single-solve latency of BBPGD and APGD at small and medium n: host-driven loop vs the pipelined form
(CCQPSolverBBPGD(..., pipeline=True): scalars on the device, iterations queued ahead) vs pipeline="graph"
(three captured iterations replayed as one hipGraph launch).  Random-SPD box QPs (13 matvecs each) and dense
contact-type Hessians (~125 matvecs).  One JSON line per problem; results are bitwise identical by construction
(tests/test_gpu_pipeline.py), only the wall-clock differs."""
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def timed(make, reps):
    def run():
        with contextlib.redirect_stdout(io.StringIO()):
            return make()
    for _ in range(3):
        r = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r


def main():
    from ccqppy_amd import problems, solvers, solution_spaces as ss
    from oracle import gen_ref
    cases = []
    for n in (256, 1024, 4096, 8192, 16384):
        op, b, _ = problems.random_problem(n, 1234)
        cases.append(("random SPD box-QP n=%d" % n, op, b, ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))))
    for m, n_ in ((300, 200), (2000, 3000)):
        A, b, _ = gen_ref.contact_problem(m, n_, 3)
        cases.append(("dense contact Hessian m=%d, x >= 0" % m, torch.as_tensor(A.toarray(), device="cuda"),
                      torch.as_tensor(b, device="cuda"), ss.LowerBoundProjOp(m, np.zeros(m))))
    for label, op, b, P in cases:
        row = {"problem": label, "solver": "BBPGD tol 1e-8"}
        for key, pipe in (("host_driven_ms", False), ("pipelined_ms", True), ("pipelined_graph_ms", "graph")):
            dt, r = timed(lambda: solvers.CCQPSolverBBPGD(1e-8, 5000, pipeline=pipe).solve(op, b, convex_proj_op=P), 20)
            row[key] = 1e3 * dt
            row["executed_mv"] = int(r.executed_matrix_vector_multiplications)
        row["speedup_pipelined"] = row["host_driven_ms"] / row["pipelined_ms"]
        print(json.dumps(row), flush=True)
        for name, cls, tol, cap in (("APGD tol 1e-8", solvers.CCQPSolverAPGD, 1e-8, 5000),
                                    ("MPRGP tol 1e-6", solvers.CCQPSolverMPRGP, 1e-6, 3000)):
            row = {"problem": label, "solver": name}
            for key, pipe in (("host_driven_ms", False), ("pipelined_ms", True)):
                dt, r = timed(lambda: cls(tol, cap, pipeline=pipe).solve(op, b, convex_proj_op=P), 10)
                row[key] = 1e3 * dt
                row["executed_mv"] = int(r.executed_matrix_vector_multiplications)
            row["speedup_pipelined"] = row["host_driven_ms"] / row["pipelined_ms"]
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
