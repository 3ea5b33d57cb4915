#!/usr/bin/env python3
"""TEST INFRASTRUCTURE / study tool (build container only: reads /root/reference).  SURVEY 8(f)2 lists a docs-only solver,
`CCQPSolverMPRGPBB`: it exists nowhere in the reference package, only as source text inside the generated documentation
(docs/src/ccqppy/solvers.html:1029-1202; its test is commented out, tests/test_module.py:69-73).  This probe executes that
snapshot as it stands (against the reference's own solution_spaces and CCQPSolverBase) on small random box QPs, 10 s per
solve, to decide whether there is a behaviour to reproduce.

Result recorded in DESIGN.md (round 3): 11 of 12 problems (n = 5 ... 200) do not terminate within 10 s -- the snapshot is a
work in progress (stale psi / beta after a CG step, a budget check only in the proportioning branch) -- so there is no
reference behaviour to be on a par with, and the solver is not part of the build.

    python oracle/probe_docs_mprgpbb.py"""
import contextlib
import html
import io
import signal
import sys
import time

import numpy as np

sys.path.insert(0, "/root/reference/src")
from ccqppy import solvers as rs, solution_spaces as rss      # noqa: E402


def snapshot_class():
    lines = open("/root/reference/docs/src/ccqppy/solvers.html").read().split("\n")
    ns = {"np": np, "time": time, "ss": rss, "CCQPSolverBase": rs.CCQPSolverBase}
    exec(html.unescape("\n".join(lines[1028:1203])), ns)       # the class as the documentation shows it
    return ns["CCQPSolverMPRGPBB"]


def main():
    K = snapshot_class()
    rng = np.random.RandomState(0)

    def too_long(*_):
        raise TimeoutError("10 s")

    signal.signal(signal.SIGALRM, too_long)
    done = hung = 0
    for n in (5, 20, 60, 200):
        for trial in range(3):
            M = rng.randn(n, n)
            A = M @ M.T + n * 0.05 * np.eye(n)
            b = A @ rng.randn(n)                                # (the snapshot minimises 1/2 x'Ax - x'b)
            P = rss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))
            s = K(1e-8, 20000)
            signal.alarm(10)
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    s.solve(A, b, convex_proj_op=P)
                    r = rs.CCQPSolverAPGD(1e-10, 200000).solve(A, -b, convex_proj_op=P)
                done += 1
                print(n, trial, "terminated: mv", s.solution_num_matrix_vector_multiplications,
                      "|x - x_APGD| = %.2e" % np.linalg.norm(s.solution - r.solution), flush=True)
            except TimeoutError:
                hung += 1
                print(n, trial, "no termination within 10 s", flush=True)
            finally:
                signal.alarm(0)
    print("terminated %d, hung %d" % (done, hung))


if __name__ == "__main__":
    main()
