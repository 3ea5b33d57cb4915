#!/usr/bin/env python3
"""Lock-step APGD batches BELOW the persistent matvec's sizes: the unfused rounds (two per accepted iteration: matvec,
decision, update -- twice; context option SMALL_SHARING = 1) against the fused rounds whose two products come from one
launch of the batched VALU kernel, one grid slice per product (SMALL_SHARING = 2) -- problems/s of whole solves, best of a
few repetitions, same bits asserted.  Where the fused form stops paying as the batch grows is where the library's default
(<= 256 MiB of matrices) comes from.
    python tools/bench_small_sharing.py [--shapes 256:4,256:32,...] [--solvers APGD,APGDAntiRelaxation]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="256:4,256:16,256:32,256:64,256:128,256:512,1024:4,1024:16,1024:32,1024:64,1024:100,"
                                        "2048:4,2048:16,2048:24,3072:4,3072:12")
    ap.add_argument("--solvers", default="APGD")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from bench import box_id
    from ccqppy_amd import _lib, lockstep, problems, solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    box = box_id(0)
    fl = 1 | _lib.LOCKSTEP_LOOKAHEAD | _lib.LOCKSTEP_FUSED_ROUNDS
    for shape in args.shapes.split(","):
        n, B = (int(v) for v in shape.split(":"))
        probs = []
        for p in range(B):
            op, b, _ = problems.random_problem(n, 1234 + p, device=0)
            probs.append((op, b, None, ss.BoxProjOp(n, -0.5 * np.ones(n), 0.5 * np.ones(n))))
        for kind in args.solvers.split(","):
            cls = getattr(solvers, "CCQPSolver" + kind)
            tolmv = (1e-8, 5000)
            form_of = lambda k: eng.lib.ccqp_bapgd_fused_form(eng.ctx, B, n, fl)             # noqa: E731
            out, res = {}, {}
            for label, pair in (("no_sharing", 1), ("valu_sharing", 2)):
                eng.set_option("SMALL_SHARING", pair)
                form = form_of(kind)
                best = np.inf
                for _ in range(args.reps + 1):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    r = lockstep.solve_batch_pipelined(lambda: cls(*tolmv), probs, device=0, single_launch=False)
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                out[label], res[label] = (best, form), r
            eng.set_option("SMALL_SHARING", 0)
            host = lambda x: x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)       # noqa: E731
            same = all(np.array_equal(host(a.solution).view(np.uint64), host(c.solution).view(np.uint64)) and
                       a.solution_num_matrix_vector_multiplications == c.solution_num_matrix_vector_multiplications
                       for a, c in zip(res["no_sharing"], res["valu_sharing"]))
            print(json.dumps({"n": n, "problems": B, "solver": kind,
                              "no_sharing_problems_per_s": round(B / out["no_sharing"][0], 1), "no_sharing_form": out["no_sharing"][1],
                              "valu_sharing_problems_per_s": round(B / out["valu_sharing"][0], 1),
                              "valu_sharing_form": out["valu_sharing"][1],
                              "gain": round(out["no_sharing"][0] / out["valu_sharing"][0], 3),
                              "default_form": form_of(kind),
                              "mean_mv": float(np.mean([r.solution_num_matrix_vector_multiplications for r in res["no_sharing"]])),
                              "same_bits_and_counts": bool(same), "box": box}), flush=True)
        del probs


if __name__ == "__main__":
    main()
