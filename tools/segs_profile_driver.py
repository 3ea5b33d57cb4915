#!/usr/bin/env python3
"""Minimal program to put behind `rocprofv3 --kernel-trace --stats --`: the projection onto
DisjointProjOp(*[SphereProjOp(100)] * 1000) (n = 100000), REPS times through the segmented engine
(ccqp_segs_project_f64: 2 launches per projection) and REPS times through the round-1 per-block path (2 launches per
block = 2000 per projection), plus the fused step and the residual.  Prints the wall-clock per projection of both.
   python tools/segs_profile_driver.py [blocks] [dim] [reps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from ccqppy_amd import solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    eng = get_engine(0)
    P = ss.DisjointProjOp(*[ss.SphereProjOp(dim)] * blocks)
    n = blocks * dim
    x = torch.randn(n, dtype=torch.float64, device="cuda") * 0.2
    g = torch.randn(n, dtype=torch.float64, device="cuda")
    out, ref = torch.zeros_like(x), torch.zeros_like(x)
    P._dev_project(eng, x, out)
    P._block_project(eng, x, ref, 0, n)
    torch.cuda.synchronize()
    same = bool(torch.equal(out.view(torch.int64), ref.view(torch.int64)))

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    t_segs = timed(lambda: P._dev_project(eng, x, out))
    t_old = timed(lambda: P._block_project(eng, x, ref, 0, n))
    t_step = timed(lambda: P._dev_step_project(eng, 1.0, x, -0.01, g, out))
    t_res = timed(lambda: P._dev_residual_sq(eng, x, g, 1.0 / (3 * n * 1e-6), 1e-6, 5))
    plan = P._segplan(eng, n)
    print(json.dumps({"set": "DisjointProjOp(%d x SphereProjOp(%d))" % (blocks, dim), "n": n, "bitwise_equal": same,
                      "work_items": plan.nwork, "norm_items": plan.nnorm,
                      "segmented_project_us": 1e6 * t_segs, "per_block_project_us": 1e6 * t_old,
                      "segmented_fused_step_us": 1e6 * t_step, "segmented_residual_us": 1e6 * t_res,
                      "speedup_project": t_old / t_segs}))


if __name__ == "__main__":
    main()
