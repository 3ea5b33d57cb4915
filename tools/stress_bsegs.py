#!/usr/bin/env python3
"""Consistency of the segmented lock-step forms (ccqp_bsegs_*) under a saturated memory system: batches of 512 ... 4096
problems on Sphere and Sphere(3)-union sets, seven solvers (SPG: ccqp_bspg.hip, one generator per problem; MPRGP:
ccqp_bsegs_mprgp_*), three
repetitions each.  Every repetition must give the
same bits and matvec counts as the first one (the schedule of the workgroups differs from run to run: a race between
a deciding block and a late reader would show as a difference), every problem must converge,
and a sample of problems is compared with its own host-driven solve().  NO problem may fall back to the host loop.
   python tools/stress_bsegs.py"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ccqppy_amd import batch, lockstep, problems, solvers, solution_spaces as ss  # noqa: E402


def main():
    bad = 0
    for n, B in ((129, 4096), (258, 2048), (513, 1024), (1023, 512)):
        probs = {}
        for fam in ("sphere", "spheres3"):
            if fam == "sphere":
                P = ss.SphereProjOp(n, 2.0)
            else:
                P = ss.DisjointProjOp(*([ss.SphereProjOp(3, 0.3)] * (n // 3) + ([ss.IdentityProjOp(n % 3)] if n % 3 else [])))
            ps = []
            for p in range(B):
                op, b, _ = problems.random_problem(n, 900 + p)
                ps.append((op, b, None, P))
            probs[fam] = ps
        for fam, ps in probs.items():
            for name, args in (("BBPGD", (1e-8, 5000)), ("BBPGDf", (1e-8, 5000)), ("PGD", (1e-7, 5000, 0.3)),
                               ("APGD", (1e-8, 5000)), ("APGDAntiRelaxation", (1e-8, 5000)), ("SPG", (1e-8, 5000)),
                               ("MPRGP", (1e-6, 3000))):
                base = getattr(solvers, "CCQPSolver" + name)
                seeds = iter(())

                def cls(*a, base=base, name=name):
                    s = base(*a)
                    if name == "SPG":                    # (ccqp_bspg.hip: the k-th solver of a run draws from stream k)
                        s.rng = np.random.RandomState(next(seeds))
                    return s
                first = None
                for rep in range(3):
                    f0 = dict(lockstep.stats)
                    seeds = iter(range(B))
                    res = batch.solve_batch_pipelined(lambda: cls(*args), ps)
                    fb = lockstep.stats["fallback_solves"] - f0["fallback_solves"]
                    x = torch.stack([r.solution for r in res])
                    mv = np.array([r.solution_num_matrix_vector_multiplications for r in res])
                    conv = all(r.solution_converged for r in res)
                    if first is None:
                        first = (x, mv)
                        same = True
                        # a sample against the host-driven loop
                        for p in range(0, B, max(B // 8, 1)):
                            seeds = iter([p])
                            with contextlib.redirect_stdout(io.StringIO()):
                                one = cls(*args).solve(*ps[p][:2], convex_proj_op=ps[p][3])
                            same = same and torch.equal(one.solution, x[p]) and \
                                one.solution_num_matrix_vector_multiplications == mv[p]
                    else:
                        same = torch.equal(first[0], x) and np.array_equal(first[1], mv)
                    ok = same and fb == 0 and conv
                    bad += not ok
                    print("n=%d B=%d %-8s %-18s rep %d: same=%s fallbacks=%d converged=%s mv max %d%s" % (
                        n, B, fam, name, rep, same, fb, conv, int(mv.max()), "" if ok else "  <-- BAD"), flush=True)
        del probs
        torch.cuda.empty_cache()
    print("BAD", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
