#!/usr/bin/env python3
"""Consistency of the two device-resident execution forms under a saturated memory system: stacked batches of
512 ... 8192 problems, six solvers, lock-step rounds vs the single-launch kernels -- two independent schedules of the
same arithmetic.  Every repetition must give bitwise equal solutions and matvec counts, all problems converged, and
NO problem handed back to the host-driven loop (the fall-back once hid a wave-level race).
   python tools/stress_lockstep.py"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ccqppy_amd import batch, lockstep
from ccqppy_amd._engine import get_engine
eng = get_engine(0); eng.bind()
bad = 0
for n, B in ((128, 8192), (256, 4096), (512, 2048), (1024, 1024), (2048, 512)):
    A = torch.empty((B, n, n), dtype=torch.float64, device="cuda"); xs = torch.empty((B, n), dtype=torch.float64, device="cuda")
    for p in range(B):
        eng.gen_spd_rows(A[p], 0, n, 77 + p); eng.gen_vec(xs[p], 77 + p)
    b = -torch.bmm(A, xs.unsqueeze(2)).squeeze(2)
    lo, up = -0.4 * np.ones(n), 0.3 * np.ones(n)
    for name, tol, cap in (("BBPGD", 1e-8, 5000), ("BBPGDf", 1e-8, 5000), ("PGD", 1e-7, 5000), ("APGD", 1e-8, 5000),
                           ("APGDAntiRelaxation", 1e-8, 5000), ("MPRGP", 1e-6, 3000)):
        for rep in range(3):
            f0 = lockstep.stats["fallback_solves"]
            r1 = batch.solve_stacked(name, A, b, lower=lo, upper=up, tol=tol, max_mv=cap, single_launch=False, step_size=0.3)
            r2 = batch.solve_stacked(name, A, b, lower=lo, upper=up, tol=tol, max_mv=cap, single_launch=True, step_size=0.3)
            fb = lockstep.stats["fallback_solves"] - f0
            same = torch.equal(r1.solution, r2.solution) and np.array_equal(r1.num_matrix_vector_multiplications, r2.num_matrix_vector_multiplications)
            ok = same and fb == 0 and r1.converged.all() and r2.converged.all()
            if not ok:
                bad += 1
            print("n=%d B=%d %s rep %d: same=%s fallbacks=%d converged %d/%d mv max %d %s" % (n, B, name, rep, same, fb, int(r2.converged.sum()), B, int(r2.num_matrix_vector_multiplications.max()), "" if ok else "  <-- BAD"), flush=True)
    del A, b, xs; torch.cuda.empty_cache()
# ---- the persistent LDS-DMA batched matvec (n >= 4096, 1 GiB) against the batched VALU kernels (an explicit rows-x-steps
# field in the tuning word), and for MPRGP the look-ahead (three products per pass) against one matvec per launch: two
# independent schedules of the same arithmetic again
for n, B in ((4096, 96), (8192, 24), (4098, 40)):
    A = torch.empty((B, n, n), dtype=torch.float64, device="cuda"); xs = torch.empty((B, n), dtype=torch.float64, device="cuda")
    for p in range(B):
        eng.gen_spd_rows(A[p], 0, n, 177 + p); eng.gen_vec(xs[p], 177 + p)
    b = -torch.bmm(A, xs.unsqueeze(2)).squeeze(2)
    lo, up = -0.4 * np.ones(n), 0.3 * np.ones(n)
    for name, tol, cap in (("BBPGD", 1e-8, 5000), ("APGD", 1e-8, 5000), ("MPRGP", 1e-6, 3000)):
        for rep in range(3):
            f0 = lockstep.stats["fallback_solves"]
            r1 = batch.solve_stacked(name, A, b, lower=lo, upper=up, tol=tol, max_mv=cap, single_launch=False)
            r2 = batch.solve_stacked(name, A, b, lower=lo, upper=up, tol=tol, max_mv=cap, single_launch=False, tune=3 << 4)
            fb = lockstep.stats["fallback_solves"] - f0
            same = torch.equal(r1.solution, r2.solution) and np.array_equal(r1.num_matrix_vector_multiplications, r2.num_matrix_vector_multiplications) \
                and np.array_equal(r1.executed_matrix_vector_multiplications, r2.executed_matrix_vector_multiplications)
            ok = same and fb == 0 and r1.converged.all() and r2.converged.all()
            if not ok:
                bad += 1
            print("n=%d B=%d %s persistent / look-ahead vs VALU rep %d: same=%s fallbacks=%d converged %d/%d mv max %d %s" % (n, B, name, rep, same, fb, int(r2.converged.sum()), B, int(r2.num_matrix_vector_multiplications.max()), "" if ok else "  <-- BAD"), flush=True)
    del A, b, xs; torch.cuda.empty_cache()
print("BAD", bad)
