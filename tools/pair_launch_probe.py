#!/usr/bin/env python3
"""The pair launch of the headline solve three ways on one box: inside the solve (HIP events of bench.py), raw back-to-back
launches, raw launches with a host sync between them -- do the events of bench.py measure the kernel or the rhythm?
    python tools/pair_launch_probe.py"""
import contextlib, io, json, sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccqppy_amd import problems, solvers, solution_spaces as ss
from ccqppy_amd._engine import get_engine
eng = get_engine(0)
n = 131072
op, b, _ = problems.random_problem(n, 1234, device=0)
P = ss.BoxProjOp(n, -0.5, 0.5)
def step():
    with contextlib.redirect_stdout(io.StringIO()):
        return solvers.CCQPSolverAPGD(1e-8, 5000).solve(op, b, convex_proj_op=P)
step()
for rep in range(3):
    eng.gemv_events = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms = [a.elapsed_time(c) for a, c in eng.gemv_events]
    eng.gemv_events = None
    print("solve %.1f ms; launches %d; min %.3f median %.3f mean %.3f max %.3f; first five %s" % (1e3*dt, len(ms), min(ms), np.median(ms), np.mean(ms), max(ms), [round(v,2) for v in ms[:5]]))
# raw back-to-back launches of the same call, same engine
x1 = torch.randn(n, dtype=torch.float64, device="cuda"); x2 = torch.randn(n, dtype=torch.float64, device="cuda")
y = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]
for rep in range(3):
    ev = []
    for _ in range(8):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.gemv2(op.A, x1, x2, y[0], y[2], b, y[1], y[3]); c.record(); ev.append((a, c))
    torch.cuda.synchronize()
    ms = [a.elapsed_time(c) for a, c in ev]
    print("raw pair launches: min %.3f median %.3f max %.3f" % (min(ms), np.median(ms), max(ms)))
# raw launches with a host sync + small kernels between them (the solve's rhythm)
for rep in range(2):
    ev = []
    for _ in range(8):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.gemv2(op.A, x1, x2, y[0], y[2], b, y[1], y[3]); c.record(); ev.append((a, c))
        eng.dots([(y[0], y[2])], 0); eng.read(0, 1)
    torch.cuda.synchronize()
    ms = [a.elapsed_time(c) for a, c in ev]
    print("pair launches with a sync between: min %.3f median %.3f max %.3f" % (min(ms), np.median(ms), max(ms)))
