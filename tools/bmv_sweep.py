#!/usr/bin/env python3
"""The batched matvec of the lock-step forms alone (no solve): B problems of size n, the two full-batch matvec launches
of a lock-step MPRGP start-up (g = A x + b; A g) timed with the C ABI's event hook, for a list of tuning words --
interleaved over several rounds in ONE process.  Prints one JSON line per tuning word (ms and TB/s of matrix bytes).
    python tools/bmv_sweep.py [--problems 128] [--size 8192] [--rounds 4] [--tunes 0,0x70,0x30,...]
tuning word: bits 1-2 loads (0 default / 1 cacheable / 2 non-temporal / 3 the LDS-DMA forms: bits 4-6 then pick the ring,
4-7 = one persistent grid walking the problems in order), bits 4-6 rows x steps per wave (1, 2, 4 = one row,
that many steps; 7 / 6 = two / four rows, one step; 3 / 5 = two rows, two / four steps), bits 8.. workgroups per problem."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools._opts import add_option_args, apply_options  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=128)
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--tunes", default="")
    add_option_args(ap)
    args = ap.parse_args()
    options = apply_options(args)                  # (per-context tuning options: the A/B switches, tools/README.md)
    from ccqppy_amd import lockstep, problems, solvers, solution_spaces as ss
    from ccqppy_amd._engine import get_engine
    from ccqppy_amd.operators import as_operator
    eng = get_engine(0)
    lib = eng.lib
    n, B = args.size, args.problems
    lb, ub = -0.5 * np.ones(n), 0.5 * np.ones(n)
    items = []
    for p in range(B):
        op, b, _ = problems.random_problem(n, 1234 + p, device=0)
        s = solvers.CCQPSolverMPRGP(1e-6, 300)
        items.append((s, as_operator(op), b, None, ss.BoxProjOp(n, lb, ub), True, None))
    spec = lockstep._spec_from_items(eng, items, n)
    if args.tunes:
        tunes = [int(t, 0) for t in args.tunes.split(",")]
    else:
        tunes = [0]
        for u in (7, 3, 5, 6, 1, 2, 4):
            for per in (0, 256, 512, 1024):
                tunes.append((u << 4) | (per << 8))
        tunes += [(1 << 1) | (7 << 4), (1 << 1) | (3 << 4)]
        tunes += [(3 << 1) | (per << 8) for per in (0, 2, 4, 8, 16, 32, 64)]       # the LDS-DMA form
        tunes = list(dict.fromkeys(tunes))
    times = {t: [] for t in tunes}
    buf = (ctypes.c_double * 8)()
    cnt = ctypes.c_int()
    for r in range(args.rounds + 1):
        for t in tunes:
            lib.ccqp_mv_timing_begin(8)
            run = lockstep._MprgpRun(eng, spec, n, np.inf, t, False)
            torch.cuda.synchronize()
            lib.ccqp_mv_timing_end(buf, 8, ctypes.byref(cnt))
            if r:                                        # round 0 warms up
                times[t] += [buf[i] for i in range(cnt.value)]
            del run
    nbytes = B * 8.0 * (n * n + 2 * n)
    best = min(np.median(v) for v in times.values())
    for t in tunes:
        ms = float(np.median(times[t]))
        print(json.dumps({"tune": hex(t), "loads": (t >> 1) & 3, "U": (t >> 4) & 7, "workgroups_per_problem": t >> 8,
                          "median_ms": ms, "min_ms": float(np.min(times[t])), "TBps": nbytes / ms / 1e9,
                          "frac_of_8TBps": nbytes / ms / 1e9 / 8.0, "vs_best": ms / best, "launches": len(times[t]),
                          "n": n, "problems": B}), flush=True)


if __name__ == "__main__":
    main()
