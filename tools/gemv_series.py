#!/usr/bin/env python3
"""The throughput-only series of SURVEY 8(d): 100 back-to-back matvecs, warm, default variant, at every size of the
BASELINE configs and at the row slabs one rank holds at 2 / 4 / 8 GPUs (the 1-GPU stand-in for the per-rank matvec of a
sharded run: same shape, same kernel, no exchange).  One pair of HIP events around the 100 launches (back-to-back rate,
launch gaps included) and, in a second pass, one pair per launch (kernel-level median).  Prints JSON lines.

    python tools/gemv_series.py [--reps 100]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=100)
    args = ap.parse_args()
    from bench import box_id, HBM_PEAK_GBS
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    shapes = [(n, n, "square n = %d" % n) for n in (1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072)]
    shapes += [(131072 // g, 131072, "rank slab of n = 131072 at %d GPUs" % g) for g in (2, 4, 8)]
    shapes += [(262144 // g, 262144, "rank slab of n = 262144 (config 4) at %d GPUs" % g) for g in (4, 8)]
    box = box_id(0)
    for rows, cols, what in shapes:
        A = torch.empty((rows, cols), dtype=torch.float64, device="cuda")
        if rows == cols:
            eng.gen_spd_rows(A, 0, cols, 1234)
        else:
            for r0 in range(0, rows, 8192):                  # (row-addressable generator: a slab of the big matrix)
                eng.gen_spd_rows(A[r0:r0 + 8192], r0, cols, 1234)
        x = torch.randn(cols, dtype=torch.float64, device="cuda")
        y = torch.empty(rows, dtype=torch.float64, device="cuda")
        for _ in range(3):
            eng.gemv(A, x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            eng.gemv(A, x, y)
        e1.record()
        torch.cuda.synchronize()
        ms_b2b = e0.elapsed_time(e1) / args.reps
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for a, b in ev:
            a.record()
            eng.gemv(A, x, y)
            b.record()
        torch.cuda.synchronize()
        ms_k = float(np.median([a.elapsed_time(b) for a, b in ev]))
        nbytes = 8.0 * (rows * cols + cols + rows)
        print(json.dumps({"shape": [rows, cols], "what": what, "kernel": eng.gemv_form(rows, cols),
                          "launches_per_matvec": eng.gemv_plan(rows, cols)[1], "matvecs": args.reps,
                          "ms_per_matvec_back_to_back": ms_b2b, "matvecs_per_s": 1e3 / ms_b2b,
                          "GBps_back_to_back": nbytes / ms_b2b / 1e6, "frac_back_to_back": nbytes / ms_b2b / 1e6 / HBM_PEAK_GBS,
                          "ms_per_matvec_kernel_median": ms_k, "frac_kernel_median": nbytes / ms_k / 1e6 / HBM_PEAK_GBS,
                          "bytes_per_matvec": nbytes, "box": box}), flush=True)
        del A, x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
