#!/usr/bin/env python3
"""The two-vector matvec (ccqp_gemv2_f64: y1 = A x1, y2 = A x2 in one pass over A) under its ring shapes, against two
single matvecs -- interleaved rounds in one process, HIP events on the launch stream.  Prints JSON lines.
    python tools/gemv2_sweep.py [--shapes 131072x131072] [--variants 0x51001800,0x71001800,...]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="131072x131072")
    ap.add_argument("--variants", default="0x51001800,0x71001800,0x11001800,0x53001800,0x51101800")
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    for shape in args.shapes.split(","):
        rows, cols = (int(v) for v in shape.split("x"))
        A = torch.empty((rows, cols), dtype=torch.float64, device="cuda")
        eng.gen_spd_rows(A, 0, cols, 1234)
        x1, x2 = (torch.randn(cols, dtype=torch.float64, device="cuda") for _ in range(2))
        b = torch.randn(rows, dtype=torch.float64, device="cuda")
        y = [torch.empty(rows, dtype=torch.float64, device="cuda") for _ in range(4)]
        variants = [int(v, 0) for v in args.variants.split(",")]

        def timed(fn):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            return a.elapsed_time(e)

        runs = {"two single matvecs (default form)": lambda: (eng.gemv(A, x1, y[0], b, y[1]), eng.gemv(A, x2, y[2], b, y[3]))}
        for v in variants:
            runs["pair, variant %s" % hex(v)] = (lambda v=v: eng.gemv2(A, x1, x2, y[0], y[2], b, y[1], y[3], variant=v))
        if rows == cols:                                    # three products of a square matrix (ccqp_gemv3_f64)
            x3 = torch.randn(cols, dtype=torch.float64, device="cuda")
            y3 = [torch.empty(rows, dtype=torch.float64, device="cuda") for _ in range(2)]
            runs["three single matvecs (default form)"] = lambda: (eng.gemv(A, x1, y[0]), eng.gemv(A, x2, y[2]), eng.gemv(A, x3, y3[0], b, y3[1]))
            runs["triple, default"] = lambda: eng.gemv3(A, [x1, x2, x3], [y[0], y[2], y3[0]], b, [None, None, y3[1]])
            runs["triple, one pass forced"] = lambda: eng.gemv3(A, [x1, x2, x3], [y[0], y[2], y3[0]], b, [None, None, y3[1]], variant=0x11001800)
        runs["pair, default"] = lambda: eng.gemv2(A, x1, x2, y[0], y[2], b, y[1], y[3])
        t = {k: [] for k in runs}
        for k, fn in runs.items():
            timed(fn)
        for _ in range(args.rounds):
            for k, fn in runs.items():
                t[k].append(timed(fn))
        pair_bytes = 8.0 * (rows * cols + 2 * cols + 2 * rows)
        for k in runs:
            m = float(np.median(t[k]))
            print(json.dumps({"shape": shape, "what": k, "ms": round(m, 4), "min_ms": round(min(t[k]), 4),
                              "TBps_of_pair_bytes": round(pair_bytes / m / 1e9, 3),
                              "matvecs_per_s": round(2e3 / m, 2)}), flush=True)
        del A
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
