#!/usr/bin/env python3
"""Ring shapes of the two-vector matvec (ccqp_gemv2_f64 / gemv_lds2_kernel<R, S>): rows per wave x ring slots x waves per
workgroup x workgroups per CU x flush group, interleaved rounds in ONE process on one box, HIP events on the launch
stream around `reps` back-to-back launches, the single LDS-staged matvec of the same matrix beside them.  Every shape is
first checked to leave the bits of the default shape.  The shapes are per-context options (ccqp_ctx_set_option), set
before each timed call.  Prints JSON lines.
    python tools/gemv2_shapes.py [--sizes 131072,32768] [--rounds 6]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# name: (GEMV2_SHAPE, GEMV2_WPB, GEMV2_GROUP, workgroups per CU)
SHAPES = {
    "R4 S4 w4 (default)": (44, 4, 0, 1),
    "R4 S4 w3": (44, 3, 0, 1),
    "R4 S3 w4": (43, 4, 0, 1),
    "R8 S2 w4": (82, 4, 0, 1),
    "R8 S2 w3": (82, 3, 0, 1),
    "R8 S3 w4": (83, 4, 0, 1),
    "R8 S3 w3": (83, 3, 0, 1),
    "R8 S4 w3": (84, 3, 0, 1),
    "R8 S2 w3 x2/CU": (82, 3, 0, 2),
    "R8 S2 w2 x2/CU": (82, 2, 0, 2),
    "R4 S3 w3 x2/CU": (43, 3, 0, 2),
    "R4 S4 w2 x2/CU": (44, 2, 0, 2),
    "R4 S6 w2": (46, 2, 0, 1),
    "R4 S8 w2": (48, 2, 0, 1),
    "R4 S6 w3": (46, 3, 0, 1),
    "R4 S8 w3 g256": (48, 3, 256, 1),
    "R4 S6 w2 x2/CU": (46, 2, 256, 2),
    "R4 S4 w5": (44, 5, 0, 1),
    "R4 S4 w6 g256": (44, 6, 256, 1),
    "R4 S3 w5": (43, 5, 0, 1),
    "R4 S3 w6": (43, 6, 0, 1),
    "R4 S3 w7": (43, 7, 0, 1),
    "R4 S3 w8 g128": (43, 8, 128, 1),
    "R4 S4 w4 g256": (44, 4, 256, 1),
    "R4 S4 w4 g128": (44, 4, 128, 1),
    "R8 S2 w4 g256": (82, 4, 256, 1),
    "R8 S2 w4 g128": (82, 4, 128, 1),
}
AUTO = (1 << 24) | (1 << 28) | (2 << 10) | (1 << 30)        # GEMV_AUTO_LDS of ccqp_gemv.hip without its workgroups field


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="131072,32768")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--only", default="", help="comma-separated substrings of shape names")
    args = ap.parse_args()
    from bench import box_id
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    box = box_id(0)
    names = [k for k in SHAPES if not args.only or any(t in k for t in args.only.split(","))]
    for n in (int(v) for v in args.sizes.split(",")):
        A = torch.empty((n, n), dtype=torch.float64, device="cuda")
        eng.gen_spd_rows(A, 0, n, 1234)
        x1, x2 = (torch.randn(n, dtype=torch.float64, device="cuda") for _ in range(2))
        b = torch.randn(n, dtype=torch.float64, device="cuda")
        y = [torch.full((n,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(4)]
        ref = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]
        reps = max(1, int(round(20.0 / (8.0 * n * n / 7.0e9))))           # ~20 ms of launches per sample

        def apply(shape):
            sh, wpb, grp, bpc = SHAPES[shape]
            eng.set_option("GEMV2_SHAPE", sh)
            eng.set_option("GEMV2_WPB", wpb)
            eng.set_option("GEMV2_GROUP", grp)
            return AUTO | (bpc << 12)

        def pair(shape, out):
            v = apply(shape)
            eng.gemv2(A, x1, x2, out[0], out[2], b, out[1], out[3], variant=v)

        def timed(fn):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            return a.elapsed_time(e) / reps

        eng.gemv(A, x1, ref[0], b, ref[1])                  # the products' own single passes: what every shape must give
        eng.gemv(A, x2, ref[2], b, ref[3])
        same = {}
        for k in names:
            for t in y:
                t.fill_(float("nan"))
            pair(k, y)
            torch.cuda.synchronize()
            same[k] = all(torch.equal(a.view(torch.int64), r.view(torch.int64)) for a, r in zip(y, ref))
        runs = {"single matvec (LDS-staged default)": lambda: eng.gemv(A, x1, y[0], b, y[1])}
        for k in names:
            runs[k] = (lambda k=k: pair(k, y))
        t = {k: [] for k in runs}
        for fn in runs.values():
            timed(fn)
        for _ in range(args.rounds):
            for k, fn in runs.items():
                t[k].append(timed(fn))
        for k in runs:
            m = float(np.median(t[k]))
            single = k.startswith("single")
            nbytes = 8.0 * (n * n + (2 if single else 4) * n)
            print(json.dumps({"n": n, "what": k, "ms": round(m, 5), "min_ms": round(min(t[k]), 5),
                              "max_ms": round(max(t[k]), 5), "frac_of_8TBps": round(nbytes / (m * 1e-3) / 1e9 / 8000.0, 4),
                              "same_bits_as_single_passes": same.get(k), "launches_per_sample": reps,
                              "rounds": args.rounds, "box": box.get("host"), "gpu_uuid": box.get("gpu_uuid")}), flush=True)
        for name in ("GEMV2_SHAPE", "GEMV2_WPB", "GEMV2_GROUP"):
            eng.set_option(name, 0)
        del A
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
