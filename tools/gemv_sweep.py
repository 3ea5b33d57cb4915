#!/usr/bin/env python3
"""GEMV tuning sweep on one MI355X: interleaved rounds of kernel variants / shapes in ONE process
(cdna guide rule 24), HIP-event timing on the launch stream.  Prints JSON lines."""
import argparse
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def time_variant(eng, A, x, y, variant, reps):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        eng.gemv(A, x, y, variant=variant)
        b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="131072x131072")
    ap.add_argument("--R", default="2,4,8")
    ap.add_argument("--U", default="1,2,4")
    ap.add_argument("--nt", default="0,1")
    ap.add_argument("--bpc", default="4,8,16")
    ap.add_argument("--variants", default="", help="explicit variant words (hex, comma separated) instead of the R/U/nt/bpc "
                                                  "product; bit 9 (0x200) = the MFMA form")
    ap.add_argument("--pad", type=int, default=0, help="doubles of padding behind every row (lda = cols + pad)")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    for shape in args.shapes.split(","):
        rows, cols = (int(v) for v in shape.split("x"))
        if args.pad:
            A = torch.empty((rows, cols + args.pad), dtype=torch.float64, device="cuda").normal_()[:, :cols]
        else:
            A = torch.empty((rows, cols), dtype=torch.float64, device="cuda")
            eng.gen_spd_rows(A, 0, cols, 1234) if rows <= cols else A.normal_()
        x = torch.randn(cols, dtype=torch.float64, device="cuda")
        y = torch.empty(rows, dtype=torch.float64, device="cuda")
        variants = [0]
        if args.variants:
            variants = [int(v, 0) for v in args.variants.split(",")]
        for R in ([] if args.variants else args.R.split(",")):
            for U in args.U.split(","):
                for nt in args.nt.split(","):
                    for bpc in args.bpc.split(","):
                        variants.append(int(R) | (int(U) << 4) | ((1 - int(nt)) << 8) | (int(bpc) << 12))
        times = {v: [] for v in variants}
        for v in variants:
            time_variant(eng, A, x, y, v, 1)          # warm
        for _ in range(args.rounds):
            for v in variants:
                times[v] += time_variant(eng, A, x, y, v, args.reps)
        nbytes = 8.0 * (rows * cols + cols + rows)
        res = []
        for v in variants:
            t = np.array(times[v])
            res.append(dict(shape=shape, lda=int(A.stride(0)), variant=hex(v), mfma=(v >> 9) & 1, R=v & 0xF, U=(v >> 4) & 0xF, nt=1 - ((v >> 8) & 1),
                            bpc=(v >> 12) & 0xFF, med_ms=float(np.median(t)), min_ms=float(t.min()),
                            gbs_med=nbytes / float(np.median(t)) / 1e6, gbs_best=nbytes / float(t.min()) / 1e6))
        res.sort(key=lambda r: r["med_ms"])
        for r in res:
            print(json.dumps(r), flush=True)
        del A, x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
