#!/usr/bin/env python3
"""Sweep of the CSR matvec (ccqp_spmv_csr_f64): subgroup width x matrix shape, HIP-event timed.
Algorithmic bytes per call = 12*nnz + 16*rows + 8*cols (values + columns, row pointers + y, x once)."""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="1048576:8,1048576:32,1048576:128,4194304:32,16777216:8")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--scramble", action="store_true", help="also time a column-scrambled copy (random gathers)")
    ap.add_argument("--stencil", action="store_true", help="with k = 26: also a 27-point-stencil pattern (column reuse)")
    ap.add_argument("--ab", action="store_true", help="also time the generic paths (diagonal windows off), width auto: "
                                                      "key w0_generic")
    args = ap.parse_args()
    from ccqppy_amd import problems
    from ccqppy_amd.operators import SparseOperator
    from ccqppy_amd._engine import get_engine
    eng = get_engine(0)
    eng.bind()
    for shape in args.shapes.split(","):
        n, k = (int(v) for v in shape.split(":"))
        rowptr, col, val = problems.sparse_spd_csr(n, k, 1234)
        variants = [("circulant", col)]
        if args.stencil:
            # banded pattern with column reuse inside a window (27-point-stencil-like: offsets {-1,0,1} x 9 planes)
            side = max(4, int(round(n ** (1.0 / 3.0))))
            offs = sorted({dz * side * side + dy * side + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)})
            if len(offs) == k + 1:
                i = torch.arange(n, device=col.device, dtype=torch.int64)[:, None]
                d = torch.tensor(offs, device=col.device, dtype=torch.int64)[None, :]
                variants.append(("stencil27", torch.sort(torch.remainder(i + d, n), dim=1).values.reshape(-1).to(torch.int32)))
        if args.scramble:
            perm = torch.randperm(n, device=col.device, dtype=torch.int64)
            variants.append(("scrambled", perm[col.long()].to(torch.int32)))
        x = problems.xstar(n, 1, eng.device)
        y = eng.empty(n)
        nnz = val.numel()
        nbytes = 12 * nnz + 16 * n + 8 * n
        for pname, c in variants:
            row = {"n": n, "k": k, "nnz": nnz, "pattern": pname, "GB": nbytes / 1e9}
            todo = [(w, True) for w in (0, 4, 8, 16, 32, 64)] + ([(0, False)] if args.ab else [])
            for width, diag in todo:
                op = SparseOperator((rowptr, c, val), shape=(n, n), width=width, check=False, diagonal_windows=diag)
                row.setdefault("plan_summary", [int(v) for v in op.summary])
                for _ in range(3):
                    op.matvec(x, y)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    op.matvec(x, y)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.reps
                row["w%d" % width if diag else "w0_generic"] = [round(ms, 4), round(nbytes / ms / 1e6)]
            print(json.dumps(row), flush=True)
        del rowptr, col, val


if __name__ == "__main__":
    main()
