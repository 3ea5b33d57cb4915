#!/usr/bin/env python3
"""Kernel durations and the gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (one stream).
    python tools/summarize_trace_gaps.py OUT_DIR [last_fraction]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.34       # the last solve of three
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
    rows.sort()
    rows = rows[int(len(rows) * (1 - frac)):]
    dur, gap = defaultdict(list), defaultdict(list)
    for i, (s, e, k) in enumerate(rows):
        dur[k].append((e - s) / 1e3)
        if i:
            gap[k].append((s - rows[i - 1][1]) / 1e3)
    out = {k: {"calls": len(v), "avg_us": round(sum(v) / len(v), 2), "avg_gap_before_us": round(sum(gap[k]) / max(len(gap[k]), 1), 2)}
           for k, v in dur.items()}
    total = (rows[-1][1] - rows[0][0]) / 1e3
    print(json.dumps({"kernels": out, "span_us": round(total, 1), "busy_us": round(sum(sum(v) for v in dur.values()), 1),
                      "launches": len(rows)}, indent=1))


if __name__ == "__main__":
    main()
