#!/bin/bash
# The round's closing run on one GPU box: the -m gpu suite, the single-solve table of all seven solvers, the default bench
# line -- into OUTDIR (default gpurun_out/final).  bash tools/final_check.sh [OUTDIR]
out=${1:-gpurun_out/final}
mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$out/gpu_suite.txt" 2>&1; echo "pytest rc=$?"; tail -2 "$out/gpu_suite.txt"
timeout -k 10 280 python tools/bench_single_mid.py --solvers PGD,APGD,APGDAntiRelaxation,BBPGD,BBPGDf,SPG,MPRGP \
  --sizes 256,1024,2048,4096,8192,16384 > "$out/single_mid_all.jsonl" 2> "$out/single.err"; echo "single rc=$?"
timeout -k 10 600 python bench.py > "$out/bench_default.json" 2> "$out/bench.err"; echo "bench rc=$?"
python - "$out" <<'PY'
import json, sys
out = sys.argv[1]
d = json.loads(open(out + "/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_ms"], r["traffic"],
      r["traffic_source"]["verified_same_source"] if r["traffic_source"] else None, d["cpu_baseline"]["value"])
for l in open(out + "/single_mid_all.jsonl"):
    d = json.loads(l)
    print(d["n"], d["solver"], d["reported_mv"], d["host_loop_ms"], d["device_resident_ms"], d["speedup"],
          d["same_bits_and_counts"], d["device_resident_fell_back"])
PY
