#!/bin/bash
# A/B of context options on device-resident single solves: tools/_ab_single.sh OUTDIR "<common args>" label1 "<opts1>" label2 "<opts2>" ...
out=$1; shift; common=$1; shift
mkdir -p "$out"
while [ $# -ge 2 ]; do
  label=$1; opts=$2; shift 2
  timeout -k 10 300 python tools/bench_single_mid.py $common $opts > "$out/$label.jsonl" 2> "$out/$label.err" || { echo "$label failed"; tail -5 "$out/$label.err"; exit 1; }
done
python - "$out" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.jsonl"))):
    for l in open(f):
        d = json.loads(l)
        print(os.path.basename(f)[:-6], d["n"], d["solver"], d["reported_mv"], d["host_loop_ms"], d["device_resident_ms"], d["device_resident_best_ms"], d["speedup"], d["same_bits_and_counts"], d["device_resident_fell_back"])
PY
