#!/usr/bin/env bash
# PMC evidence for the diagonal windows of the CSR matvec: FETCH_SIZE / WRITE_SIZE and the kernel time of the
# hashed-circulant n = 4M, 33 entries per row case with the diagonal windows on and off (same plan otherwise).
#   bash tools/profile_spmv_diag.sh TAG
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in diag nodiag; do
  extra=""; [ "$mode" = nodiag ] && extra="nodiag"
  for group in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    g=$(echo $group | cut -d' ' -f1)
    rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/prof_${TAG}_spmv_${mode}_$g" -- \
      python3 "$ROOT/tools/spmv_profile_driver.py" 4194304 32 0 $extra > "$OUT/${TAG}_spmv_${mode}_$g.log" 2>&1 || exit 1
  done
done
cd "$ROOT"
python3 - "$TAG" <<'PY'
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
out = {}
for mode in ("diag", "nodiag"):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/prof_%s_spmv_%s_*/**/*_counter_collection.csv" % (tag, mode), recursive=True):
        for r in csv.DictReader(open(f)):
            if "spmv_stream_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                agg["_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    e = {k: sum(v) / len(v) for k, v in agg.items()}
    e["launches"] = len(agg["_ms"])
    out[mode] = e
nnz, n = 4194304 * 33, 4194304
out["algorithmic_bytes"] = 12 * nnz + 24 * n
out["note"] = ("FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (FETCH_SIZE is uncalibrated for 8-byte / 4-byte per-lane "
               "loads, profiles/README.md: compare the two modes with each other); _ms = kernel time under the counters")
json.dump(out, open("gpurun_out/%s_spmv_diag_pmc.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
