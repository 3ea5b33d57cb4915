#!/usr/bin/env bash
# PMC evidence for the diagonal windows of the CSR matvec: FETCH_SIZE / WRITE_SIZE, L2 hits / misses, VALU / TA load and
# the kernel time of a hashed-circulant case with the diagonal windows on and off (same plan otherwise).
#   bash tools/profile_spmv_diag.sh TAG [n k]      default 4194304 32 (33 entries per row);  16777216 8 = the short-row
#                                                  path (9 entries per row, one thread per row)
set -u
TAG=${1:-r02}
N=${2:-4194304}
K=${3:-32}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in diag nodiag; do
  extra=""; [ "$mode" = nodiag ] && extra="nodiag"
  for group in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "VALUBusy OccupancyPercent MemUnitStalled"; do
    g=$(echo $group | cut -d' ' -f1)
    rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/prof_${TAG}_spmv_${mode}_$g" -- \
      python3 "$ROOT/tools/spmv_profile_driver.py" $N $K 0 $extra > "$OUT/${TAG}_spmv_${mode}_$g.log" 2>&1 || exit 1
  done
done
cd "$ROOT"
python3 - "$TAG" "$N" "$K" <<'PY'
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
n, k = int(sys.argv[2]), int(sys.argv[3])
nnz = n * (k + 1)
out["shape"] = {"n": n, "entries_per_row": k + 1, "nnz": nnz}
out["algorithmic_bytes"] = 12 * nnz + 24 * n
sys.path.insert(0, os.getcwd())
try:
    from bench import box_id
    out["box"] = box_id(0)
except Exception as e:  # noqa: BLE001
    out["box"] = {"error": str(e)}
out["note"] = ("FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (FETCH_SIZE is uncalibrated for 8-byte / 4-byte per-lane "
               "loads, profiles/README.md: compare the two modes with each other); _ms = kernel time under the counters")
json.dump(out, open("gpurun_out/%s_spmv_diag_pmc_n%dk%d.json" % (tag, n, k), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
