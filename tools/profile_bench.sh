#!/usr/bin/env bash
# Evidence run of the bench workload on the GPU box (one MI355X): the plain bench line, `rocprofv3 --kernel-trace
# --stats` of the same command, and the PMC passes (one counter group per pass, never combined with tracing domains
# other than --kernel-trace) condensed by tools/summarize_prof.py.  Everything lands under gpurun_out/ (scratch,
# merged back by gpurun); the summaries worth keeping are copied to profiles/ by hand afterwards.
#   bash tools/profile_bench.sh TAG            e.g. TAG = r02
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "[profile] bench (default flags)"; date
python3 "$ROOT/bench.py" > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_default.err" || exit 1
echo "[profile] kernel trace"; date
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_trace" -- \
  python3 "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-n 0 > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_trace.err" || exit 1
i=0
for group in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "VALUBusy MfmaUtil OccupancyPercent MemUnitStalled"; do
  i=$((i + 1))
  echo "[profile] pmc pass $i: $group"; date
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/prof_${TAG}_pmc$i" -- \
    python3 "$ROOT/bench.py" --steps 1 --warmup 0 --cpu-n 0 > "$OUT/${TAG}_pmc$i.json" 2> "$OUT/${TAG}_pmc$i.err" || exit 1
done
# what a WAVE of the matvec kernels does with its cycles (SQ block, 8 slots; quad-cycle units): parked at s_waitcnt
# (WAIT_ANY), issue-stalled (WAIT_INST_ANY), issuing (ACTIVE_INST_ANY) -- the pair kernel against the single one, which
# the untimed reuse-matvec extension of the same run launches.  Not fatal: a counter name this ROCm does not know only
# loses this pass.
i=4
group="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
echo "[profile] pmc pass $i: $group"; date
rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/prof_${TAG}_pmc$i" -- \
  python3 "$ROOT/bench.py" --steps 1 --warmup 0 --cpu-n 0 > "$OUT/${TAG}_pmc$i.json" 2> "$OUT/${TAG}_pmc$i.err" || echo "[profile] pass 4 failed (kept going)"
cd "$ROOT"
# per KERNEL launch: a matvec of this size is several launches (ccqp_gemv_plan); the bench line says how many
export PROF_META=$(python3 - "$OUT/${TAG}_bench_default.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["roofline"]
print(json.dumps({"n": 131072, "rows": r["rows_per_launch"], "launches_per_matvec": r["launches_per_matvec"],
                  "algorithmic_bytes_per_launch": r["bytes_per_launch"],
                  "command": "rocprofv3 --pmc <one group per pass: FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum | "
                             "VALUBusy MfmaUtil OccupancyPercent MemUnitStalled | SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                             "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT> --kernel-trace -- python3 bench.py "
                             "--steps 1 --warmup 0 --cpu-n 0"}))
PY
)
python3 tools/summarize_prof.py "${TAG}_bench_n131072" "$OUT/prof_${TAG}_trace" "$OUT/prof_${TAG}_pmc1" "$OUT/prof_${TAG}_pmc2" "$OUT/prof_${TAG}_pmc3" "$OUT/prof_${TAG}_pmc4" > "$OUT/${TAG}_summary.log" 2>&1 || exit 1
cp profiles/${TAG}_bench_n131072_* "$OUT/" 2>/dev/null
echo "[profile] done"; date
