#!/usr/bin/env bash
# Evidence run of BASELINE configs 2, 3 and 5 (1-GPU share) with the discipline of the headline (tools/profile_bench.sh):
# the plain line with its live roofline object, `rocprofv3 --kernel-trace --stats` of the same program, and the three
# PMC passes (one counter group per pass, --kernel-trace only), condensed by tools/summarize_prof.py into
# profiles/<TAG>_config<k>_{kernel_stats.csv,pmc.json}.  The program goes directly behind `--`.
#   bash tools/profile_configs.sh TAG [configs, default "2 3 5"]
set -u
TAG=${1:-r03}
WHICH=${2:-"2 3 5"}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in $WHICH; do
  echo "[configs] config $c: sizes of the launch (a first plain run; the line that is kept is taken LAST, below)"; date
  python3 "$ROOT/tools/config_profile_driver.py" $c > "$OUT/${TAG}_config${c}_meta.json" 2> "$OUT/${TAG}_config${c}_meta.err" || exit 1
  echo "[configs] config $c: kernel trace"; date
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_c${c}_trace" -- \
    python3 "$ROOT/tools/config_profile_driver.py" $c --reps 2 > "$OUT/${TAG}_config${c}_under_rocprof.json" 2> "$OUT/${TAG}_c${c}_trace.err" || exit 1
  i=0
  for group in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "VALUBusy MfmaUtil OccupancyPercent MemUnitStalled"; do
    i=$((i + 1))
    echo "[configs] config $c: pmc pass $i: $group"; date
    rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/prof_${TAG}_c${c}_pmc$i" -- \
      python3 "$ROOT/tools/config_profile_driver.py" $c --reps 1 --warm 0 > "$OUT/${TAG}_c${c}_pmc$i.json" 2> "$OUT/${TAG}_c${c}_pmc$i.err" || exit 1
  done
  cd "$ROOT"
  if [ "$c" = "5" ]; then
    export PROF_FULL_ONLY=1 PROF_KERNEL_FILTER="bp_gemv" PROF_SOURCES="ccqppy_amd/csrc/ccqp_bmprgp.hip,ccqppy_amd/csrc/ccqp_lockstep.h,ccqppy_amd/csrc/ccqp_bodies.h,ccqppy_amd/csrc/ccqp_internal.h"
  else
    unset PROF_FULL_ONLY; export PROF_KERNEL_FILTER="gemv_" PROF_SOURCES=""
  fi
  export PROF_META=$(python3 - "$OUT/${TAG}_config${c}_meta.json" "$c" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
bpl = r["full_launch"]["bytes_per_launch"] if sys.argv[2] == "5" else r["bytes_per_launch"]
print(json.dumps({"config": d["config"], "algorithmic_bytes_per_launch": bpl,
                  "note": "config 5: means over the launches that carry the whole batch (duration >= 90 % of the longest)"
                          if sys.argv[2] == "5" else None,
                  "command": "rocprofv3 --pmc <one group per pass: FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum | VALUBusy "
                             "MfmaUtil OccupancyPercent MemUnitStalled> --kernel-trace -- python3 tools/config_profile_driver.py "
                             + sys.argv[2] + " --reps 1 --warm 0"}))
PY
)
  python3 tools/summarize_prof.py "${TAG}_config${c}" "$OUT/prof_${TAG}_c${c}_trace" "$OUT/prof_${TAG}_c${c}_pmc1" "$OUT/prof_${TAG}_c${c}_pmc2" "$OUT/prof_${TAG}_c${c}_pmc3" > "$OUT/${TAG}_config${c}_summary.log" 2>&1 || exit 1
  # (summarize_prof.py writes into profiles/: only ITS files travel back -- the line files of this run are already in $OUT)
  cp profiles/${TAG}_config${c}_kernel_stats* profiles/${TAG}_config${c}_pmc.json "$OUT/" 2>/dev/null
  cd /tmp
  # the plain line last: its "traffic" is read from the PMC summary just written -- same sources (sha1), same box, same call
  echo "[configs] config $c: plain line"; date
  python3 "$ROOT/tools/config_profile_driver.py" $c --cpu > "$OUT/${TAG}_config${c}_line.json" 2> "$OUT/${TAG}_config${c}_line.err" || exit 1
done
echo "[configs] done"; date
