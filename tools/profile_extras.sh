#!/usr/bin/env bash
# Evidence runs beside the bench (GPU box, one MI355X): the MFMA matvec under PMC counters, the CSR matvec sweep with
# the diagonal-window A/B, the ensemble harness, the segmented projection under the kernel tracer, lock-step batches on
# non-elementwise sets.  Output under gpurun_out/.      bash tools/profile_extras.sh TAG
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for group in "VALUBusy MfmaUtil OccupancyPercent MemUnitStalled" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i + 1))
  echo "[extras] mfma pmc pass $i: $group"; date
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/prof_${TAG}_mfma_pmc$i" -- \
    python3 "$ROOT/tools/gemv_sweep.py" --shapes 131072x131072 --variants 0x0,0x240 --rounds 1 --reps 2 \
    > "$OUT/${TAG}_mfma_pmc$i.log" 2>&1 || exit 1
done
echo "[extras] segmented projection under the kernel tracer"; date
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_segs" -- \
  python3 "$ROOT/tools/segs_profile_driver.py" 1000 100 20 > "$OUT/${TAG}_segs_driver.json" 2> "$OUT/${TAG}_segs.err" || exit 1
cd "$ROOT"
export PROF_META='{"n": 131072, "rows": 131072, "algorithmic_bytes_per_launch": 137441050624.0, "command": "rocprofv3 --pmc <group> --kernel-trace -- python3 tools/gemv_sweep.py --shapes 131072x131072 --variants 0x0,0x240 --rounds 1 --reps 2"}'
python3 tools/summarize_prof.py "${TAG}_gemv_mfma" "$OUT/prof_${TAG}_mfma_pmc1" "$OUT/prof_${TAG}_mfma_pmc1" "$OUT/prof_${TAG}_mfma_pmc2" "$OUT/prof_${TAG}_mfma_pmc3" > "$OUT/${TAG}_mfma_summary.log" 2>&1
cp profiles/${TAG}_gemv_mfma_pmc.json "$OUT/" 2>/dev/null
find "$OUT/prof_${TAG}_segs" -name "*_kernel_stats.csv" -exec cp {} "$OUT/${TAG}_segs_kernel_stats.csv" \;
echo "[extras] CSR matvec sweep"; date
python3 tools/spmv_sweep.py --ab --shapes 1048576:8,1048576:32,1048576:128,4194304:32,16777216:8 > "$OUT/${TAG}_spmv_sweep.jsonl" 2> "$OUT/${TAG}_spmv.err" || exit 1
python3 tools/spmv_sweep.py --ab --stencil --shapes 2097152:26 >> "$OUT/${TAG}_spmv_sweep.jsonl" 2>> "$OUT/${TAG}_spmv.err" || exit 1
echo "[extras] ensemble"; date
python3 tools/run_ensemble.py "$TAG" > "$OUT/${TAG}_ensemble.log" 2>&1 || exit 1
cp "profiles/${TAG}_ensemble.json" "$OUT/"
echo "[extras] lock-step batches on non-elementwise sets"; date
python3 tools/bench_small_batch.py --set spheres3 63:1024 255:1024 1023:512 2046:256 > "$OUT/${TAG}_bsegs_bench.jsonl" 2> "$OUT/${TAG}_bsegs.err" || exit 1
python3 tools/bench_small_batch.py --set sphere 64:1024 256:1024 1024:512 2048:256 >> "$OUT/${TAG}_bsegs_bench.jsonl" 2>> "$OUT/${TAG}_bsegs.err" || exit 1
echo "[extras] lock-step batch on a union of Sphere(3) blocks under the kernel tracer"; date
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_bsegs" -- \
  python3 "$ROOT/tools/bsegs_profile_driver.py" 1023 512 > "$OUT/${TAG}_bsegs_driver.json" 2> "$OUT/${TAG}_bsegs_prof.err" || exit 1
find "$OUT/prof_${TAG}_bsegs" -name "*_kernel_stats.csv" -exec cp {} "$OUT/${TAG}_bsegs_kernel_stats.csv" \;
cd "$ROOT"
echo "[extras] done"; date
