#!/usr/bin/env bash
# Round 4's evidence runs beside the headline profile (tools/profile_bench.sh r04) and the config profiles
# (tools/profile_configs.sh r04): the mid-size single solves, the pair kernel's ring shapes, the lock-step batch forms,
# the pair launch probe, the two-rank rehearsal of bench.py, the host split of a large small-problem batch, the
# ensemble harness.  Output under gpurun_out/ (scratch); what is kept is copied into profiles/ by hand.
#   bash tools/evidence_r04.sh TAG
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
echo "[evidence] single solves 64 ... 16384, host loop against the device-resident form"; date
python3 tools/bench_single_mid.py --sizes 256,1024,2048,4096,8192,16384 > "$OUT/${TAG}_single_mid.jsonl" 2> "$OUT/${TAG}_single_mid.err"
python3 tools/bench_single_mid.py --sizes 4096,8192 --solvers APGD --grid 1 > "$OUT/${TAG}_single_mid_grid.jsonl" 2>> "$OUT/${TAG}_single_mid.err"
echo "[evidence] pair kernel ring shapes"; date
python3 tools/gemv2_shapes.py --sizes 131072,32768 --rounds 6 > "$OUT/${TAG}_gemv2_shapes.jsonl" 2> "$OUT/${TAG}_gemv2_shapes.err"
echo "[evidence] pair launch: inside the solve, raw, raw with a sync"; date
python3 tools/pair_launch_probe.py > "$OUT/${TAG}_pair_launch_probe.txt" 2>&1
echo "[evidence] lock-step batch forms"; date
python3 tools/bench_lockstep_matvec_forms.py --shapes 512:4096,1024:512,2048:512,4096:128,8192:64 --solvers APGD,MPRGP > "$OUT/${TAG}_lockstep_matvec_forms.jsonl" 2> "$OUT/${TAG}_lockstep_forms.err"
python3 tools/bench_lockstep_matvec_forms.py --shapes 512:4096 --solvers MPRGP --opt MPRGP_LOOKAHEAD_MIN_N=1024 >> "$OUT/${TAG}_lockstep_matvec_forms.jsonl" 2>> "$OUT/${TAG}_lockstep_forms.err"
echo "[evidence] config 5 share"; date
python3 tools/bench_config5.py > "$OUT/${TAG}_bench_config5.json" 2> "$OUT/${TAG}_bench_config5.err"
echo "[evidence] bench.py with two gloo ranks on one GPU (its own launcher)"; date
CCQP_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 1 --warmup 1 --size 16384 --cpu-n 16384 > "$OUT/${TAG}_bench_gloo2_rehearsal.json" 2> "$OUT/${TAG}_bench_gloo2_rehearsal.err"
echo "[evidence] host split of 8192 problems of n = 256"; date
python3 tools/host_profile_lockstep.py --problems 8192 --size 256 --solver BBPGD > "$OUT/${TAG}_host_profile_8192x256.txt" 2>&1
echo "[evidence] ensemble"; date
python3 tools/run_ensemble.py "$TAG" > "$OUT/${TAG}_ensemble.log" 2>&1
cp "profiles/${TAG}_ensemble.json" "profiles/${TAG}_ensemble_summary.json" "$OUT/" 2>/dev/null
echo "[evidence] done"; date
