#!/bin/bash
# rocprofv3 passes behind profiles/r01_*: kernel trace + stats, then one PMC pass per counter group (never combined with
# sys/hip/hsa traces).  Run on the GPU box:  bash tools/profile_round.sh <tag> [bench.py flags]
set -u
TAG=${1:-d}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 200 --warmup 20 --no-cpu-baseline $*"
python3 $REPO/bench.py $ARGS > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py $ARGS > "$OUT/trace.log" 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -- python3 $REPO/bench.py $ARGS > "$OUT/pmc$i.log" 2>&1
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT" k_step > "$OUT/summary.md" 2>&1
# keep what travels back small: drop the per-launch traces, keep stats + counter files trimmed to the step kernel
find "$OUT" -name "*_kernel_trace.csv" -delete
for f in $(find "$OUT" -name "*_counter_collection.csv"); do head -1 "$f" > "$f.k"; grep k_step "$f" | tail -400 >> "$f.k"; mv "$f.k" "$f"; done
du -sh "$OUT"; cat "$OUT/summary.md"; cat "$OUT/bench_unprofiled.json" | cut -c1-400
