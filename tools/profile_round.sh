#!/bin/bash
# rocprofv3 passes behind profiles/rNN_<tag>_*: one unprofiled bench line, kernel trace + stats, then one PMC pass per counter group
# (never combined with sys / hip / hsa traces).  The counter averages and the pmc.json cover the LAST 1000 launches of the step kernel = the timed
# region of the profiled command (the launches before it are preparation: staggered resets, pre-roll, warmup); the kernel stats CSV is
# rocprofv3's own, over all launches.  The raw per-launch CSVs are deleted afterwards.
#   bash tools/profile_round.sh <tag> [bench.py flags]     on the GPU box; results in gpurun_out/prof_<tag>/
set -u
TAG=${1:-d}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps ${QS_PROF_LAST:-1000} --warmup 50 --no-cpu-baseline --no-info-line --no-body-contacts-line $*"
python3 $REPO/bench.py $ARGS 2> "$OUT/bench.err" | tail -1 > "$OUT/bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py $ARGS > "$OUT/trace.log" 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -- python3 $REPO/bench.py $ARGS > "$OUT/pmc$i.log" 2>&1
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT" ${QS_PROF_KERNEL:-k_step} "$OUT/bench.json" "$OUT/pmc.json" --last ${QS_PROF_LAST:-1000} > "$OUT/summary.md" 2>&1
{ echo; echo "(rocprofv3's VGPR / AGPR / scratch columns above come from its dispatch records: it reports the architected VGPR half only.  The code object itself -- tools/kernel_resources.py on the library that ran:)"; echo '```'; python3 tools/kernel_resources.py ${QS_PROF_KERNEL:-k_step}; echo '```'; } >> "$OUT/summary.md" 2>&1
cp $(find "$OUT/trace" -name "*_kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/trace" "$OUT"/pmc[0-9] "$OUT"/*.log
du -sh "$OUT"; cat "$OUT/summary.md"; cut -c1-300 "$OUT/bench.json"
