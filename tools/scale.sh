#!/bin/bash
# 1 / 2 / 4 / 8-GPU curves of bench.py on ONE node (nothing here has run on hardware yet: no multi-GPU node was available).
#   weak:    8192 environments per GPU (the metric's configuration on every rank)
#   strong:  65536 environments split over the ranks (--total-envs 65536)
#   sharded: BASELINE.json configs[3] -- the learner-side exchange (actions broadcast + one all-gather of [n, o + 2] per step)
# usage: bash tools/scale.sh [out_dir] [gpu counts ...]      default: gpurun_out/scale 1 2 4 8
# Every run appends ONE line to $OUT/scale.jsonl: {"mode": ..., "n_gpus": ..., "line": <the JSON line bench.py printed>}; tools/scale_table.py
# turns the file into the table (value, ms_per_step, efficiency against the 1-GPU line of the same mode, rccl_ranks, per-rank min / max).
OUT=${1:-gpurun_out/scale}; shift
GPUS=("$@"); [ ${#GPUS[@]} -eq 0 ] && GPUS=(1 2 4 8)
mkdir -p "$OUT"; : > "$OUT/scale.jsonl"
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # mode n_gpus bench-args...
  mode=$1; n=$2; shift 2
  line=$(timeout 900 python bench.py --gpus "$n" --steps 1000 --warmup 50 --no-cpu-baseline --no-info-line "$@" 2>"$OUT/${mode}_$n.err" | tail -1)
  if [ -z "$line" ]; then echo "{\"mode\": \"$mode\", \"n_gpus\": $n, \"line\": null, \"error\": \"see ${mode}_$n.err\"}" >> "$OUT/scale.jsonl"
  else echo "{\"mode\": \"$mode\", \"n_gpus\": $n, \"line\": $line}" >> "$OUT/scale.jsonl"; fi
}
for n in "${GPUS[@]}"; do
  run weak "$n"
  run strong "$n" --total-envs 65536
  run sharded "$n" --workload config4_sharded
done
python tools/scale_table.py "$OUT/scale.jsonl"
