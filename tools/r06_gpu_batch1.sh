#!/bin/bash
# round 6, first GPU call: the GPU suite with the new driver-run gates (impact rows under the oracle's own yardstick, terminal observations of
# >= 2000 falls at N = 8192, the fallen / lookahead fuzz with both step kernels) and their durations; the driver's command and the headline.
#     bash tools/r06_gpu_batch1.sh <dir>
OUT=gpurun_out/${1:-r06a}; mkdir -p $OUT
rm -f gpurun_out/impact_parity.jsonl gpurun_out/full_size_oracle_sampled.jsonl gpurun_out/terminal_observation_parity.json
timeout 2400 python -m pytest tests -m gpu -q --durations=30 > $OUT/pytest_gpu.log 2>&1; tail -60 $OUT/pytest_gpu.log
cp gpurun_out/impact_parity.jsonl gpurun_out/full_size_oracle_sampled.jsonl gpurun_out/terminal_observation_parity.json $OUT/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 2>$OUT/steps20.err | tail -1 > $OUT/steps20_command.json
timeout 600 python bench.py --no-cpu-baseline 2>$OUT/headline.err | tail -1 > $OUT/headline_8192.json
python - $OUT <<'P'
import json, sys
for f in ("steps20_command", "headline_8192"):
    d = json.load(open(f"{sys.argv[1]}/{f}.json"))
    print(f, {k: (round(v / 1e6, 2) if k.startswith("value") else v) for k, v in d.items() if k.startswith("value") or k == "ms_per_step"}, d["config"].get("settle_work_ratio"),
          d["roofline"].get("traffic_note"), {k: v for k, v in d["roofline"].items() if k.startswith("kernel_ms")})
P
