#!/bin/bash
# round 5, the WARM build on the GPU: the suite, the benchmark (value = body_contacts=True, value_body_contacts_auto next to it) with this
# library and with round 4's on the same box, the rare-path timings.      bash tools/r05_gpu_batch2.sh <dir> [skip-tests]
OUT=gpurun_out/${1:-r05c}; mkdir -p $OUT
show() { python - "$1" <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], {k: (round(v / 1e6, 2) if k.startswith("value") else v) for k, v in d.items() if k.startswith("value") or k == "ms_per_step"}, "ratio", d["config"].get("settle_work_ratio"),
      {k: round(v, 4) for k, v in d["roofline"].items() if k.startswith("kernel_ms")}, "many-rows wave-substeps", d["config"].get("joint_limit_path_wave_substeps"), "stalls", d["config"].get("stalls"))
P
}
timeout 600 python bench.py --no-cpu-baseline --no-info-line 2>$OUT/new.err | tail -1 > $OUT/new_1000.json; show $OUT/new_1000.json
QS_ALLOW_ABI_MISMATCH=1 QS_LIB_PATH=$PWD/tools/bin/r04.so timeout 600 python bench.py --no-cpu-baseline --no-info-line 2>$OUT/r04.err | tail -1 > $OUT/r04_1000.json; show $OUT/r04_1000.json
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-info-line 2>/dev/null | tail -1 > $OUT/new_20_$i.json; show $OUT/new_20_$i.json; done
timeout 600 python tools/time_rare_path.py > $OUT/rare_path.txt 2>&1; grep "ms per step" $OUT/rare_path.txt
if [ -z "$2" ]; then timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log | cut -c1-400; fi
