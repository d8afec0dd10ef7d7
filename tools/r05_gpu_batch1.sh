#!/bin/bash
# round 5, first GPU call: the GPU suite on the refactored boundary, the driver's command (three timed loops now), what body_contacts="auto"
# changes (tools/body_contacts_delta.py), the counters of the launch size where the chip is full (N = 65536, k_step_dense), the switch point
# between the two step kernels (N = 12288), the dense build's phase cycles.      bash tools/r05_gpu_batch1.sh <dir>
OUT=gpurun_out/${1:-r05a}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 2>$OUT/steps20.err | tail -1 > $OUT/steps20_command.json
timeout 600 python bench.py 2>$OUT/headline.err | tail -1 > $OUT/headline_8192.json
python - $OUT <<'P'
import json, sys
for f in ("steps20_command", "headline_8192"):
    d = json.load(open(f"{sys.argv[1]}/{f}.json"))
    print(f, {k: (round(v / 1e6, 2) if k.startswith("value") else v) for k, v in d.items() if k.startswith("value") or k == "ms_per_step"}, d["config"].get("settle_work_ratio"),
          d["roofline"].get("traffic_note"), {k: v for k, v in d["roofline"].items() if k.startswith("kernel_ms")})
P
timeout 900 python tools/body_contacts_delta.py 9000 $OUT/body_contacts_delta.json > $OUT/body_contacts_delta.md 2>$OUT/body_contacts_delta.err; cat $OUT/body_contacts_delta.md
for v in 1 2; do for n in 12288 10240 14336; do
  QS_STEP_VARIANT=$v timeout 300 python bench.py --envs-per-gpu $n --no-cpu-baseline --no-info-line --no-body-contacts-line 2>/dev/null | tail -1 > $OUT/n${n}_variant$v.json
  python -c "import json; d=json.load(open('$OUT/n${n}_variant$v.json')); print('N=$n variant $v:', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'ms')"
done; done
QS_ALLOW_ABI_MISMATCH=1 QS_LIB_PATH=$PWD/tools/bin/prof.so QS_STEP_VARIANT=2 timeout 300 python tools/phase_profile.py 65536 > $OUT/phase_cycles_dense_n65536.txt 2>&1; cat $OUT/phase_cycles_dense_n65536.txt
QS_ALLOW_ABI_MISMATCH=1 QS_LIB_PATH=$PWD/tools/bin/prof.so timeout 300 python tools/phase_profile.py 8192 > $OUT/phase_cycles_n8192.txt 2>&1; tail -25 $OUT/phase_cycles_n8192.txt
QS_PROF_KERNEL=k_step_dense QS_PROF_LAST=300 bash tools/profile_round.sh r05a_n65536 --envs-per-gpu 65536 > $OUT/profile_n65536.log 2>&1; tail -40 $OUT/profile_n65536.log
