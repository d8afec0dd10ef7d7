#!/bin/bash
# round 2 GPU call: the whole GPU suite (full log), then the headline bench with the new library and with round 1's (A/B)
OUT=gpurun_out/${1:-r02b}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > $OUT/pytest.log
R01=$PWD/quadruped-springs_amd/qs_amd/libqs_hip_r01.so
for cfg in "cone 1e-7" "pyramid 0" "cone 0" "pyramid 1e-7"; do
  set -- $cfg
  timeout 300 python bench.py --no-cpu-baseline --friction-model $1 --solver-residual-threshold $2 2>/dev/null | tail -1 > $OUT/bench_new_$1_$2.json
  QS_LIB_PATH=$R01 timeout 300 python bench.py --no-cpu-baseline --friction-model $1 --solver-residual-threshold $2 2>/dev/null | tail -1 > $OUT/bench_r01_$1_$2.json
done
tail -25 $OUT/pytest.log
for f in $OUT/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms', d['roofline']['kernel_ms'], d['config']['joint_limit_path_wave_substeps'], d['config'].get('self_collision_narrow_phase_wave_substeps'))"; done
