#!/bin/bash
# A/B of compile-time variants of the step library on the headline bench (which part of round 2's additions costs the common path what)
OUT=gpurun_out/${1:-r02c}; mkdir -p $OUT
V=$PWD/quadruped-springs_amd/qs_amd
for lib in $V/libqs_hip.so $V/variants/*.so $V/libqs_hip_r01.so; do
  name=$(basename $lib .so)
  for cfg in "cone 1e-7" "pyramid 0"; do
    set -- $cfg
    QS_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --friction-model $1 --solver-residual-threshold $2 2>/dev/null | tail -1 > $OUT/bench_${name}_$1_$2.json
    python -c "import json; d=json.load(open('$OUT/bench_${name}_$1_$2.json')); print('$name $1 $2:', round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms kernel', round(d['roofline']['kernel_ms'],4))"
  done
done
