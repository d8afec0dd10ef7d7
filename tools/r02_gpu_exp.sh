#!/bin/bash
OUT=gpurun_out/${1:-r02h}; mkdir -p $OUT
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OUT/$name.json; python -c "import json; d=json.load(open('$OUT/$name.json')); print('$name:', round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms kernel', round(d['roofline']['kernel_ms'],4))"; }
run default
run no_self --env-kw self_collision=False
run static_pool --no-pool-streaming
QS_LIB_PATH=$PWD/quadruped-springs_amd/qs_amd/libqs_hip_r01.so run r01
QS_LIB_PATH=$PWD/quadruped-springs_amd/qs_amd/libqs_hip_r01.so run r01_static --no-pool-streaming
for f in quadruped-springs_amd/qs_amd/variants/*.so; do QS_LIB_PATH=$PWD/$f run $(basename $f .so); done
