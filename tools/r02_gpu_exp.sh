#!/bin/bash
OUT=gpurun_out/${1:-r02h}; mkdir -p $OUT
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OUT/$name.json; python -c "import json; d=json.load(open('$OUT/$name.json')); print('$name:', round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms kernel', round(d['roofline']['kernel_ms'],4), 'ratio', d['config']['settle_work_ratio'])"; }
for n in 8192 10240 12288 14336 16384 24576 32768; do run n$n --envs-per-gpu $n; done
