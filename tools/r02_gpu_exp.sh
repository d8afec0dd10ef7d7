#!/bin/bash
OUT=gpurun_out/${1:-r02h}; mkdir -p $OUT
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" 2>$OUT/$name.err | tail -1 > $OUT/$name.json; python -c "import json; d=json.load(open('$OUT/$name.json')); print('$name:', round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms kernel', round(d['roofline']['kernel_ms'],4), d['config'].get('settle_work_ratio'))" || tail -3 $OUT/$name.err; }
run default
run default2
run cone0 --solver-residual-threshold 0
run pyramid0 --friction-model pyramid --solver-residual-threshold 0
run config2 --workload config2_4096
run config5 --workload config5_8192
run n65536 --envs-per-gpu 65536 --no-pool-streaming
python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log | head -6
