#!/bin/bash
OUT=gpurun_out/${1:-r02h}; mkdir -p $OUT
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OUT/$name.json; python -c "import json; d=json.load(open('$OUT/$name.json')); print('$name:', round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms kernel', round(d['roofline']['kernel_ms'],4), 'settle ratio', d['config'].get('settle_work_ratio'))"; }
run pool65536
run pool16384 --reset-pool 16384
python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -s -k "pooled or streaming or ragged" -p no:cacheprovider 2>&1 | grep -i "pooled resets\|passed\|failed"
