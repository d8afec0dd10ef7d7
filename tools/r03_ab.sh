#!/bin/bash
# A/B of step-kernel builds on ONE box: every library given (paths; "-" = the product library) runs the headline twice, interleaved.
# usage: bash tools/r03_ab.sh out_dir lib1 lib2 ... [-- extra bench args]
OUT=gpurun_out/$1; shift; mkdir -p $OUT
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for rep in 1 2; do
  for lib in "${LIBS[@]}"; do
    name=$(basename $lib .so); [ "$lib" == "-" ] && name=product
    if [ "$lib" == "-" ]; then unset QS_LIB_PATH; else export QS_LIB_PATH=$PWD/$lib; fi
    timeout 600 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-info-line "$@" 2>$OUT/${name}_$rep.err | tail -1 > $OUT/${name}_$rep.json
    python -c "import json; d=json.load(open('$OUT/${name}_$rep.json')); print('$name rep $rep:', round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms kernel', round(d['roofline']['kernel_ms'],4), 'stalls', d['config'].get('stalls'), 'ratio', d['config'].get('settle_work_ratio'))"
  done
done
unset QS_LIB_PATH
