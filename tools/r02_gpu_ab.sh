#!/bin/bash
# A/B of libraries built from earlier commits (quadruped-springs_amd/qs_amd/exp/lib_<commit>.so) at N = 65536 and the headline
OUT=gpurun_out/${1:-r02ab}; mkdir -p $OUT
for lib in "" $(ls quadruped-springs_amd/qs_amd/exp/lib_*.so 2>/dev/null); do
    name=$(basename "${lib:-HEAD}" .so)
    for cfgname in n65536 headline; do
        if [ $cfgname = n65536 ]; then args="--envs-per-gpu 65536 --no-pool-streaming"; else args=""; fi
        QS_LIB_PATH=${lib:+$PWD/$lib} timeout 300 python bench.py --no-cpu-baseline $args 2>$OUT/$name.$cfgname.err | tail -1 > $OUT/$name.$cfgname.json
        python -c "import json; d=json.load(open('$OUT/$name.$cfgname.json')); print('$name $cfgname:', round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms kernel', round(d['roofline']['kernel_ms'],4))" 2>/dev/null || tail -2 $OUT/$name.$cfgname.err
    done
done
python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
