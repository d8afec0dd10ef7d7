#!/bin/bash
# how the 64-batch action ring shapes what a short timed region sees: the benchmark with rings of 64 and 1024 batches, 20- and 1000-step regions,
# and 20-step regions at other places of the ring (--preroll moves the region)
OUT=gpurun_out/${1:-r05e}; mkdir -p $OUT
run() { name=$1; shift; python bench.py --no-cpu-baseline --no-info-line "$@" 2>/dev/null | tail -1 > $OUT/$name.json; python -c "
import json; d=json.load(open('$OUT/$name.json')); c=d['config']; print('$name: True', round(d['value_body_contacts_true']/1e6,2), 'auto', round(d['value_body_contacts_auto']/1e6,2), 'M; kernel_ms', round(d['roofline']['kernel_ms_body_contacts_true'],4), round(d['roofline']['kernel_ms_body_contacts_auto'],4), '; resets/step', round(c['resets_in_timed_region']/d['steps'],2), 'rare wave-substeps/step', round(c['joint_limit_path_wave_substeps']/d['steps'],2), 'ratio', c['settle_work_ratio'])"; }
run ring64_1000
run ring1024_1000 --action-ring 1024
for p in 2048 2064 2080 2096; do run ring64_20_preroll$p --steps 20 --warmup 5 --preroll $p; done
for p in 2048 2064 2080 2096; do run ring1024_20_preroll$p --steps 20 --warmup 5 --preroll $p --action-ring 1024; done
