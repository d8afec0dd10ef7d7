#!/bin/bash
# round 6: where the dense kernel's loss comes from -- round 5's source (base), + predicted hand-over + third support point with the allocator knob
# (vC) and without it (final), at the launch sizes where k_step_dense runs.
OUT=gpurun_out/${1:-r06i}; mkdir -p $OUT
for n in 16384 65536; do for rep in 1 2; do for kv in base=tools/bin/r06_base.so vC=tools/bin/r06_vC.so final=quadruped-springs_amd/qs_amd/libqs_hip.so; do
  name=${kv%%=*}; lib=${kv#*=}
  QS_LIB_PATH=$PWD/$lib timeout 300 python bench.py --envs-per-gpu $n --no-cpu-baseline --no-info-line 2>/dev/null | tail -1 > $OUT/${name}_n${n}_$rep.json
  python -c "import json; d=json.load(open('$OUT/${name}_n${n}_$rep.json')); print('N=$n $name', round(d['value']/1e6,2), 'auto', round((d.get('value_body_contacts_auto') or 0)/1e6,2))"
done; done; done
