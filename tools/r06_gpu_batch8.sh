#!/bin/bash
# round 6: the votes evaluated early (joint stops from q at the substep's top, support points with the start-of-substep velocities right behind
# the kinematics) against the library before them, same box.   bash tools/r06_gpu_batch8.sh <dir>
OUT=gpurun_out/${1:-r06h}; mkdir -p $OUT
for rep in 1 2 3; do for kv in before=tools/bin/r06_final0.so early=quadruped-springs_amd/qs_amd/libqs_hip.so; do
  name=${kv%%=*}; lib=${kv#*=}
  QS_LIB_PATH=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-info-line 2>/dev/null | tail -1 > $OUT/${name}_$rep.json
  QS_LIB_PATH=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-info-line 2>/dev/null | tail -1 > $OUT/${name}_steps20_$rep.json
  python -c "
import json
d=json.load(open('$OUT/${name}_$rep.json')); e=json.load(open('$OUT/${name}_steps20_$rep.json'))
print('$name', round(d['value']/1e6,2), 'auto', round((d.get('value_body_contacts_auto') or 0)/1e6,2), '| steps20', round(e['value']/1e6,2), 'auto', round((e.get('value_body_contacts_auto') or 0)/1e6,2), '| many-rows wave-substeps', d['config'].get('joint_limit_path_wave_substeps'))"
done; done
for kv in before=tools/bin/r06_final0.so early=quadruped-springs_amd/qs_amd/libqs_hip.so; do name=${kv%%=*}; lib=${kv#*=}; echo "== $name"; QS_LIB_PATH=$PWD/$lib timeout 300 python tools/time_rare_path.py 2>&1 | grep "ms per step"; done
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -x -k "wave_mates or fuzz or bitwise or resynced or fallen or support or body_contacts or native" > $OUT/pytest_focus.log 2>&1; tail -4 $OUT/pytest_focus.log
