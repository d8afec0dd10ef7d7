#!/bin/bash
# round 6, second GPU call: the hand-over predicted one substep ahead (qs_core.h substep() -> 2) against the library before it, same box:
# the GPU suite on the new library, then headline / 20-step command / rare-path scenarios of both, interleaved.   bash tools/r06_gpu_batch2.sh <dir>
OUT=gpurun_out/${1:-r06b}; mkdir -p $OUT
rm -f gpurun_out/impact_parity.jsonl gpurun_out/full_size_oracle_sampled.jsonl gpurun_out/terminal_observation_parity.json
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest_gpu.log 2>&1; tail -25 $OUT/pytest_gpu.log
cp gpurun_out/impact_parity.jsonl gpurun_out/full_size_oracle_sampled.jsonl gpurun_out/terminal_observation_parity.json $OUT/ 2>/dev/null
NEW=quadruped-springs_amd/qs_amd/libqs_hip.so
bash tools/ab_libs.sh $OUT/ab base=tools/bin/r06_base.so new=$NEW
for rep in 1 2; do for kv in base=tools/bin/r06_base.so new=$NEW; do
  name=${kv%%=*}; lib=${kv#*=}
  QS_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-info-line 2>/dev/null | tail -1 > $OUT/ab/${name}_steps20_$rep.json
  python -c "import json; d=json.load(open('$OUT/ab/${name}_steps20_$rep.json')); print('$name steps20', round(d['value']/1e6,2), 'auto', round((d.get('value_body_contacts_auto') or 0)/1e6,2))"
done; done
for kv in base=tools/bin/r06_base.so new=$NEW; do name=${kv%%=*}; lib=${kv#*=}; echo "== $name"; QS_LIB_PATH=$PWD/$lib python tools/time_rare_path.py 2>&1 | grep "ms per step"; done
