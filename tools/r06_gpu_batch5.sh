#!/bin/bash
# round 6, fifth GPU call: the GPU suite on the library that stays (hand-over predicted a substep ahead, a third support point for a leg whose
# foot is in the air; no rescue workgroups), and the A/B of the last allocator knob (-greedy-regclass-priority-trumps-globalness=1), three runs each.
OUT=gpurun_out/${1:-r06e}; mkdir -p $OUT
rm -f gpurun_out/impact_parity.jsonl gpurun_out/full_size_oracle_sampled.jsonl gpurun_out/terminal_observation_parity.json
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $OUT/pytest_gpu.log 2>&1; tail -14 $OUT/pytest_gpu.log
cp gpurun_out/impact_parity.jsonl gpurun_out/full_size_oracle_sampled.jsonl gpurun_out/terminal_observation_parity.json $OUT/ 2>/dev/null
for rep in 1 2 3; do for kv in flag=quadruped-springs_amd/qs_amd/libqs_hip.so noflag=tools/bin/r06_noflag.so; do
  name=${kv%%=*}; lib=${kv#*=}
  QS_LIB_PATH=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-info-line 2>/dev/null | tail -1 > $OUT/${name}_$rep.json
  python -c "import json; d=json.load(open('$OUT/${name}_$rep.json')); print('$name', round(d['value']/1e6,2), 'auto', round((d.get('value_body_contacts_auto') or 0)/1e6,2))"
done; done
