#!/bin/bash
# A/B on one box: a variant library of the many-rows solve (tools/bin/$1.so) against the tree's: parity on the rare paths first, then the timings
V=${1:?variant}; OUT=gpurun_out/ab_$V; mkdir -p $OUT
export QS_LIB_PATH=$PWD/tools/bin/$V.so
timeout 900 python -m pytest tests -m gpu -q -x -k "wave_mates or body_contacts or fallen or resynced or hand_over or soft_payload or limit" 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 600 python tools/fuzz_parity.py 200 95 fallen 2>&1 | tail -1
unset QS_LIB_PATH
B="python bench.py --no-cpu-baseline --no-info-line --no-body-contacts-line"
for rep in 1 2; do for lib in base $V; do
  if [ $lib = base ]; then unset QS_LIB_PATH; else export QS_LIB_PATH=$PWD/tools/bin/$lib.so; fi
  $B 2>/dev/null | tail -1 > $OUT/${lib}_$rep.json; python -c "
import json; d=json.load(open('$OUT/${lib}_$rep.json')); print('$lib rep $rep: True', round(d['value']/1e6,2), 'M; kernel', round(d['roofline']['kernel_ms'],4))"
done; done
for lib in base $V; do if [ $lib = base ]; then unset QS_LIB_PATH; else export QS_LIB_PATH=$PWD/tools/bin/$lib.so; fi; echo $lib; python tools/time_rare_path.py 2>&1 | grep "ms per step"; done
