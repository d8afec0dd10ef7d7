# A/B on one box: the support points' rule in the hot build's vote (tools/bin/vote.so) against the tree's library
OUT=gpurun_out/r05aa; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-info-line"
for rep in 1 2; do for lib in base vote; do
  if [ $lib = base ]; then unset QS_LIB_PATH; else export QS_LIB_PATH=$PWD/tools/bin/$lib.so; fi
  $B 2>/dev/null | tail -1 > $OUT/${lib}_$rep.json; python -c "
import json; d=json.load(open('$OUT/${lib}_$rep.json')); print('$lib rep $rep: True', round(d['value']/1e6,2), 'auto', round(d['value_body_contacts_auto']/1e6,2), 'M; kernel', round(d['roofline']['kernel_ms'],4), 'many-rows wave-substeps', d['config']['joint_limit_path_wave_substeps'])"
done; done
export QS_LIB_PATH=$PWD/tools/bin/vote.so
timeout 900 python -m pytest tests -m gpu -q -x -k "wave_mates or body_contacts or fallen or resynced or hand_over or trace" 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 600 python tools/fuzz_parity.py 200 91 fallen 2>&1 | tail -1
timeout 600 python tools/fuzz_parity.py 200 92 2>&1 | tail -1
