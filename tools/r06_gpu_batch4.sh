#!/bin/bash
# round 6, fourth GPU call: what the rescue workgroups cost and buy, variant by variant on ONE box (QS_LIB_PATH; the same python side):
#   base = round 5's final source (tools/bin/r06_base.so), vC = + hand-over predicted a substep ahead + a third support point for a leg in the air,
#   vA = vC + the `need` mask out of the hot build's votes, the hot build's resume entry and the ejection code (no rescue workgroups in the kernel),
#   vB = the tree's library: vA + the rescue workgroups' turns as a function of their own; with QS_RESCUE=0 and =1.
OUT=gpurun_out/${1:-r06d}; mkdir -p $OUT
line() { python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']/1e6,2), 'auto', round((d.get('value_body_contacts_auto') or 0)/1e6,2), 'kernel_ms', round(d['roofline'].get('kernel_ms') or 0,4))" $1 "$2"; }
for rep in 1 2; do
  for kv in base=tools/bin/r06_base.so vC=tools/bin/r06_vC.so vA=tools/bin/r06_vA.so vB0=quadruped-springs_amd/qs_amd/libqs_hip.so vB1=quadruped-springs_amd/qs_amd/libqs_hip.so; do
    name=${kv%%=*}; lib=${kv#*=}; r=1; [ $name = vB0 ] && r=0; [ $name = vA ] && r=0
    QS_RESCUE=$r QS_LIB_PATH=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-info-line 2>$OUT/${name}_$rep.err | tail -1 > $OUT/${name}_$rep.json; line $OUT/${name}_$rep.json "$name headline"
  done
done
for kv in vC=tools/bin/r06_vC.so vB1=quadruped-springs_amd/qs_amd/libqs_hip.so; do name=${kv%%=*}; lib=${kv#*=}
  QS_LIB_PATH=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-info-line 2>/dev/null | tail -1 > $OUT/${name}_steps20.json; line $OUT/${name}_steps20.json "$name steps20"
  echo "== $name"; QS_LIB_PATH=$PWD/$lib timeout 300 python tools/time_rare_path.py 2>&1 | grep "ms per step"
  QS_LIB_PATH=$PWD/$lib timeout 300 python bench.py --envs-per-gpu 65536 --no-cpu-baseline --no-info-line 2>/dev/null | tail -1 > $OUT/${name}_n65536.json; line $OUT/${name}_n65536.json "$name N=65536"
done
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -x -k "wave_mates or fuzz or bitwise or resynced or terminal or fallen or support or body_contacts or native" > $OUT/pytest_focus.log 2>&1; tail -8 $OUT/pytest_focus.log
