#!/bin/bash
# kernel-trace average of the timed region's launches for two libraries on one box
REPO=$(pwd); OUT=$REPO/gpurun_out/r04q; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for name in r03 new; do
  if [ $name = r03 ]; then export QS_LIB_PATH=$REPO/tools/bin/r03.so; else unset QS_LIB_PATH; fi
  rm -rf $OUT/tr_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$name -- python3 $REPO/bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-info-line > $OUT/tr_$name.log 2>&1
  tail -1 $OUT/tr_$name.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name live under profiler: value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'])"
  python3 - $OUT/tr_$name <<'P'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)[0]
d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("void k_step<")]
last = d[-1000:]
print(len(d), "launches; last 1000: avg %.2f us, median %.2f, min %.2f" % (sum(last) / len(last), sorted(last)[500], min(last)))
P
  rm -rf $OUT/tr_$name
done
