#!/bin/bash
# launch-time spread of the step kernel: PyBullet's early exit (sweep counts differ between waves) against all 30 sweeps everywhere
OUT=$PWD/gpurun_out/${1:-r02h}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in default "--solver-residual-threshold 0"; do
  tag=$(echo $v | tr -d ' -' | cut -c1-12)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1000 --warmup 50 --no-cpu-baseline ${v/default/} > $OUT/tr_$tag.log 2>&1
  f=$(find $OUT/tr_$tag -name "*_kernel_stats.csv" | head -1)
  echo "== $v"; grep "k_step" $f | cut -d, -f1-9 | sed 's/(qs_config.*DemoTab)//'
  t=$(find $OUT/tr_$tag -name "*_kernel_trace.csv" | head -1)
  python3 - "$t" <<'PY'
import csv, sys, numpy as np
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("void k_step")]
d=np.array(d[200:]); print("launches", len(d), "us: min %.1f p10 %.1f median %.1f mean %.1f p90 %.1f p99 %.1f max %.1f" % (d.min(), *np.percentile(d,[10,50]), d.mean(), *np.percentile(d,[90,99]), d.max()))
PY
  rm -rf $OUT/tr_$tag
done
