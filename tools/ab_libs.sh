#!/bin/bash
# A/B of step libraries on ONE box: bash tools/ab_libs.sh <outdir> <name>=<lib.so> ... ; the headline bench line of each, twice, interleaved
out=$1; shift; mkdir -p $out
for rep in 1 2; do
  for kv in "$@"; do
    name=${kv%%=*}; lib=${kv#*=}
    QS_LIB_PATH=$PWD/$lib python bench.py --no-cpu-baseline --no-info-line > $out/${name}_$rep.json 2> $out/${name}_$rep.err
  done
done
python - "$out" "$@" <<'P'
import json, sys, glob, os
out = sys.argv[1]
for kv in sys.argv[2:]:
    name = kv.split("=")[0]
    v = [json.load(open(f))["value"] / 1e6 for f in sorted(glob.glob(os.path.join(out, name + "_*.json")))]
    print(f"{name:12s} " + " ".join(f"{x:7.2f}" for x in v) + " M env-steps/s")
P
