#!/bin/bash
# A/B of step libraries in the two-waves-per-SIMD regime: bash tools/ab_sizes_libs.sh <outdir> <name>=<lib.so> ... ; N = 16384 and 65536, twice each
out=$1; shift; mkdir -p $out
for rep in 1 2; do
  for kv in "$@"; do
    name=${kv%%=*}; lib=${kv#*=}
    for n in 16384 65536; do
      QS_LIB_PATH=$PWD/$lib python bench.py --no-cpu-baseline --no-info-line --envs-per-gpu $n > $out/${name}_${n}_$rep.json 2> $out/${name}_${n}_$rep.err
    done
  done
done
python - "$out" "$@" <<'P'
import json, sys, glob, os
out = sys.argv[1]
for kv in sys.argv[2:]:
    name = kv.split("=")[0]
    row = []
    for n in (16384, 65536):
        row.append(f"N = {n}: " + " ".join(f"{json.load(open(f))['value'] / 1e6:7.2f}" for f in sorted(glob.glob(os.path.join(out, f"{name}_{n}_*.json")))))
    print(f"{name:10s} " + "   ".join(row) + "  M env-steps/s")
P
