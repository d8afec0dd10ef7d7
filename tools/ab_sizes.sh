#!/bin/bash
# headline lines at several launch sizes for several libraries on ONE box: bash tools/ab_sizes.sh <outdir> "<N1 N2 ...>" <name>=<lib.so> ...
out=$1; sizes=$2; shift 2; mkdir -p $out
for n in $sizes; do
  for kv in "$@"; do
    name=${kv%%=*}; lib=${kv#*=}
    QS_LIB_PATH=$PWD/$lib python bench.py --no-cpu-baseline --no-info-line --envs-per-gpu $n > $out/${name}_n$n.json 2> $out/${name}_n$n.err
  done
done
python - "$out" "$sizes" "$@" <<'P'
import json, sys, os
out, sizes = sys.argv[1], sys.argv[2].split()
for kv in sys.argv[3:]:
    name = kv.split("=")[0]
    row = []
    for n in sizes:
        try:
            d = json.load(open(os.path.join(out, f"{name}_n{n}.json"))); row.append(f"N={n}: {d['value'] / 1e6:7.2f} M (stalls {d['config']['stalls']})")
        except Exception as e:
            row.append(f"N={n}: failed")
    print(f"{name:10s} " + "   ".join(row))
P
