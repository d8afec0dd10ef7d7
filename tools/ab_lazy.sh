#!/bin/bash
# A/B of a second build of the library (quadruped-springs_amd/qs_amd/exp/lazy.so: QS_BUILD_OUT=... python quadruped-springs_amd/build.py --force) against the
# one in place on what a change of the rare paths moves: the headline, the headline with body_contacts=True (twice each) and tools/time_rare_path.py.
# The support points' rule of DESIGN.md 4a was measured with it (gpurun_out/r04zu).   bash tools/ab_lazy.sh   on the GPU box
out=gpurun_out/r04zu; mkdir -p $out
for rep in 1 2; do for lib in cur lazy; do
  if [ $lib = cur ]; then L=$PWD/quadruped-springs_amd/qs_amd/libqs_hip.so; else L=$PWD/quadruped-springs_amd/qs_amd/exp/lazy.so; fi
  QS_LIB_PATH=$L python bench.py --no-cpu-baseline --no-info-line > $out/${lib}_head_$rep.json 2>/dev/null
  QS_LIB_PATH=$L python bench.py --no-cpu-baseline --no-info-line --env-kw body_contacts=True > $out/${lib}_bc_$rep.json 2>/dev/null
done; done
for lib in cur lazy; do
  if [ $lib = cur ]; then L=$PWD/quadruped-springs_amd/qs_amd/libqs_hip.so; else L=$PWD/quadruped-springs_amd/qs_amd/exp/lazy.so; fi
  QS_LIB_PATH=$L python tools/time_rare_path.py 2>&1 | grep "ms per step" > $out/${lib}_rare.txt
done
python - <<'P'
import json
for w in ("head","bc"):
    for l in ("cur","lazy"):
        v=[json.load(open(f"gpurun_out/r04zu/{l}_{w}_{r}.json")) for r in (1,2)]
        print(w, l, [round(x["value"]/1e6,2) for x in v], "many-rows wave-substeps/step", [x["config"]["joint_limit_path_wave_substeps"]/x["steps"] for x in v])
P
echo cur; cat $out/cur_rare.txt; echo lazy; cat $out/lazy_rare.txt
