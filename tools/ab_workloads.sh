#!/bin/bash
out=gpurun_out/r04zn; mkdir -p $out
for lib in cur lot11; do
  if [ $lib = cur ]; then L=$PWD/quadruped-springs_amd/qs_amd/libqs_hip.so; else L=$PWD/quadruped-springs_amd/qs_amd/exp/lot11.so; fi
  QS_LIB_PATH=$L python bench.py --no-cpu-baseline --no-info-line --envs-per-gpu 65536 > $out/${lib}_n65536.json 2>/dev/null
  QS_LIB_PATH=$L python bench.py --no-cpu-baseline --no-info-line --envs-per-gpu 16384 > $out/${lib}_n16384.json 2>/dev/null
  QS_LIB_PATH=$L python bench.py --no-cpu-baseline --no-info-line --env-kw body_contacts=True > $out/${lib}_bc.json 2>/dev/null
  QS_LIB_PATH=$L python bench.py --no-cpu-baseline --no-info-line --env-kw payload=soft env_randomizer_mode=MASS_RANDOMIZER > $out/${lib}_soft.json 2>/dev/null
  QS_LIB_PATH=$L python bench.py --no-cpu-baseline --no-info-line --workload config5_8192 > $out/${lib}_c5.json 2>/dev/null
  QS_LIB_PATH=$L python bench.py --no-cpu-baseline --no-info-line --workload config2_4096 > $out/${lib}_c2.json 2>/dev/null
  QS_LIB_PATH=$L python bench.py --no-cpu-baseline --no-info-line --friction-model pyramid --solver-residual-threshold 0 > $out/${lib}_pyr.json 2>/dev/null
done
python - <<'P'
import json,glob,os
for w in ("n65536","n16384","bc","soft","c5","c2","pyr"):
    v=[json.load(open(f"gpurun_out/r04zn/{l}_{w}.json"))["value"]/1e6 for l in ("cur","lot11")]
    print(f"{w:8s} cur {v[0]:7.2f}  lot11 {v[1]:7.2f}  {100*(v[1]/v[0]-1):+.1f} %")
P
