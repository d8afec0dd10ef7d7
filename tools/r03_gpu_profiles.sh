#!/bin/bash
# round 3's measurement artefacts in one GPU call: one bench line per BASELINE config and variant, the numpy-path rate, the split probe,
# the rare-path timings, the rocprofv3 passes of the headline.   bash tools/r03_gpu_profiles.sh <dir> <tag>
OUT=gpurun_out/${1:-r03p}; mkdir -p $OUT
run() { name=$1; shift; timeout 900 python bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/$name.json; python -c "import json; d=json.load(open('$OUT/$name.json')); c=d['config']; print('$name:', round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms stalls', c.get('stalls'), 'ratio', c.get('settle_work_ratio'), 'info', d.get('value_info_fields_true'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; }
run headline_8192
run config2_4096 --workload config2_4096 --no-cpu-baseline --no-info-line
run config3_8192 --workload config3_8192 --no-cpu-baseline --no-info-line
run config5_8192 --workload config5_8192 --no-cpu-baseline --no-info-line
run config4_sharded --workload config4_sharded --no-cpu-baseline
run lookahead8 --reset-lookahead 8 --no-cpu-baseline --no-info-line
run lookahead0_exact --reset-lookahead 0 --steps 200 --preroll 0 --no-cpu-baseline --no-info-line
run pyramid_resid0 --friction-model pyramid --solver-residual-threshold 0 --no-cpu-baseline --no-info-line
run cone_resid0 --solver-residual-threshold 0 --no-cpu-baseline --no-info-line
run n4096 --envs-per-gpu 4096 --no-cpu-baseline --no-info-line
run n16384 --envs-per-gpu 16384 --no-cpu-baseline --no-info-line
run n65536 --envs-per-gpu 65536 --no-cpu-baseline --no-info-line
run masses_weld --env-kw env_randomizer_mode=MASS_RANDOMIZER --no-cpu-baseline --no-info-line
run masses_soft --env-kw env_randomizer_mode=MASS_RANDOMIZER payload=soft --steps 100 --warmup 20 --preroll 200 --no-cpu-baseline --no-info-line
python tools/numpy_path_rate.py $OUT/numpy_path.json 2>&1 | grep "numpy VecEnv\|step_async"
python tools/time_rare_path.py > $OUT/rare_path.txt 2>&1; grep "ms per step" $OUT/rare_path.txt
python tools/falling_policy_rate.py 16 2>&1 | grep "K =" > $OUT/falling_policy.txt; cat $OUT/falling_policy.txt
for x in 0 185; do ./tools/bin/split_probe --extra $x; done > $OUT/split_probe.jsonl 2>&1; cat $OUT/split_probe.jsonl
bash tools/profile_round.sh ${2:-r03a} > $OUT/profile.log 2>&1
tail -30 $OUT/profile.log
