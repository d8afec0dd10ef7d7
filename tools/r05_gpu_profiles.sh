#!/bin/bash
# round 5's measurement artefacts in one GPU call.  Since this round `value` is taken with the reference's all-links contact response
# (body_contacts=True, the default) and every line carries value_body_contacts_auto next to it.
#   bash tools/r05_gpu_profiles.sh <dir> <tag>;  then python tools/collect_profiles.py gpurun_out/<dir> gpurun_out/prof_<tag> r05_<x>
OUT=gpurun_out/${1:-r05p}; mkdir -p $OUT
run() { name=$1; shift; timeout 900 python bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/$name.json; python -c "
import json; d=json.load(open('$OUT/$name.json')); c=d['config']; print('$name:', round(d['value']/1e6,2),'M (body_contacts', c.get('body_contacts'), ') auto', round((d.get('value_body_contacts_auto') or 0)/1e6,2), 'M;', round(d['ms_per_step'],4),'ms stalls', c.get('stalls'), 'ratio', c.get('settle_work_ratio'), 'info', d.get('value_info_fields_true'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'many-rows wave-substeps', c.get('joint_limit_path_wave_substeps'), 'traffic', d['roofline'].get('traffic'), d['roofline'].get('traffic_note'))"; }
run headline_8192
run steps20_command --steps 20 --warmup 5 --no-cpu-baseline
run config2_4096 --workload config2_4096 --no-cpu-baseline --no-info-line
run config3_8192 --workload config3_8192 --no-cpu-baseline --no-info-line
run config5_8192 --workload config5_8192 --no-cpu-baseline --no-info-line
run config4_sharded --workload config4_sharded --no-cpu-baseline
run lookahead0_exact --reset-lookahead 0 --steps 200 --preroll 0 --no-cpu-baseline --no-info-line --no-body-contacts-line
run pyramid_resid0 --friction-model pyramid --solver-residual-threshold 0 --no-cpu-baseline --no-info-line
run n4096 --envs-per-gpu 4096 --no-cpu-baseline --no-info-line
run n16384 --envs-per-gpu 16384 --no-cpu-baseline --no-info-line
run n65536 --envs-per-gpu 65536 --no-cpu-baseline --no-info-line
run masses_weld --env-kw env_randomizer_mode=MASS_RANDOMIZER --no-cpu-baseline --no-info-line --no-body-contacts-line
run masses_soft --env-kw env_randomizer_mode=MASS_RANDOMIZER payload=soft --steps 100 --warmup 20 --preroll 200 --no-cpu-baseline --no-info-line --no-body-contacts-line
python tools/numpy_path_rate.py $OUT/numpy_path.json 2>&1 | grep "numpy VecEnv\|step_async"
QS_BODY_CONTACTS=auto python tools/numpy_path_rate.py $OUT/numpy_path_auto.json 2>&1 | grep "numpy VecEnv\|step_async"
python tools/time_rare_path.py > $OUT/rare_path.txt 2>&1; grep "ms per step" $OUT/rare_path.txt
python tools/falling_policy_rate.py 16 2>&1 | grep "K =" > $OUT/falling_policy.txt; { echo "the same with body_contacts=\"auto\":"; QS_BODY_CONTACTS=auto python tools/falling_policy_rate.py 16 2>&1 | grep "K ="; } >> $OUT/falling_policy.txt; cat $OUT/falling_policy.txt
python tools/gym_env_rate.py > $OUT/gym_env_rate.txt 2>&1; tail -3 $OUT/gym_env_rate.txt
bash tools/profile_round.sh ${2:-r05a} > $OUT/profile.log 2>&1
tail -30 $OUT/profile.log
