#!/bin/bash
# round 6's measurement artefacts in one GPU call.  Order: the PMC passes FIRST (profile_round.sh writes
# gpurun_out/prof_<tag>/pmc.json with this tree's source fingerprint), copied to profiles/ at once so that every bench line that follows finds
# the counters of its own binary and carries roofline.traffic / valu_issue; then the driver's command (--steps 20 --warmup 5: the number
# that is quoted first), the 1000-step headline, the other workloads.
#   bash tools/r06_gpu_profiles.sh <dir> <tag>;  then python tools/collect_profiles.py gpurun_out/<dir> gpurun_out/prof_<tag> r06_<x>
OUT=gpurun_out/${1:-r06p}; mkdir -p $OUT
TAG=${2:-r06a}
bash tools/profile_round.sh $TAG > $OUT/profile.log 2>&1; tail -12 $OUT/profile.log
cp gpurun_out/prof_$TAG/pmc.json profiles/${TAG:0:3}_${TAG:3}_pmc.json 2>/dev/null     # (r06a -> profiles/r06_a_pmc.json: where bench.py looks)
run() { name=$1; shift; timeout 900 python bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/$name.json; python -c "
import json; d=json.load(open('$OUT/$name.json')); c=d['config']; print('$name:', round(d['value']/1e6,2),'M (body_contacts', c.get('body_contacts'), ') auto', round((d.get('value_body_contacts_auto') or 0)/1e6,2), 'M;', round(d['ms_per_step'],4),'ms stalls', c.get('stalls'), 'ratio', c.get('settle_work_ratio'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'many-rows wave-substeps', c.get('joint_limit_path_wave_substeps'), 'traffic', d['roofline'].get('traffic'), d['roofline'].get('traffic_note'))"; }
run steps20_command --steps 20 --warmup 5
run headline_8192
run config2_4096 --workload config2_4096 --no-cpu-baseline --no-info-line
run config3_8192 --workload config3_8192 --no-cpu-baseline --no-info-line
run config5_8192 --workload config5_8192 --no-cpu-baseline --no-info-line
run config4_sharded --workload config4_sharded --no-cpu-baseline
run n4096 --envs-per-gpu 4096 --no-cpu-baseline --no-info-line
run n16384 --envs-per-gpu 16384 --no-cpu-baseline --no-info-line
run n65536 --envs-per-gpu 65536 --no-cpu-baseline --no-info-line
run pyramid_resid0 --friction-model pyramid --solver-residual-threshold 0 --no-cpu-baseline --no-info-line
run masses_weld --env-kw env_randomizer_mode=MASS_RANDOMIZER --no-cpu-baseline --no-info-line --no-body-contacts-line
run masses_soft --env-kw env_randomizer_mode=MASS_RANDOMIZER payload=soft --steps 100 --warmup 20 --preroll 200 --no-cpu-baseline --no-info-line --no-body-contacts-line
python tools/numpy_path_rate.py $OUT/numpy_path.json 2>&1 | grep "numpy VecEnv\|step_async"
python tools/time_rare_path.py > $OUT/rare_path.txt 2>&1; grep "ms per step" $OUT/rare_path.txt
python tools/falling_policy_rate.py 16 2>&1 | grep "K =" > $OUT/falling_policy.txt; cat $OUT/falling_policy.txt
python tools/gym_env_rate.py > $OUT/gym_env_rate.txt 2>&1; tail -3 $OUT/gym_env_rate.txt
