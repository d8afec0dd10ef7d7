#!/bin/bash
# the round's measurement artefacts: one bench line per BASELINE config (+ solver variants), the rocprofv3 passes of the headline
OUT=gpurun_out/${1:-r02p}; mkdir -p $OUT
run() { name=$1; shift; timeout 600 python bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/$name.json; python -c "import json; d=json.load(open('$OUT/$name.json')); print('$name:', round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms', d['config'].get('exchange_us'), d['config'].get('settle_work_ratio'))"; }
run headline_8192
run config2_4096 --workload config2_4096 --no-cpu-baseline
run config3_8192 --workload config3_8192 --no-cpu-baseline
run config5_8192 --workload config5_8192 --no-cpu-baseline
run config4_sharded --workload config4_sharded --no-cpu-baseline
run pyramid_resid0 --friction-model pyramid --solver-residual-threshold 0 --no-cpu-baseline
run cone_resid0 --solver-residual-threshold 0 --no-cpu-baseline
run pyramid_resid1e-7 --friction-model pyramid --no-cpu-baseline
run n65536 --envs-per-gpu 65536 --no-pool-streaming --no-cpu-baseline
run n16384 --envs-per-gpu 16384 --no-pool-streaming --no-cpu-baseline
run masses_weld --env-kw env_randomizer_mode=MASS_RANDOMIZER --no-cpu-baseline
run masses_soft --env-kw env_randomizer_mode=MASS_RANDOMIZER payload=soft --steps 100 --warmup 20 --no-cpu-baseline
python tools/time_rare_path.py > $OUT/rare_path.txt 2>&1; grep "ms per step" $OUT/rare_path.txt
[ -f quadruped-springs_amd/qs_amd/exp/prof.so ] && QS_LIB_PATH=$PWD/quadruped-springs_amd/qs_amd/exp/prof.so python tools/phase_profile.py > $OUT/phase_cycles.txt 2>&1
tail -20 $OUT/phase_cycles.txt
bash tools/profile_round.sh ${2:-r02b} > $OUT/profile.log 2>&1
tail -30 $OUT/profile.log
# the same passes with the streaming refill off: what the environments' own steps move (no settle lanes in the launch)
bash tools/profile_round.sh ${2:-r02b}_static --no-pool-streaming > $OUT/profile_static.log 2>&1
tail -16 $OUT/profile_static.log | head -14
