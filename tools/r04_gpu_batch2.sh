#!/bin/bash
# round 4, second half: VecNormalize in one launch + on the host path, spin-wait of the host path, what a handle costs to create,
# empty settle-lane workgroups, N = 1 latency.   usage: bash tools/r04_gpu_batch2.sh <out dir>
out=${1:-gpurun_out/r04z}
mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -k "vec_normalize or examples_run" > $out/pytest_norm.txt 2>&1; tail -2 $out/pytest_norm.txt
python tools/numpy_path_rate.py $out/numpy_spin.json > $out/numpy_spin.txt 2>&1; grep -E "numpy VecEnv|step_async" $out/numpy_spin.txt
QS_HOST_SPIN_US=0 python tools/numpy_path_rate.py $out/numpy_block.json > $out/numpy_block.txt 2>&1; grep -E "numpy VecEnv|step_async" $out/numpy_block.txt
python tools/time_vecnormalize.py $out/vecnorm_one_launch.json > $out/vecnorm_one_launch.txt 2>&1; cat $out/vecnorm_one_launch.txt | tail -8
python tools/create_cost.py $out/create_cost.json > $out/create_cost.txt 2>&1; cat $out/create_cost.txt | tail -7
python tools/empty_lanes_cost.py $out/empty_lanes.json > $out/empty_lanes.txt 2>&1; tail -4 $out/empty_lanes.txt
python tools/gym_env_rate.py $out/gym_env_rate.json > $out/gym_env_rate.txt 2>&1; tail -3 $out/gym_env_rate.txt
