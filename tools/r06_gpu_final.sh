#!/bin/bash
# round 6, closing call: the GPU suite on the final tree with its parity records kept (profiles/r06_impact_parity.jsonl, r06_full_size_oracle_sampled.jsonl,
# r06_terminal_observation_parity.json), smoke(), and the counter passes at N = 65536 (k_step_dense: where the chip is full).
OUT=gpurun_out/${1:-r06z}; mkdir -p $OUT
rm -f gpurun_out/impact_parity.jsonl gpurun_out/full_size_oracle_sampled.jsonl gpurun_out/terminal_observation_parity.json
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
cp gpurun_out/impact_parity.jsonl gpurun_out/full_size_oracle_sampled.jsonl gpurun_out/terminal_observation_parity.json $OUT/ 2>/dev/null
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
QS_PROF_KERNEL=k_step_dense QS_PROF_LAST=300 bash tools/profile_round.sh r06a_n65536 --envs-per-gpu 65536 > $OUT/profile_n65536.log 2>&1; tail -30 $OUT/profile_n65536.log | head -24
