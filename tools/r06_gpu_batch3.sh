#!/bin/bash
# round 6, third GPU call: the rescue workgroups (a falling robot leaves its wave; QS_RESCUE=0 switches them off at qs_create) -- bounded
# first checks, counters, then A/B on one box with the same library.      bash tools/r06_gpu_batch3.sh <dir>
OUT=gpurun_out/${1:-r06c}; mkdir -p $OUT
timeout 300 python - > $OUT/rescue_counters.txt 2>&1 <<'P'
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import torch
from qs_amd.vec_env import QuadrupedVecEnv
env = QuadrupedVecEnv(num_envs=8192, auto_reset=True, seed=5, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", action_space_mode="SYMMETRIC",
                      motor_control_mode="PD", enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", info_fields=False)
env.reset_tensor()
g = torch.Generator(device="cuda").manual_seed(2)
acts = torch.rand((64, 8192, env.action_dim), generator=g, device="cuda") * 2 - 1
for i in range(1500):
    env.step_tensor(acts[i % 64])
torch.cuda.synchronize(); t0 = time.perf_counter()
c0 = {k: env.counter(k) for k in ("rescued", "rescue_lost", "limit_path_substeps", "resets")}
for i in range(1000):
    env.step_tensor(acts[i % 64])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
c1 = {k: env.counter(k) for k in c0}
print({k: (c1[k] - c0[k]) / 1000 for k in c0}, "per step;", round(8192 * 1000 / dt / 1e6, 2), "M env-steps/s through the python loop")
P
cat $OUT/rescue_counters.txt | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -x -k "wave_mates or fuzz or bitwise or resynced or terminal or fallen or support or body_contacts or native" > $OUT/pytest_focus.log 2>&1; tail -15 $OUT/pytest_focus.log
for rep in 1 2; do for r in 0 1; do
  QS_RESCUE=$r timeout 300 python bench.py --no-cpu-baseline --no-info-line 2>/dev/null | tail -1 > $OUT/headline_rescue${r}_$rep.json
  QS_RESCUE=$r timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-info-line 2>/dev/null | tail -1 > $OUT/steps20_rescue${r}_$rep.json
  python -c "
import json
for f in ('headline', 'steps20'):
    d = json.load(open('$OUT/%s_rescue${r}_$rep.json' % f)); print(f, 'QS_RESCUE=$r', round(d['value']/1e6,2), 'auto', round((d.get('value_body_contacts_auto') or 0)/1e6,2), 'kernel_ms', d['roofline'].get('kernel_ms'))"
done; done
for r in 0 1; do echo "== QS_RESCUE=$r"; QS_RESCUE=$r timeout 300 python tools/time_rare_path.py 2>&1 | grep "ms per step"; done
for r in 0 1; do QS_RESCUE=$r timeout 300 python bench.py --envs-per-gpu 65536 --no-cpu-baseline --no-info-line 2>/dev/null | tail -1 > $OUT/n65536_rescue$r.json
  python -c "import json; d=json.load(open('$OUT/n65536_rescue$r.json')); print('N=65536 QS_RESCUE=$r', round(d['value']/1e6,2), 'auto', round((d.get('value_body_contacts_auto') or 0)/1e6,2))"; done
