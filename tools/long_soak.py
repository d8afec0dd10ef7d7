#!/usr/bin/env python3
"""Hours of simulated time on the benchmark's workload: N = 8192, jump-in-place, U(-1,1) actions, exact look-ahead resets, millions of steps.
Every `check` steps: observations / rewards / states finite, unit quaternions, bounded heights and velocities, counters consistent (every
reset served from its look-ahead slot or counted as a stall; states settled = resets, give or take the ones in flight), rate of the interval.
usage: python tools/long_soak.py [steps=2000000] [check=100000] [envs=8192]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import torch
from qs_amd.vec_env import QuadrupedVecEnv

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
check = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
env = QuadrupedVecEnv(num_envs=n, auto_reset=True, seed=5, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", action_space_mode="SYMMETRIC",
                      motor_control_mode="PD", enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", info_fields=False)
env.reset_tensor()
NAMES = ("resets", "lookahead_served", "lookahead_settled", "reset_stalls")
base = [env.counter(k) for k in NAMES]      # (the initial reset of all environments is not an auto-reset)
g = torch.Generator(device="cuda").manual_seed(2)
acts = torch.rand((256, n, env.action_dim), generator=g, device="cuda") * 2 - 1
done_sum = torch.zeros((), dtype=torch.int64, device="cuda")
t0 = time.perf_counter(); last = 0
for i in range(steps):
    obs, rew, done, trunc = env.step_tensor(acts[(i * 7 + i // 256) % 256])
    done_sum += done.sum()
    if (i + 1) % check == 0 or i == steps - 1:
        st = env.get_state()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        assert torch.isfinite(st).all() and torch.isfinite(obs).all() and torch.isfinite(rew).all(), i
        assert (st[:, 3:7].norm(dim=1) - 1).abs().max() < 1e-3, i
        assert st[:, 2].min() > -0.05 and st[:, 2].max() < 3.0 and st[:, 7:13].abs().max() <= 30.2 and st[:, 25:].abs().max() <= 30.2, i
        resets, served, settled, stalls = (env.counter(k) - b for k, b in zip(NAMES, base))
        assert resets == int(done_sum) and served + stalls == resets, (i, resets, int(done_sum), served, stalls)
        assert abs(settled - resets) <= 16 * n, (i, settled, resets)
        print(f"step {i + 1}: {(i + 1 - last) * n / (t1 - t0) / 1e6:.1f} M env-steps/s over the interval, {resets} resets so far, {served} from look-ahead slots, "
              f"{stalls} stalls, {settled} states settled by the lanes, simulated {(i + 1) * 0.01 / 3600:.2f} h per robot", flush=True)
        last = i + 1; t0 = time.perf_counter()
env.close()
print("ok")
