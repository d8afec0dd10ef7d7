#!/usr/bin/env python3
"""How much a 20-step window of the benchmark's workload (the links' contact response on) differs from the long-run mean: resets and
many-rows wave-substeps per step in windows of 20 and 500 steps over 16384 steps of random actions.  (No spreading of the episode ages here,
unlike bench.py's preparation: the time-limit resets come in bursts every 1000 steps, the falls do not care.)
    python tools/fall_rate_series.py [steps=16384]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np, torch
from qs_amd.vec_env import QuadrupedVecEnv
n, W = 8192, 20
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
env = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_lookahead=16, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True,
                      enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1234)
env.reset_tensor()
g = torch.Generator(device="cuda").manual_seed(1234)
ring = torch.rand((64, n, 6), generator=g, device="cuda") * 2 - 1
snaps = []
for i in range(steps):
    env.step_tensor(ring[i % 64])
    if (i + 1) % W == 0:
        snaps.append(env.counters_snapshot())
c = torch.stack(snaps).cpu().numpy()
rs = np.diff(c[:, 1], prepend=0) / W; mr = np.diff(c[:, 4], prepend=0) / W
print("window of 500 steps starting at step: resets per step, many-rows wave-substeps per step; min / max of its 20-step windows (many-rows)")
for k in range(0, len(rs), 25):
    print(f"{k * W:6d}: {rs[k:k + 25].mean():6.2f} {mr[k:k + 25].mean():6.2f}   {mr[k:k + 25].min():5.2f} / {mr[k:k + 25].max():5.2f}")
