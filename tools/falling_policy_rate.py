#!/usr/bin/env python3
"""What the look-ahead resets do under a policy that throws the robot down all the time (the start of a training run): env-steps/s, resets
per step, stalls and backlog at N = 8192.  usage: python tools/falling_policy_rate.py [K ...]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import torch
from qs_amd.vec_env import QuadrupedVecEnv
BC = {"true": True, "auto": "auto", "false": False}[__import__("os").environ.get("QS_BODY_CONTACTS", "true").lower()]   # the links' contact response: the default (True) or QS_BODY_CONTACTS=auto

n = 8192
for K in [int(x) for x in sys.argv[1:]] or [8, 16]:
    v = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_lookahead=K, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True,
                        enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=3, info_fields=False, body_contacts=BC)
    v.reset_tensor()
    gen = torch.Generator(device=v.device).manual_seed(1)
    acts = torch.rand((64, n, v.action_dim), generator=gen, device=v.device) * 2 - 1
    acts[:, :, 1::3] = -1.0                                     # thighs back ...
    acts[0::2, :, 2::3] = 1.0; acts[1::2, :, 2::3] = -0.5       # ... calves flailing: the robots fall within a few dozen steps
    for phase in ("first 1000 steps", "next 2000 steps"):
        steps = 1000 if phase.startswith("first") else 2000
        c0 = {k: v.counter(k) for k in ("resets", "reset_stalls", "settle_substeps")}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            v.step_tensor(acts[i % 64])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        c1 = {k: v.counter(k) for k in c0}
        print(f"K = {K:2d}, {phase}: {n * steps / dt / 1e6:7.2f} M env-steps/s, {(c1['resets'] - c0['resets']) / steps:6.1f} resets per step, "
              f"{c1['reset_stalls'] - c0['reset_stalls']} stalls, settle substeps per env substep {(c1['settle_substeps'] - c0['settle_substeps']) / (n * steps * 10):.2f}, "
              f"backlog {v.counter('lookahead_backlog')}")
    v.close()
