#!/usr/bin/env python3
"""Steps per second of ONE environment through the reference's own interface (QuadrupedGymEnv.step on a numpy action, gym_env.py:227-256;
BASELINE.json configs[0]: N = 1, springs off, PD, jump-in-place) -- the latency of one launch, not a throughput figure.
usage: python tools/gym_env_rate.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
from qs_amd.env.quadruped_gym_env import QuadrupedGymEnv

out = {}
for name, kw in (("configs0_springs_off", dict(enable_springs=False)), ("springs_on_filter_on", dict(enable_springs=True, enable_action_filter=True))):
    env = QuadrupedGymEnv(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", action_space_mode="SYMMETRIC", motor_control_mode="PD", seed=0, **kw)
    np.random.seed(0)
    env.reset()
    for _ in range(200):
        env.step(np.zeros(env.action_dim))
    n, t0, resets = 1000, time.perf_counter(), 0      # mirrors gym_env.py:460-473: U(-1, 1) actions, reset when done
    for _ in range(n):
        obs, rew, done, info = env.step(np.random.uniform(-1, 1, env.action_dim))
        if done:
            resets += 1
            env.reset()
    dt = time.perf_counter() - t0
    out[name] = dict(steps_per_s=n / dt, us_per_step=1e6 * dt / n, resets=resets)
    print(f"{name}: {n / dt:.0f} env-steps/s ({1e6 * dt / n:.1f} us per step incl. {resets} resets of 2500 settle substeps each)")
    env.close()
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
