#!/usr/bin/env python3
"""What do the environments that reach the many-rows solve look like in the benchmark with body_contacts=True?  A -DQS_PROBE_WARM build counts,
per environment-substep with rows of its own: a joint at its stop / a leg with two live support points / at most one point per leg (1, 2, 3+
points in all), and per wave-substep of the full build: with a many-rows solve, with only "simple" environments in it.
    QS_HIPCC_EXTRA=-DQS_PROBE_WARM QS_BUILD_OUT=$PWD/tools/bin/probe_warm.so python quadruped-springs_amd/build.py --force
    QS_LIB_PATH=$PWD/tools/bin/probe_warm.so python tools/probe_warm.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
import torch
from qs_amd import QuadrupedVecEnv

N = 8192
env = QuadrupedVecEnv(num_envs=N, device=0, auto_reset=True, reset_lookahead=16, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                      enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1234, info_fields=False, body_contacts=True)
env.reset_tensor()
g = torch.Generator(device="cuda").manual_seed(1234)
acts = torch.rand((64, N, 6), generator=g, device="cuda") * 2 - 1
for i in range(1500):
    env.step_tensor(acts[i % 64])


def probe():
    out = (C.c_uint64 * 10)()
    assert env.lib.qs_probe_counters(env.h, out) == 0
    return np.array(list(out), dtype=np.int64)


steps = 1000
p0, r0 = probe(), env.counter("resets")
for i in range(steps):
    env.step_tensor(acts[i % 64])
d, r1 = probe() - p0, env.counter("resets")
t = max(d[0], 1)
print(f"{steps} steps of the benchmark workload with body_contacts=True ({(r1 - r0) / steps:.1f} resets per step)")
print(f"environment-substeps with rows of their own (many-rows solves): {d[0] / steps:.2f} per step")
print(f"   a joint at its stop:                         {100 * d[1] / t:5.1f} %")
print(f"   no stop, some leg with two live points:      {100 * d[2] / t:5.1f} %")
print(f"   no stop, one point per leg at most, 1 point: {100 * d[3] / t:5.1f} %")
print(f"                                      2 points: {100 * d[4] / t:5.1f} %")
print(f"                                     3+ points: {100 * d[5] / t:5.1f} %")
print(f"wave-substeps in the full build: {d[6] / steps:.2f} per step; with a many-rows solve {d[7] / steps:.2f}; of those with only simple environments {d[8] / steps:.2f} ({100 * d[8] / max(d[7], 1):.0f} %)")
env.close()
