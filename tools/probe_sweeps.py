#!/usr/bin/env python3
"""How many sweeps of the contact solver does an ENVIRONMENT need, and how many does its WAVE run?  The wave leaves the sweeps when all of its
sixteen environments are frozen (PyBullet's solverResidualThreshold, per environment): the difference is what sharing a wave costs, and what
grouping environments by how hard their contacts are could at best win back where the chip is bound by instruction issue (N >= 16384).
A -DQS_PROBE_SWEEPS build of the library counts both in the common-path solver.
usage: QS_LIB_PATH=<probe.so> python tools/probe_sweeps.py [N]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
import torch
from qs_amd import QuadrupedVecEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
env = QuadrupedVecEnv(num_envs=N, device=0, auto_reset=True, reset_lookahead=16, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                      enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1234, info_fields=False)
env.settle_lanes(False)      # (the settle lanes run the same solver: the count is of the environments' own steps)
env.reset_tensor()
g = torch.Generator(device="cuda").manual_seed(1234)
acts = torch.rand((64, N, 6), generator=g, device="cuda") * 2 - 1


def probe():
    out = (C.c_uint64 * 10)()
    assert env.lib.qs_probe_counters(env.h, out) == 0
    return np.array(list(out), dtype=np.int64)


for i in range(600):
    env.step_tensor(acts[i % 64])
p0 = probe()
steps = 300
for i in range(steps):
    env.step_tensor(acts[i % 64])
d = probe() - p0
print(f"{steps} steps of the benchmark workload at N = {N} (settle lanes off): {d[0] / steps / 10:.1f} solves per substep of {N // 16} waves; "
      f"the waves ran {d[1] / max(d[0], 1):.2f} sweeps per solve, an environment needed {d[3] / max(d[2], 1):.2f} until frozen; "
      f"{100 * d[4] / max(d[0], 1):.1f} % of the waves' solves ran every sweep, {100 * d[5] / max(d[2], 1):.2f} % of the environments' never froze")
env.close()
