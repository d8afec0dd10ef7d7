#!/usr/bin/env python3
"""Step time when waves take the many-rows solver path: NO_TASK (body contacts on), every 4th / every robot lying on its side under small
random torques, against the same robots standing.  usage: python tools/time_rare_path.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
import torch
from scipy.spatial.transform import Rotation as Rot
from qs_amd.vec_env import QuadrupedVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
env = QuadrupedVecEnv(num_envs=n, auto_reset=False, task_env="NO_TASK", observation_space_mode="ENCODER", enable_springs=True, enable_action_filter=False,
                      isRLGymInterface=False, motor_control_mode="TORQUE", env_randomizer_mode="NONE", noise=False)
env.reset_tensor()
stand = env.get_state().clone()
g = torch.Generator(device="cuda").manual_seed(0)
for name, every in (("all standing", 0), ("one robot in 64 on its side", 64), ("every 4th robot on its side", 4), ("every robot on its side", 1)):
    s = stand.clone()
    if every:
        idx = torch.arange(0, n, every, device=s.device)
        s[idx, 2] = 0.12
        s[idx, 3:7] = torch.tensor(Rot.from_euler("x", 1.45).as_quat(), dtype=torch.float32, device=s.device)
        s[idx, 13:25] = torch.tensor(np.tile([0.0, 1.2, -2.4], 4), dtype=torch.float32, device=s.device)
    env.set_state(s)
    ring = [(torch.rand((n, 12), generator=g, device="cuda") - 0.5) * 4 for _ in range(16)]     # (the actions are made outside the timed loop)
    for k in range(20):
        env.step_tensor(ring[k % 16])
    c0 = env.counter("limit_path_substeps")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(50):
        env.step_tensor(ring[k % 16])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"{name:32s} {1e3 * dt:8.3f} ms per step  ({n / dt / 1e6:6.1f} M env-steps/s), many-rows wave-substeps per step: {(env.counter('limit_path_substeps') - c0) / 50:.0f}")
