#!/usr/bin/env python3
"""The long soak's workload with the links' contact response on (body_contacts=True): every fall goes through the hand-over and the
many-rows solve.  usage: python tools/soak_body_contacts.py [steps=200000] [check=50000]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import torch
from qs_amd.vec_env import QuadrupedVecEnv

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
check = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
n = 8192
env = QuadrupedVecEnv(num_envs=n, auto_reset=True, seed=5, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True,
                      enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", info_fields=False, body_contacts=True)
env.reset_tensor()
g = torch.Generator(device="cuda").manual_seed(2)
acts = torch.rand((256, n, env.action_dim), generator=g, device="cuda") * 2 - 1
t0 = time.perf_counter(); last = 0; r0 = env.counter("resets"); m0 = env.counter("limit_path_substeps")
for i in range(1, steps + 1):
    obs, rew, done, trunc = env.step_tensor(acts[i % 256])
    if i % check == 0:
        s = env.get_state()
        ok = bool(torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(s).all())
        qn = float((torch.linalg.norm(s[:, 3:7], dim=1) - 1).abs().max())
        dt = time.perf_counter() - t0; t0 = time.perf_counter()
        print(f"step {i}: finite {ok}, |quat| - 1 <= {qn:.1e}, z in [{float(s[:, 2].min()):.3f}, {float(s[:, 2].max()):.3f}], max |qd| {float(s[:, 25:].abs().max()):.1f}, "
              f"resets {env.counter('resets') - r0}, stalls {env.counter('reset_stalls')}, many-rows wave-substeps {env.counter('limit_path_substeps') - m0}, "
              f"{n * (i - last) / dt / 1e6:.1f} M env-steps/s", flush=True)
        last = i
        assert ok and qn < 1e-4
print("done")
