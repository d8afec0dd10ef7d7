#!/usr/bin/env python3
"""How often a pooled auto-reset draws an entry that another reset already used and the streaming refill has not re-settled yet, against the
pool size (the draw is Philox(env, episode) mod P; tests/test_gpu_round2.py bounds it at the bench's pool size).  usage: python tools/pool_reuse.py [P ...]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
import torch
from qs_amd.vec_env import QuadrupedVecEnv

n, steps = 8192, 2000
for P in [int(x) for x in sys.argv[1:]] or [16384, 65536, 262144]:
    t0 = time.perf_counter()
    v = QuadrupedVecEnv(num_envs=n, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True,
                        env_randomizer_mode="TEST_RANDOMIZER", auto_reset=True, reset_pool=P, seed=11, noise=False)
    v.reset_tensor(); torch.cuda.synchronize()
    t_create = time.perf_counter() - t0
    v.pool_streaming(True)
    gen = torch.Generator(device=v.device).manual_seed(2)
    draws = []
    for t in range(steps):
        act = torch.rand((n, v.action_dim), generator=gen, device=v.device) * 2 - 1
        _, _, done, _ = v.step_tensor(act)
        if t % 4 == 0:
            idx = torch.nonzero(done).flatten()
            if len(idx):
                draws.append(v.get_info("params")[idx].cpu().numpy())
    d = np.concatenate(draws)
    _, counts = np.unique(d.view(np.dtype((np.void, d.dtype.itemsize * d.shape[1]))), return_counts=True)
    reused = int((counts - 1).sum())
    print(f"pool {P:7d}: create + first reset {t_create:.2f} s, {len(d)} resets sampled, {100.0 * reused / len(d):.2f} % drew an entry already used and not yet re-settled")
    v.close()
