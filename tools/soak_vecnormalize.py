#!/usr/bin/env python3
"""Soak of the numpy path under VecNormalize (DeviceVecNormalize.step in training mode: qs_host_step_* with qs_host_set_norm, k_norm_moments ->
k_norm_finish writing the mapped host block): many steps at N = 8192 with checks every `check` steps -- everything finite and inside the
clip range, the statistics' count exact, the normalised observations centred, terminal observations present for every episode end, the
process's resident memory flat.   usage: python tools/soak_vecnormalize.py [steps] [check]"""
import os
import resource
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
from qs_amd import DeviceVecNormalize, QuadrupedVecEnv

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
check = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
N = 8192
venv = QuadrupedVecEnv(num_envs=N, device=0, auto_reset=True, reset_lookahead=16, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                       enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=7)
env = DeviceVecNormalize(venv, training=True, norm_reward=True)
obs = env.reset()
rng = np.random.default_rng(0)
ring = rng.uniform(-1, 1, size=(64, N, 6)).astype(np.float32)
ends = with_term = 0
rss0 = None
t0 = time.perf_counter()
for i in range(1, steps + 1):
    obs, rew, done, infos = env.step(ring[i % 64])
    k = np.flatnonzero(done)
    ends += k.size
    if i % 997 == 0:      # (the per-environment check costs a Python loop: sampled)
        with_term += sum(1 for j in k if "terminal_observation" in infos[j] and np.abs(infos[j]["terminal_observation"]).max() <= 10.0) - k.size
    if i % check == 0:
        s = env.get_stats()
        rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
        rss0 = rss0 or rss
        ok = (np.isfinite(obs).all() and np.isfinite(rew).all() and np.abs(obs).max() <= 10.0 and np.abs(rew).max() <= 10.0 and
              abs(s["obs_count"] - (1e-4 + (i + 1) * N)) < 1e-3 and np.all(s["obs_var"] > 0) and with_term == 0 and venv.counter("reset_stalls") == 0)
        dt = time.perf_counter() - t0
        print(f"step {i}: ok {bool(ok)}, {i * N / dt / 1e6:.1f} M env-steps/s, {ends} episode ends, |mean of normalised obs| <= {np.abs(obs.mean(axis=0)).max():.3f}, "
              f"obs_count {s['obs_count']:.4f}, max RSS {rss:.0f} MiB (+{rss - rss0:.0f} since the first check)", flush=True)
        if not ok:
            sys.exit(1)
print("done")
env.close()
