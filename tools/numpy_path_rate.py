#!/usr/bin/env python3
"""Rate of the SB3 numpy path (VecEnv.step on host arrays: load_model.py:113-133) at N = 8192, and where its time goes.
usage: python tools/numpy_path_rate.py [out.json]"""
import cProfile
import json
import os
import pstats
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
import torch
from qs_amd.vec_env import QuadrupedVecEnv

N = 8192
out = {}
for name, kw in (("default", {}), ("views_no_info_block", dict(copy_outputs=False, info_fields=False))):
    env = QuadrupedVecEnv(num_envs=N, auto_reset=True, reset_lookahead=16, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                          enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1, **kw)
    env.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(64, N, 6)).astype(np.float32)
    for i in range(1200):          # spread the episode phases (untimed)
        env.step(acts[i % 64])
    steps, dones, term = 1000, 0, 0
    t0 = time.perf_counter()
    for i in range(steps):
        obs, rew, done, infos = env.step(acts[i % 64])
    dt = (time.perf_counter() - t0) / steps
    for i in range(200):
        obs, rew, done, infos = env.step(acts[i % 64])
        dones += int(done.sum()); term += sum(1 for k in np.flatnonzero(done) if "terminal_observation" in infos[k])
    # the device-tensor path of the same handle, for comparison
    a_dev = torch.as_tensor(acts[:8], device=env.device)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for i in range(steps):
        env.step_tensor(a_dev[i % 8])
    torch.cuda.synchronize(); dt_dev = (time.perf_counter() - t1) / steps
    print(f"{name}: numpy VecEnv.step path {dt * 1e3:.3f} ms/step = {N / dt / 1e6:.1f} M env-steps/s; device-tensor path {dt_dev * 1e3:.3f} ms/step = "
          f"{N / dt_dev / 1e6:.1f} M; {dones} episode ends in 200 steps, {term} with terminal_observation; stalls {env.counter('reset_stalls')}")
    out[name] = dict(ms_per_step=dt * 1e3, env_steps_per_s=N / dt, device_tensor_env_steps_per_s=N / dt_dev, keywords=kw)
    if name == "default":
        pr = cProfile.Profile(); pr.enable()
        for i in range(200):
            env.step(acts[i % 64])
        pr.disable(); pstats.Stats(pr).sort_stats("cumtime").print_stats(10)
    env.close()
if len(sys.argv) > 1:
    out["what"] = "QuadrupedVecEnv.step(numpy actions) -> numpy obs / rewards / dones / infos at N = 8192, jump-in-place, U(-1,1) actions, reset_lookahead 16"
    json.dump(out, open(sys.argv[1], "w"), indent=1)
