#!/usr/bin/env python3
"""The numpy (host) path over many steps next to a twin on the device path: same seed, same actions -- observations, rewards, dones and the
terminal observations of every step must be equal, in both host modes.  usage: python tools/diag/r03_host_path_soak.py [steps=20000]"""
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
import torch
from qs_amd.vec_env import QuadrupedVecEnv

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n = 2048
kw = dict(num_envs=n, auto_reset=True, seed=9, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True,
          enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", noise=True)
for mode in ("zero_copy", "copy"):
    if mode == "copy":
        os.environ["QS_HOST_PATH"] = "copy"
    a, b = QuadrupedVecEnv(copy_outputs=False, **kw), QuadrupedVecEnv(**kw)
    assert np.array_equal(a.reset(), b.reset_tensor().cpu().numpy())
    rng = np.random.default_rng(3)
    acts = rng.uniform(-1, 1, size=(64, n, 6)).astype(np.float32)
    dones = terms = 0
    for t in range(steps):
        obs, rew, done, infos = a.step(acts[t % 64])
        ob, rb, db, tb = b.step_tensor(torch.as_tensor(acts[t % 64], device=b.device))
        if t % 50 == 0 or done.any():
            assert np.array_equal(obs, ob.cpu().numpy()) and np.array_equal(rew, rb.cpu().numpy()) and np.array_equal(done, db.cpu().numpy().astype(bool)), (mode, t)
        if done.any():
            term = b.get_info("terminal_obs").cpu().numpy()
            for i in np.flatnonzero(done):
                assert np.array_equal(infos[i]["terminal_observation"], term[i]), (mode, t, i)
                terms += 1
            dones += int(done.sum())
    print(f"{mode}: {steps} steps x {n} environments, {dones} episode ends, {terms} terminal observations equal, stalls {a.counter('reset_stalls')} / {b.counter('reset_stalls')}")
    a.close(); b.close()
print("ok")
