#!/usr/bin/env python3
"""What DeviceVecNormalize (SB3's VecNormalize on the device, load_model.py:109-137) adds to a step at N = 8192: the device-tensor path in
training and evaluation mode against the bare step, and the numpy path (VecNormalize.step on host arrays, what load_model.py's loop calls).
usage: python tools/time_vecnormalize.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
import torch
from qs_amd import DeviceVecNormalize, QuadrupedVecEnv
BC = {"true": True, "auto": "auto", "false": False}[__import__("os").environ.get("QS_BODY_CONTACTS", "true").lower()]   # the links' contact response: the default (True) or QS_BODY_CONTACTS=auto

N = 8192
venv = QuadrupedVecEnv(num_envs=N, device=0, auto_reset=True, reset_lookahead=16, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                       enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1, body_contacts=BC)
env = DeviceVecNormalize(venv)
obs = env.reset_tensor()
ring = torch.rand((64, N, 6), device="cuda") * 2 - 1
ring_np = ring.cpu().numpy()
k = [0]


def dev_step(e):
    k[0] += 1
    e.step_tensor(ring[k[0] % 64])


def np_step(e):
    k[0] += 1
    e.step(ring_np[k[0] % 64])


def legacy_np_step(e):
    """the numpy path as rounds 2-4 had it: device step, normalisation on the device, four pageable copies back, N fresh dicts, terminal
    observations normalised on the host"""
    k[0] += 1
    v = e.venv
    v._act.copy_(torch.from_numpy(ring_np[k[0] % 64]))
    obs, rew, done, trunc = e.step_tensor(v._act)
    obs, rew = obs.cpu().numpy().copy(), rew.cpu().numpy().copy()
    done, trunc = done.cpu().numpy().astype(bool), trunc.cpu().numpy().astype(bool)
    infos = [{} for _ in range(e.num_envs)]
    if done.any():
        term = e.normalize_obs(v.get_info("terminal_obs").cpu().numpy())
        for i in np.nonzero(done)[0]:
            infos[i]["TimeLimit.truncated"] = bool(trunc[i])
            infos[i]["terminal_observation"] = term[i].copy()
    return obs, rew, done, infos


def timeit(f, n=400, warm=40):
    for _ in range(warm):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for _ in range(1500):      # spread the episode phases / let the settle lanes reach their steady state
    dev_step(venv)
out = {}
out["raw_step_tensor_ms"] = timeit(lambda: dev_step(venv))
env.training = True
out["normalised_training_ms"] = timeit(lambda: dev_step(env))
env.training = False
out["normalised_eval_ms"] = timeit(lambda: dev_step(env))
out["raw_numpy_step_ms"] = timeit(lambda: np_step(venv), n=200, warm=20)
env.training = True
out["normalised_numpy_training_ms"] = timeit(lambda: np_step(env), n=100, warm=10)
env.training = False; env.norm_reward = False      # load_model.py:113-116
out["normalised_numpy_eval_ms"] = timeit(lambda: np_step(env), n=100, warm=10)
out["legacy_normalised_numpy_eval_ms"] = timeit(lambda: legacy_np_step(env), n=100, warm=10)
for key, v in out.items():
    if key.endswith("_ms"):
        print(f"{key:36s} {v:.4f} ms  ({N / v / 1e3:.1f} M env-steps/s)")
out["what"] = f"N = {N}, jump-in-place, U(-1,1) actions, reset_lookahead 16; device-tensor path and numpy path, per step"
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
env.close()
