#!/usr/bin/env python3
"""Substep by substep: the env step of tools/diag/r04_full_size_case.py's deviating environment (config2_4096, step 51, environment 692) on the
device and in the float64 oracle, through both trace taps.  usage: python tools/diag/r04_full_size_trace.py [name step env]"""
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
from test_gpu_parity import FULL_SIZE
from oracle.qso import Oracle
from qs_amd.config import build_config
from qs_amd.vec_env import QuadrupedVecEnv

name, at_step, at_env = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("config2_4096", 51, 692)
n, kw = FULL_SIZE[name]
kw = dict(dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True), **kw, seed=7, noise=False)
v = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_lookahead=16, **kw)
rng = np.random.default_rng(sum(map(ord, name)))
blocks = [int(b) * 64 for b in sorted(rng.choice(n // 64, size=4, replace=False))]
o64 = [Oracle(build_config(n_envs=64, auto_reset=True, env_id_offset=b, **kw)[0]) for b in blocks]
d, dt = v.action_dim, float(v.cfg.dt)
v.reset_tensor()
for o in o64:
    o.reset()
np.set_printoptions(precision=5, suppress=True, linewidth=250)
for i in range(at_step + 1):
    a = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    if i % 20 > 8:
        rough = [1.0, 1.0, 1.0, 1.0, -1.0] if d == 5 else (np.tile([0.0, -1.0, 1.0], 4)[:d] if d != 4 else np.tile([-1.0, 1.0], 2))
        for b in blocks:
            a[b:b + 32] = rough
    s = v.get_state().cpu().numpy()
    warm = v.get_info("foot_force").cpu().numpy() * dt
    tr_o = None
    for o, b in zip(o64, blocks):
        o.set_state(s[b:b + 64]); o.set_warm(warm[b:b + 64])
        if i == at_step and b <= at_env < b + 64:
            tr_o = o.set_trace(at_env - b)
    if i == at_step:
        v.set_trace(at_env)
        print("filtered action device", v.get_info("filtered_action").cpu().numpy()[at_env], "warm", warm[at_env])
    v.step_tensor(torch.from_numpy(a).to(v.device))
    for o, b in zip(o64, blocks):
        o.step(a[b:b + 64])
    if i == at_step:
        tv = v.get_trace(as_dict=False)
        for k in range(tv.shape[0]):
            dq, dqd = np.abs(tv[k, 14:26] - tr_o[k, 14:26]).max(), np.abs(tv[k, 26:38] - tr_o[k, 26:38]).max()
            print(f"substep {k}: max |dq| {dq:.2e} |dqd| {dqd:.2e} |dtau| {np.abs(tv[k, 38:50] - tr_o[k, 38:50]).max():.2e} |dspring| {np.abs(tv[k, 50:62] - tr_o[k, 50:62]).max():.2e} "
                  f"foot forces device {tv[k, 62:66]} oracle {tr_o[k, 62:66]}")
            print("   tau device", tv[k, 38:50]); print("   tau oracle", tr_o[k, 38:50])
            print("   spring device", tv[k, 50:62]); print("   spring oracle", tr_o[k, 50:62])
            print("   q device", tv[k, 14:26]); print("   q oracle", tr_o[k, 14:26])
