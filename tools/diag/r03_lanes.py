#!/usr/bin/env python3
"""Lanes vs k_reset under the friction pyramid: slicing or wave-mates?"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np, torch
from qs_amd.vec_env import QuadrupedVecEnv


def snap(v):
    return np.concatenate([v.get_state().cpu().numpy(), v.get_info("params").cpu().numpy()], axis=1)


def fall(v, n):
    s = v.get_state().cpu().numpy()
    s[:, 2] = 0.05; s[:, 3:7] = [0.7071, 0, 0, 0.7071]
    v.set_state(s)
    v.step_tensor(torch.zeros((n, v.action_dim), device=v.device))


for label, n, extra in (("pyramid settle 300", 64, dict(friction_model="pyramid", settle_steps=300)),
                        ("pyramid settle 300 resid 0", 64, dict(friction_model="pyramid", settle_steps=300, solver_residual_threshold=0.0)),
                        ("pyramid settle 10 (one slice)", 64, dict(friction_model="pyramid", settle_steps=10)),
                        ("pyramid settle 20 (two slices)", 64, dict(friction_model="pyramid", settle_steps=20)),
                        ("pyramid settle 300, 16 envs", 16, dict(friction_model="pyramid", settle_steps=300)),
                        ("pyramid settle 300, no randomizer", 64, dict(friction_model="pyramid", settle_steps=300, env_randomizer_mode="NONE")),
                        ("cone settle 300", 64, dict(settle_steps=300))):
    KW = dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True,
              env_randomizer_mode="TEST_RANDOMIZER", seed=5, noise=False)
    KW.update(extra)
    a = QuadrupedVecEnv(num_envs=n, auto_reset=False, **KW)
    ka = []
    for ep in range(4):
        a.reset_tensor(); ka.append(snap(a))
    a.close()
    c = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_lookahead=2, **KW)
    c.reset_tensor(); fall(c, n)
    z = torch.zeros((n, c.action_dim), device=c.device)
    res = []
    for ep in (2, 3):
        for _ in range(45):
            c.step_tensor(z)
        fall(c, n)
        k = snap(c)
        d = np.abs(k - ka[ep])
        res.append((float(d.max()), int((d > 0).any(1).sum())))
    print(f"{label}: lanes vs k_reset, episodes 2 and 3: max diff / environments differing {res}; stalls {c.counter('reset_stalls')}")
    c.close()
