#!/usr/bin/env python3
"""Does a reset's settle reach a fixed point of the substep map before its 2500 substeps are over?  Reset states of the same environments
(same seed: same parameter draws, same spawn) under settle_steps = 500 ... 2500, compared bit for bit with the 2500-substep ones.
usage: python tools/settle_fixed_point.py"""
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import torch
from qs_amd import QuadrupedVecEnv

N = 2048
for rand in ("GROUND_RANDOMIZER", "TEST_RANDOMIZER"):
    ref = None
    for steps in (2500, 2400, 2000, 1500, 1000, 750, 500):
        env = QuadrupedVecEnv(num_envs=N, device=0, auto_reset=True, reset_lookahead=0, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                              enable_springs=True, enable_action_filter=True, env_randomizer_mode=rand, seed=5, settle_steps=steps, noise=False)
        env.reset_tensor()
        st = env.get_state().clone()
        env.close()
        if ref is None:
            ref = st
            continue
        same = (st == ref).all(dim=1)
        d = (st - ref).abs()
        print(f"{rand}: settle_steps {steps}: {int(same.sum())} of {N} reset states bitwise those of 2500 substeps; max |difference| {float(d.max()):.3e} (q {float(d[:, 13:25].max()):.2e}, pose {float(d[:, :7].max()):.2e}, velocities {float(d[:, 7:13].max()):.2e} / {float(d[:, 25:].max()):.2e})")
