#!/usr/bin/env python3
"""The imitation (DEMO) tasks end to end: record a demonstration, run the reference's own smoke test on it, then imitate it in
4096 environments with reference-state initialisation.

    python examples/imitation.py [--envs 4096] [--steps 300]

1. get_demonstration_wrapper.py's job: one environment under JUMPING_IN_PLACE makes a scripted jump, `demo_rows()` after every
   step gives the rows the reference would np.save (and that its repository does not hold).
2. quadruped_gym_env.py:439-476 (`test_env`, the reference's only smoke test): JUMPING_IN_PLACE_DEMO, springs, action filter,
   PPO_BASIC, random actions until the episode ends.
3. The same task on N environments under ReferenceStateInitVecEnv: a noisy copy of the demonstration's actions as the "policy"."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "quadruped-springs_amd"))

import numpy as np
import torch

from qs_amd import QuadrupedGymEnv, QuadrupedVecEnv, ReferenceStateInitVecEnv

KW = dict(motor_control_mode="PD", action_repeat=10, enable_springs=True, enable_action_filter=True, observation_space_mode="PPO_BASIC",
          action_space_mode="SYMMETRIC", env_randomizer_mode="GROUND_RANDOMIZER")


def record_demonstration(steps=110):
    venv = QuadrupedVecEnv(num_envs=1, auto_reset=False, task_env="JUMPING_IN_PLACE", noise=False, **KW)
    venv.reset()
    rows = []
    for t in range(steps):
        a = np.array([0.0, 0.9, -0.9] if t < 45 else [0.0, -0.5, 0.6] if t < 54 else [0.0, 0.1, 0.2], np.float32)
        _, _, done, _ = venv.step(np.tile(a, 2)[None])
        rows.append(venv.demo_rows(done)[0].cpu().numpy())
        if done[0]:
            break
    venv.close()
    return np.array(rows[:-1])      # save_demo drops the last row (get_demonstration_wrapper.py:30)


def reference_smoke(demo):
    env = QuadrupedGymEnv(task_env="JUMPING_IN_PLACE_DEMO", curriculum_level=1.0, demo=demo, **KW)
    env.reset()
    rew = 0.0
    for i in range(1500):
        _, reward, done, _ = env.step(np.random.rand(env.action_dim) * 2 - 1)
        rew += reward
        if done:
            break
    print(f"reference smoke test: episode of {i + 1} steps, rew: {rew:.4f}")
    env.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    demo = record_demonstration()
    z = demo[:, 6 + 24 + 2]
    print(f"demonstration: {demo.shape[0]} rows of {demo.shape[1]}, base height {z.min():.3f} .. {z.max():.3f} m")
    reference_smoke(demo)
    venv = ReferenceStateInitVecEnv(QuadrupedVecEnv(num_envs=args.envs, auto_reset=False, task_env="JUMPING_IN_PLACE_DEMO", demo=demo, **KW), seed=0)
    table = torch.as_tensor(demo[:, :6], device=venv.device)
    venv.reset_tensor()
    ret, returns = torch.zeros(args.envs, device=venv.device), []
    for _ in range(args.steps):
        a = table[venv.demo_counter().clamp(max=len(demo) - 1)] + 0.1 * torch.randn((args.envs, 6), device=venv.device)
        _, rew, done, _ = venv.step_tensor(a)
        ret += rew
        d = done.bool()
        returns += ret[d].tolist()
        ret[d] = 0
    print(f"{len(returns)} episodes in {args.steps} steps of {args.envs} environments, mean imitation return {np.mean(returns):.3f} "
          f"(1.0 = the demonstration's actions exactly)")
    venv.close()


if __name__ == "__main__":
    main()
