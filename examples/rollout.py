#!/usr/bin/env python3
"""What quadruped_spring/load_model.py:109-137 does with one PyBullet environment, on 8192 environments at once.

    python examples/rollout.py [--envs 8192] [--steps 1000] [--wrapper LANDING]

`policy` stands in for `model.predict`: any callable from an observation batch to an action batch.  With --device-policy the
loop never leaves the GPU (step_tensor); without it the SB3 numpy convention is used (VecEnv.step, infos with
"terminal_observation" / "TimeLimit.truncated")."""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "quadruped-springs_amd"))

import numpy as np
import torch

from qs_amd import DeviceVecNormalize, QuadrupedVecEnv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--wrapper", default=None, help="LANDING | LANDING2 | LANDING_BACKFLIP | LANDING_CONTINUOUS | GO_TO_REST")
    ap.add_argument("--device-policy", action="store_true")
    args = ap.parse_args()
    env_kwargs = dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", action_space_mode="SYMMETRIC", motor_control_mode="PD",
                      enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER")   # an args.yml of the reference
    venv = QuadrupedVecEnv(num_envs=args.envs, device=0, auto_reset=True, wrapper=args.wrapper, **env_kwargs)
    env = DeviceVecNormalize(venv, training=True, norm_reward=True)   # VecNormalize.load(stats_path, env) in the reference
    rng = np.random.default_rng(0)
    returns, lengths, ep_ret, ep_len = [], [], np.zeros(args.envs), np.zeros(args.envs, int)
    if args.device_policy:
        policy = lambda obs: torch.tanh(obs[:, :6] * 3.0)       # stand-in for a network living on the GPU
        obs = env.reset_tensor()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            obs, rew, done, trunc = env.step_tensor(policy(obs))
        torch.cuda.synchronize()
    else:
        policy = lambda obs: np.tanh(obs[:, :6] * 3.0) + 0.3 * rng.standard_normal((len(obs), 6))
        obs = env.reset()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            obs, rew, done, infos = env.step(policy(obs))
            ep_ret += env.get_original_reward(); ep_len += 1
            for i in np.nonzero(done)[0]:
                returns.append(ep_ret[i]); lengths.append(ep_len[i]); ep_ret[i] = 0; ep_len[i] = 0
    dt = time.perf_counter() - t0
    print(f"{args.envs * args.steps / dt / 1e6:.1f} M env-steps/s over {args.steps} steps of {args.envs} environments")
    if returns:
        print(f"{len(returns)} episodes, mean return {np.mean(returns):.3f}, mean length {np.mean(lengths):.0f} steps")
    s = env.get_stats()
    print("obs_rms.mean[:4] =", np.round(s["obs_mean"][:4], 4), " count =", s["obs_count"])
    env.close()


if __name__ == "__main__":
    main()
