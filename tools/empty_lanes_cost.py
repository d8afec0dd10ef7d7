#!/usr/bin/env python3
"""What the settle lanes' workgroups cost a launch when they have nothing to do: every launch of a look-ahead handle carries the workgroups
of five full cohorts (QS_COHORTS x slice / 16; slice = 2 N clamped to [2048, 131072]) and one beyond its cohort's jobs leaves at once.
Robots that stand still do not reset before the 1000-step time limit (the run stays below it), so all of them are empty: the step time with the lanes on against the lanes off is their price.
usage: python tools/empty_lanes_cost.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import torch
from qs_amd import QuadrupedVecEnv

rows = []
for n in (8192, 16384, 65536):
    env = QuadrupedVecEnv(num_envs=n, device=0, auto_reset=True, reset_lookahead=2, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                          enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1, info_fields=False)
    env.reset_tensor()
    still = torch.zeros((n, env.action_dim), device="cuda")
    res = {}
    for name, on in (("lanes_on", True), ("lanes_off", False), ("lanes_on_again", True)):
        env.settle_lanes(on)
        for _ in range(100):
            env.step_tensor(still)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            env.step_tensor(still)
        torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / 200 * 1e3
    slice_ = min(max(2 * n, 2048), 131072)
    res.update(n_envs=n, env_workgroups=n // 16, lane_workgroups=5 * slice_ // 16, resets=env.counter("resets"))
    res["empty_workgroups_cost_pct"] = 100 * (0.5 * (res["lanes_on"] + res["lanes_on_again"]) / res["lanes_off"] - 1)
    rows.append(res)
    print(res)
    env.close()
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
