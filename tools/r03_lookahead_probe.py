#!/usr/bin/env python3
"""Round-3 design probe.  RUNS ONLY AGAINST THE ROUND-2 LIBRARY (commit 5c69ebe, QS_LIB_PATH=<that libqs_hip.so>): reset_pool / pool_streaming are its API and are gone since; result: profiles/r03_lookahead_probe.json.
: what a per-environment look-ahead of reset states has to cope with.

1. episode lengths under the benchmark's U(-1,1) actions at N = 8192: how often do K consecutive episodes of one environment together
   last fewer steps than one settle takes (250 launches + the wait for a cohort, ~300) -- the stall rate of a K-deep look-ahead;
2. is a settle the same bits whichever code path computes it: k_reset (one loop of 2500 substeps), the pool fill, and the settle lanes of
   k_step (250 slices of 10 substeps through the step's own substep loop)."""
import json
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
import torch
from qs_amd.vec_env import QuadrupedVecEnv

KW = dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True,
          env_randomizer_mode="GROUND_RANDOMIZER", action_space_mode="SYMMETRIC", motor_control_mode="PD", noise=False)
out = {}


def episode_lengths(n=8192, steps=6000):
    v = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_pool=65536, seed=1234, info_fields=False, **KW)
    v.reset_tensor()
    v.pool_streaming(True)
    gen = torch.Generator(device=v.device).manual_seed(1234)
    acts = torch.rand((64, n, v.action_dim), generator=gen, device=v.device) * 2 - 1
    dones = torch.zeros((steps, n), dtype=torch.uint8, device=v.device)
    for t in range(steps):
        _, _, d, _ = v.step_tensor(acts[t % 64])
        dones[t] = d
    d = dones.cpu().numpy().astype(bool)
    v.close()
    lens = []            # per environment: the lengths of its complete episodes, in order
    for e in range(n):
        ends = np.nonzero(d[:, e])[0]
        lens.append(np.diff(np.concatenate(([-1], ends))))
    allv = np.concatenate(lens)
    res = {"episodes": int(len(allv)), "mean": float(allv.mean()), "quantiles_1_5_25_50_75": [float(x) for x in np.percentile(allv, [1, 5, 25, 50, 75])],
           "min": int(allv.min()), "share_below_50_100_150_300": [float((allv < x).mean()) for x in (50, 100, 150, 300)]}
    for K in (1, 2, 3, 4, 6, 8):
        stalls = total = 0
        for L in lens:
            if len(L) >= K:
                c = np.convolve(L, np.ones(K, dtype=np.int64), mode="valid")
                stalls += int((c < 300).sum()); total += len(c)
        res[f"K{K}_windows_below_300"] = [stalls, total]
    return res


def settle_paths():
    """pool entries are resets of the virtual environments 0x40000000 + p; a handle whose env_id_offset is that number resets the same
    (seed, id, episode) in place with k_reset."""
    P, base = 160, 0x40000000
    kw = dict(KW, seed=77)
    a = QuadrupedVecEnv(num_envs=P, auto_reset=False, env_id_offset=base, **kw)
    gens = {}
    for g in range(48):                        # episodes 0 .. 47 of every virtual environment (a streaming refill's episode = its generation)
        a.reset_tensor()
        gens[g] = np.concatenate([a.get_state().cpu().numpy(), a.get_info("params").cpu().numpy()], axis=1)
    a.close()
    res = {}
    b = QuadrupedVecEnv(num_envs=64, auto_reset=True, reset_pool=P, **kw)     # static pool: k_pool_fill, generation 0
    b.reset_tensor()
    s = b.get_state().cpu().numpy()
    s[:, 2] = 0.05; s[:, 3:7] = [0.7071, 0, 0, 0.7071]
    b.set_state(s)
    b.step_tensor(torch.zeros((64, b.action_dim), device=b.device))
    got = np.concatenate([b.get_state().cpu().numpy(), b.get_info("params").cpu().numpy()], axis=1)
    keys = {r.tobytes() for r in gens[0]}
    res["pool_fill_vs_k_reset_bitwise"] = [int(sum(r.tobytes() in keys for r in got)), len(got)]
    # streaming: consume, let the lanes re-settle (generations >= 1), consume again
    b.pool_streaming(True)
    z = torch.zeros((64, b.action_dim), device=b.device)
    hits = tot = 0
    keysets = {g: {x.tobytes() for x in gens[g]} for g in gens}
    seen_gen = set()
    for rnd in range(6):
        for _ in range(330):
            b.step_tensor(z)
        b.set_state(s)
        b.step_tensor(z)
        got = np.concatenate([b.get_state().cpu().numpy(), b.get_info("params").cpu().numpy()], axis=1)
        for r in got:
            tot += 1
            for g in gens:
                if r.tobytes() in keysets[g]:
                    hits += 1; seen_gen.add(g)
                    break
    res["settle_lanes_vs_k_reset_bitwise"] = [hits, tot, sorted(seen_gen)]
    b.close()
    return res


out["settle_paths"] = settle_paths()
print(json.dumps(out["settle_paths"]))
out["episode_lengths"] = episode_lengths()
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r03_lookahead_probe.json", "w"), indent=1)
