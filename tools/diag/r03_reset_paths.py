#!/usr/bin/env python3
"""Which reset paths produce the same bits, per friction model / payload model: k_reset (in place), the in-step settle of a handle without
look-ahead, the look-ahead slot filled at create (k_lookahead_fill), and the slots the settle lanes deliver."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np, torch
from qs_amd.vec_env import QuadrupedVecEnv

n = 64


def snap(v):
    return np.concatenate([v.get_state().cpu().numpy(), v.get_info("params").cpu().numpy()], axis=1)


def fall(v):
    s = v.get_state().cpu().numpy()
    s[:, 2] = 0.05; s[:, 3:7] = [0.7071, 0, 0, 0.7071]
    v.set_state(s)
    return v.step_tensor(torch.zeros((n, v.action_dim), device=v.device))[0].cpu().numpy()


def report(tag, x, y):
    d = np.abs(x - y)
    print(f"   {tag}: equal={np.array_equal(x, y)} max diff {d.max():.3e} differing entries {int((d > 0).sum())}")


for model in ("cone", "pyramid"):
    for payload in ("weld", "soft"):
        KW = dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True,
                  env_randomizer_mode="TEST_RANDOMIZER", seed=5, noise=False, friction_model=model, payload=payload, settle_steps=300)
        print(f"{model} / {payload}")
        a = QuadrupedVecEnv(num_envs=n, auto_reset=False, **KW)          # k_reset, episodes 0 .. 3
        ka = []
        for ep in range(4):
            a.reset_tensor(); ka.append(snap(a))
        a.close()
        b = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_lookahead=0, **KW)   # episode 0 by k_reset, 1.. by the in-step settle
        b.reset_tensor(); kb = [snap(b)]
        for ep in range(2):
            fall(b); kb.append(snap(b))
        b.close()
        c = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_lookahead=2, **KW)   # 0, 1 from slots filled at create; 2, 3 from the lanes
        c.reset_tensor(); kc = [snap(c)]
        fall(c); kc.append(snap(c))
        z = torch.zeros((n, c.action_dim), device=c.device)
        for ep in range(2):
            for _ in range(45):
                c.step_tensor(z)
            fall(c); kc.append(snap(c))
        print(f"   (look-ahead handle: {c.counter('reset_stalls')} stalls, {c.counter('lookahead_settled')} states from the lanes)")
        c.close()
        for ep in range(3):
            report(f"episode {ep}: k_reset vs in-step (K = 0 handle)", ka[ep], kb[ep])
        for ep in range(4):
            report(f"episode {ep}: k_reset vs look-ahead ({'fill' if ep < 2 else 'lanes'})", ka[ep], kc[ep])
