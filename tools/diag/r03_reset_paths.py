#!/usr/bin/env python3
"""Which reset paths produce the same bits: k_reset (in place), the in-step settle of a handle without look-ahead, the look-ahead slot
filled at create (k_lookahead_fill), and the slots the settle lanes deliver."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np, torch
from qs_amd.vec_env import QuadrupedVecEnv

KW = dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True,
          env_randomizer_mode="GROUND_RANDOMIZER", seed=5, noise=False)
n = 64
names = ["pos"] * 3 + ["quat"] * 4 + ["vlin"] * 3 + ["vang"] * 3 + ["q"] * 12 + ["qd"] * 12


def snap(v):
    return np.concatenate([v.get_state().cpu().numpy(), v.get_info("params").cpu().numpy(), v.get_info("foot_force").cpu().numpy(),
                           v.get_info("torque").cpu().numpy()], axis=1)


def fall(v):
    s = v.get_state().cpu().numpy()
    s[:, 2] = 0.05; s[:, 3:7] = [0.7071, 0, 0, 0.7071]
    v.set_state(s)
    return v.step_tensor(torch.zeros((n, v.action_dim), device=v.device))[0].cpu().numpy()


def report(tag, x, y):
    d = np.abs(x - y)
    bad = np.argwhere(d > 0)
    print(f"{tag}: equal={np.array_equal(x, y)} max diff {d.max():.3e} differing entries {len(bad)}; first: {[(int(i), int(j), float(x[i, j]), float(y[i, j])) for i, j in bad[:6]]}")


res = {}
for ep in range(3):
    pass
a = QuadrupedVecEnv(num_envs=n, auto_reset=False, **KW)          # k_reset, episodes 0, 1, 2
ka = []
for ep in range(3):
    a.reset_tensor(); ka.append(snap(a))
b = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_lookahead=0, **KW)   # episode 0 by k_reset, 1 and 2 by the in-step settle
b.reset_tensor(); kb = [snap(b)]
ob1 = fall(b); kb.append(snap(b)); ob2 = fall(b); kb.append(snap(b))
c = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_lookahead=2, **KW)   # all from slots filled at create
c.settle_lanes(False)
c.reset_tensor(); kc = [snap(c)]
oc1 = fall(c); kc.append(snap(c))
for ep in range(2):
    report(f"episode {ep}: k_reset vs in-step/k_reset(K=0 handle)", ka[ep], kb[ep])
    report(f"episode {ep}: k_reset vs look-ahead slot", ka[ep], kc[ep])
report("episode 2: k_reset vs in-step", ka[2], kb[2])
report("obs after reset to episode 1: in-step vs slot", ob1, oc1)
print("stalls c:", c.counter("reset_stalls"), "served", c.counter("lookahead_served"))
