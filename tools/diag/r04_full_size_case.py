#!/usr/bin/env python3
"""The deviations tests/test_gpu_parity.py::test_full_size_oracle_sampled found at first (config2_4096 step 51, config3_8192 step 75): the same run
with the oracle's float32 build next to its float64 build, printing for every environment beyond the tolerances what the three say.
usage: python tools/diag/r04_full_size_case.py config2_4096"""
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
from test_gpu_parity import FULL_SIZE, TOL_POS, TOL_BASE_V, TOL_Q, TOL_QD
from oracle.qso import Oracle
from qs_amd.config import build_config
from qs_amd.vec_env import QuadrupedVecEnv

name = sys.argv[1]
n, kw = FULL_SIZE[name]
kw = dict(dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True), **kw, seed=7, noise=False)
v = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_lookahead=16, **kw)
rng = np.random.default_rng(sum(map(ord, name)))
blocks = [int(b) * 64 for b in sorted(rng.choice(n // 64, size=4, replace=False))]
o64 = [Oracle(build_config(n_envs=64, auto_reset=True, env_id_offset=b, **kw)[0]) for b in blocks]
o32 = [Oracle(build_config(n_envs=64, auto_reset=True, env_id_offset=b, **kw)[0], "f32") for b in blocks]
d, dt = v.action_dim, float(v.cfg.dt)
v.reset_tensor()
for o in o64 + o32:
    o.reset()
for i in range(100):
    a = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    if i % 20 > 8:
        rough = [1.0, 1.0, 1.0, 1.0, -1.0] if d == 5 else (np.tile([0.0, -1.0, 1.0], 4)[:d] if d != 4 else np.tile([-1.0, 1.0], 2))
        for b in blocks:
            a[b:b + 32] = rough
    s = v.get_state().cpu().numpy()
    warm = v.get_info("foot_force").cpu().numpy() * dt
    for o, p, b in zip(o64, o32, blocks):
        o.set_state(s[b:b + 64]); o.set_warm(warm[b:b + 64]); p.set_state(s[b:b + 64]); p.set_warm(warm[b:b + 64])
    vo, rv, dv, tv = (x.cpu().numpy() for x in v.step_tensor(torch.from_numpy(a).to(v.device)))
    sv = v.get_state().cpu().numpy()
    cf = v.get_info("foot_contact").cpu().numpy()
    for o, p, b in zip(o64, o32, blocks):
        _, _, do, _ = o.step(a[b:b + 64]); _, _, dp, _ = p.step(a[b:b + 64])
        so, sp = o.get_state(), p.get_state().astype(np.float64)
        tol = np.concatenate([np.full(7, TOL_POS), np.full(6, TOL_BASE_V), np.full(12, TOL_Q), np.full(12, TOL_QD)])
        for e in np.nonzero(~do)[0]:
            dev = np.abs(sv[b + e] - so[e]) / tol
            if dev.max() > 1.0:
                k = int(np.argmax(dev))
                print(f"step {i} env {b + e} element {k}: device {sv[b + e, k]:.8f} oracle64 {so[e, k]:.8f} oracle32 {sp[e, k]:.8f}  "
                      f"|dev - o64| = {abs(sv[b + e, k] - so[e, k]):.2e} ({dev.max():.1f} x tol), |o32 - o64| = {abs(sp[e, k] - so[e, k]):.2e}, "
                      f"max over elements |o32 - o64| / tol = {(np.abs(sp[e] - so[e]) / tol).max():.1f}; contact flags device {cf[b + e]} oracle64 {o.get_info(1)[e]} oracle32 {p.get_info(1)[e]}; "
                      f"z {s[b + e, 2]:.4f}, joint limits hit: {bool((np.abs(s[b + e, 13:25:3]) > 1.04).any() or (s[b + e, 15:25:3] < -2.71).any() or (s[b + e, 14:25:3] < -0.66).any())}")
                if dev.max() > 5.0:
                    for nm, wd, wo in (("torque", "torque", 2), ("spring torque", "spring_torque", 3), ("params", "params", 6), ("counters", "counters", 7), ("last action", "last_action", 8), ("task", "task", 4)):
                        x, y = v.get_info(wd).cpu().numpy()[b + e], o.get_info(wo)[e]
                        m = min(len(x), len(y))
                        print(f"    {nm}: max |device - oracle64| = {np.abs(x[:m] - y[:m]).max():.3e}   device {np.round(x[:m], 4).tolist()[:14]}  oracle {np.round(y[:m], 4).tolist()[:14]}")
                    print(f"    action {a[b + e].tolist()}  state before {np.round(s[b + e], 5).tolist()}")
print("done")
