#!/usr/bin/env python3
"""Does the deviating environment of a dumped fuzz configuration (QS_FUZZ_DUMP, tools/fuzz_parity.py) depend on its wave-mates?  Its step
again with every other environment (a) as dumped, (b) standing still -- bitwise comparison of its state afterwards, and both against the oracle.
usage: python tools/diag/r03_fuzz_case_mates.py dir/case55.pkl"""
import os
import pickle
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
from qs_amd.vec_env import QuadrupedVecEnv
from oracle.qso import Oracle

d = pickle.load(open(sys.argv[1], "rb"))
kw, n, s, a, e = d["kw"], d["n"], d["state"], d["action"], d["env"]
o = Oracle(QuadrupedVecEnv(num_envs=n, auto_reset=False, **kw).cfg, "f32")
o.reset(); o.set_state(s); o.step(a); so = o.get_state()[e]
out = {}
for name in ("as dumped", "mates standing", "mates = copies of it"):
    v = QuadrupedVecEnv(num_envs=n, auto_reset=False, **kw)
    v.reset()
    st = v.get_state().cpu().numpy()          # settled, standing
    act = np.zeros_like(a)
    if name == "as dumped":
        st, act = s.astype(np.float32).copy(), a.copy()
    elif name == "mates = copies of it":
        st[:] = s[e].astype(np.float32); act[:] = a[e]
    st[e] = s[e].astype(np.float32); act[e] = a[e]
    v.set_state(st); v.step(act)
    out[name] = v.get_state().cpu().numpy()[e]
    print(f"{name:22s}: base velocity {np.round(out[name][7:13], 4).tolist()}   max |kernel - oracle| over the state: {np.abs(out[name] - so).max():.2e}")
    v.close()
print("oracle                : base velocity", np.round(so[7:13], 4).tolist())
print("bitwise equal, as dumped vs mates standing:", np.array_equal(out["as dumped"], out["mates standing"]))
# ... and is the kernel itself at a discontinuity there?  Its own result from states 1e-6 away
rng = np.random.default_rng(1)
v = QuadrupedVecEnv(num_envs=n, auto_reset=False, **kw)
v.reset()
for trial in range(8):
    sp = s + 1e-6 * rng.standard_normal(s.shape) * np.maximum(np.abs(s), 1.0)
    sp[:, 3:7] /= np.linalg.norm(sp[:, 3:7], axis=1, keepdims=True)
    v.set_state(sp.astype(np.float32)); v.step(a)
    r = v.get_state().cpu().numpy()[e]
    o.set_state(sp); o.step(a); ro = o.get_state()[e]
    print(f"state + 1e-6 noise #{trial}: kernel moves by {np.abs(r - out['as dumped']).max():.2e} (base velocity {np.abs(r[7:13] - out['as dumped'][7:13]).max():.2e}), "
          f"oracle moves by {np.abs(ro - so).max():.2e}, kernel - oracle {np.abs(r - ro).max():.2e}")
