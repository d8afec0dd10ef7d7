#!/usr/bin/env python3
"""A configuration that tools/fuzz_parity.py dumped (QS_FUZZ_DUMP=dir): the deviating environment's step again with the per-substep trace
of kernel and oracle side by side -- where the two part, and by how much the oracle's own substeps move under a 1e-6 perturbation.
usage: python tools/diag/r03_fuzz_case.py dir/case55.pkl"""
import os
import pickle
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
from qs_amd.vec_env import QuadrupedVecEnv
from oracle.qso import Oracle

d = pickle.load(open(sys.argv[1], "rb"))
kw, n, s, a, e = d["kw"], d["n"], d["state"], d["action"], d["env"]
print("environment", e, {k: kw[k] for k in ("friction_model", "solver_residual_threshold", "payload", "env_randomizer_mode", "seed")})
v = QuadrupedVecEnv(num_envs=n, auto_reset=False, **kw)
o, o2 = Oracle(v.cfg, "f32"), Oracle(v.cfg, "f32")
v.reset(); o.reset(); o2.reset()
to, to2 = o.set_trace(e), o2.set_trace(e)
v.set_trace(e)
o.set_state(s); v.set_state(s.astype(np.float32))
rng = np.random.default_rng(0)
sp = s + 1e-6 * rng.standard_normal(s.shape) * np.maximum(np.abs(s), 1.0)
sp[:, 3:7] /= np.linalg.norm(sp[:, 3:7], axis=1, keepdims=True)
o2.set_state(sp)
o.step(a); o2.step(a); v.step(a)
tv = v.get_trace(as_dict=False)
names = [("pose", 1, 8), ("base velocity", 8, 14), ("q", 14, 26), ("qd", 26, 38), ("torque", 38, 50), ("foot force", 62, 66), ("foot contact", 66, 70)]
print("substep | " + " | ".join(f"{nm:>13s}" for nm, _, _ in names) + "   (max |kernel - oracle|, in brackets: |perturbed oracle - oracle|)")
for k in range(tv.shape[0]):
    print(f"{k:7d} | " + " | ".join(f"{np.abs(tv[k, lo:hi] - to[k, lo:hi]).max():.1e} [{np.abs(to2[k, lo:hi] - to[k, lo:hi]).max():.0e}]" for _, lo, hi in names))
print("foot contacts, kernel :", tv[:, 66:70].astype(int).tolist())
print("foot contacts, oracle :", to[:, 66:70].astype(int).tolist())
print("foot forces, kernel   :", np.round(tv[:, 62:66], 1).tolist())
print("foot forces, oracle   :", np.round(to[:, 62:66], 1).tolist())
JLO, JHI = np.tile([-1.0471975512, -0.663225115758, -2.72271363311], 4), np.tile([1.0471975512, 2.96705972839, -0.837758040957], 4)
print("distance of the nearest joint to its stop (negative: beyond it, a limit row exists), joint index:")
for name, t in (("kernel", tv), ("oracle", to), ("oracle, perturbed", to2)):
    dist = np.minimum(t[:, 14:26] - JLO, JHI - t[:, 14:26])
    print(f"  {name:18s}", [f"{dist[k].min():+.1e} (j{int(dist[k].argmin())})" for k in range(t.shape[0])])
