#!/usr/bin/env python3
"""Twins in a wave with / without a fallen neighbour (tests/test_gpu_round2.py::test_results_do_not_depend_on_wave_mates): which entries differ."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from test_gpu_round2 import vec_env, RAW, fallen_states
for variant in (dict(friction_model="pyramid", solver_residual_threshold=0.0), dict(friction_model="pyramid", solver_residual_threshold=0.0, body_contacts=False), dict(solver_residual_threshold=0.0)):
    for neighbour in ("fallen", "joint_limit"):
        n = 32
        v = vec_env(n, **dict(RAW, **variant)); v.reset()
        rng = np.random.default_rng(11)
        s = v.get_state().cpu().numpy(); s[16:] = s[:16]
        s[:, 13:25] += np.tile(rng.uniform(-0.1, 0.1, size=(16, 12)), (2, 1)).astype(np.float32); s[16:] = s[:16]
        odd = 5
        if neighbour == "fallen":
            s[odd] = fallen_states(s[odd:odd + 1], rng)[0]; s[odd, 2] = 0.16
        else:
            s[odd, 13 + 2] = -2.76
        v.set_state(s)
        tau = np.tile(rng.uniform(-4, 4, size=(16, 12)), (2, 1)).astype(np.float32)
        v.step(tau)
        st = v.get_state().cpu().numpy()
        tw = np.array([k for k in range(16) if k != odd])
        d = np.abs(st[tw] - st[tw + 16])
        ff = v.get_info("foot_force").cpu().numpy(); fc = v.get_info("foot_contact").cpu().numpy()
        print(variant, neighbour, "max diff", d.max(), "envs differing", tw[(d > 0).any(1)].tolist(), "columns", np.nonzero((d > 0).any(0))[0].tolist()[:12],
              "contacts of twins", fc[tw].sum(1).tolist())
        v.close()
