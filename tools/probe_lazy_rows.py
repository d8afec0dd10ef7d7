#!/usr/bin/env python3
"""How many environment-substeps with a support point in range need the many-rows solve?  With body_contacts=True a non-foot link within its
contact range (4 mm) is a contact candidate; while it is still approaching, its rows are speculative and end every sweep at zero impulse.
The step builds such a point's rows only once its normal row can act (qs_core.h, SUPPORT_MARGIN).  A -DQS_PROBE_LAZY build of the library
(QS_HIPCC_EXTRA=-DQS_PROBE_LAZY QS_BUILD_OUT=<probe.so> python quadruped-springs_amd/build.py --force) builds every point's rows as before
the rule and counts what the rule would have done at four margins, and where it would have been wrong.
usage: QS_LIB_PATH=<probe.so> python tools/probe_lazy_rows.py"""
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import torch
from qs_amd import QuadrupedVecEnv

N = 8192
env = QuadrupedVecEnv(num_envs=N, device=0, auto_reset=True, reset_lookahead=16, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                      enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1234, info_fields=False, body_contacts=True)
env.reset_tensor()
g = torch.Generator(device="cuda").manual_seed(1234)
acts = torch.rand((64, N, 6), generator=g, device="cuda") * 2 - 1
for i in range(1500):
    env.step_tensor(acts[i % 64])
import ctypes as C
import numpy as np
def probe():
    out = (C.c_uint64 * 10)()
    assert env.lib.qs_probe_counters(env.h, out) == 0
    return np.array(list(out), dtype=np.int64)
steps = 400
p0, r0 = probe(), env.counter("resets")
for i in range(steps):
    env.step_tensor(acts[i % 64])
p1, r1 = probe(), env.counter("resets")
d = p1 - p0
print(f"{steps} steps of the benchmark workload with body_contacts=True ({(r1 - r0) / steps:.1f} resets per step): {d[0] / steps:.1f} environment-substeps with rows of their own on the "
      f"many-rows path per step (a support point in range), of which {d[1] / steps:.1f} ({100 * d[1] / max(d[0], 1):.0f} %) ended with every support-point row at zero impulse and no joint at its stop")
for m, margin in enumerate((0.0, 0.25, 0.5, 1.0)):
    print(f"   rule 'rows only if the normal row starts within {margin:.2f} m/s of acting': keeps {d[2 + 2 * m] / steps:.2f} per step ({100 * d[2 + 2 * m] / max(d[0], 1):.0f} %), "
          f"drops a row that ended with an impulse in {d[3 + 2 * m]} of {d[0]} ({100 * d[3 + 2 * m] / max(d[0], 1):.2f} %)")
env.close()
