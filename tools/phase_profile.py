#!/usr/bin/env python3
"""Cycles per phase of the rigid-body substep, measured with s_memtime in workgroup 0 of a -DQS_PROFILE_PHASES build:

    QS_HIPCC_EXTRA=-DQS_PROFILE_PHASES QS_BUILD_OUT=quadruped-springs_amd/qs_amd/exp/prof.so python quadruped-springs_amd/build.py --force
    QS_LIB_PATH=$PWD/quadruped-springs_amd/qs_amd/exp/prof.so python tools/phase_profile.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import torch
from qs_amd.vec_env import QuadrupedVecEnv

NAMES = {1: "base rotation, velocities", 2: "leg kinematics", 3: "link inertias", 4: "RNEA bias", 5: "CRBA (B, D, K)", 6: "Schur + Cholesky",
         7: "accelerations, v*", 8: "collision", 9: "contact rows + Delassus", 10: "PGS sweeps", 11: "delta v", 12: "integrate positions"}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
env = QuadrupedVecEnv(num_envs=N, auto_reset=True, reset_lookahead=8, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                      enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1)
env.reset_tensor()
if os.environ.get("QS_PP_LANES") == "0":
    env.settle_lanes(False)      # (round 3: with the lanes, substeps 0-5 of a step take 16.7 k cycles and substep 6 -- when the settle waves leave -- 19.1 k; without them all ten take 17.0 k; the sums agree)
a = torch.rand((16, N, 6), device="cuda") * 2 - 1
for i in range(20):
    env.step_tensor(a[i % 16])
out = (C.c_uint64 * 48)()
env.lib.qs_debug_phases(out, 1)
n = 200
for i in range(n):
    env.step_tensor(a[i % 16])
env.lib.qs_debug_phases(out, 0)
tot = sum(out[:16])
print(f"{'phase':32s} cycles/substep   share")
for k in range(1, 13):
    print(f"{NAMES[k]:32s} {out[k] / (n * 10):12.0f}   {100 * out[k] / tot:5.1f} %")
print(f"{'substeps total':32s} {sum(out[1:13]) / (n * 10):12.0f}")
print(f"per env-step: tile load {out[13] / n:.0f}, E::step outside the substeps {(out[14] - 0) / n:.0f} (phase 14 = after the last substep marker .. end of E::step, "
      f"phase 1 also absorbs the action prologue of the first substep), auto-reset + stores {out[15] / n:.0f} cycles")
print("cycles of substep k of the env step (k = 0 pays the cold instruction cache): " + ", ".join(f"{out[16 + k] / n:.0f}" for k in range(10)))
print("kernel entry: config / kernarg reads %.0f, issue of the tile loads %.0f, action loads %.0f, wait + barrier %.0f, rest of the prologue %.0f cycles per env-step"
      % tuple(out[k] / n for k in (26, 27, 28, 29, 13)))
print("epilogue of E::step: load task %.0f, task_on_step %.0f, reward + termination %.0f, wrapper machine %.0f, store state / task %.0f, "
      "observations + noise %.0f cycles per env-step" % tuple(out[k] / n for k in (33, 34, 35, 36, 37, 38)))
