#!/usr/bin/env python3
"""Cycles per phase of a substep on the MANY-ROWS path, measured with s_memtime in workgroup 0 of a -DQS_PROFILE_PHASES build whose first
environment lies on its side (tools/time_rare_path.py's "one robot in 64" scenario: NO_TASK, raw torques, body contacts on):

    QS_HIPCC_EXTRA=-DQS_PROFILE_PHASES QS_BUILD_OUT=$PWD/quadruped-springs_amd/qs_amd/exp/prof.so python quadruped-springs_amd/build.py --force
    QS_LIB_PATH=$PWD/quadruped-springs_amd/qs_amd/exp/prof.so python tools/phase_profile_rare.py [every]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
import torch
from scipy.spatial.transform import Rotation as Rot
from qs_amd.vec_env import QuadrupedVecEnv

NAMES = {1: "base rotation, velocities", 2: "leg kinematics", 3: "link inertias", 4: "RNEA bias", 5: "CRBA (B, D, K)", 6: "Schur + Cholesky",
         7: "accelerations", 8: "collision (+ link-link tests), v*, foot rows", 39: "payload rows",
         40: "support-point and limit rows", 41: "rows -> LDS -> lanes", 42: "Delassus columns", 43: "wave-wide sweeps", 44: "impulses -> quad",
         45: "delta v (many rows)", 9: "Delassus block (common-path solve)", 10: "its sweeps", 11: "its delta v, select", 12: "integrate positions"}
n = 8192
if len(sys.argv) > 1 and sys.argv[1] == "--headline":
    # the benchmark's workload with the links' contact response on: how many many-rows solves a step holds, and how long they are
    env = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_lookahead=16, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                          enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1, body_contacts=True)
    env.reset_tensor()
    a = torch.rand((64, n, 6), device="cuda") * 2 - 1
    for i in range(600):
        env.step_tensor(a[i % 64])
    out = (C.c_uint64 * 48)()
    env.lib.qs_debug_phases(out, 1)
    steps = 400
    for i in range(steps):
        env.step_tensor(a[i % 64])
    env.lib.qs_debug_phases(out, 0)
    print(f"benchmark workload, body_contacts=True, {steps} steps: {out[0] / steps:.1f} many-rows solves per step, {out[46] / max(out[0], 1):.1f} sweeps, "
          f"{out[47] / max(out[0], 1):.1f} live rows, {out[30] / max(out[0], 1):.1f} contact points per solve")
    sys.exit(0)
every = int(sys.argv[1]) if len(sys.argv) > 1 else 64
TQ = float(os.environ.get("QS_PP_TORQUE", "4"))     # peak-to-peak of the random joint torques (0: the robot lies still -- no joint reaches its stop, the solves take the small instantiations)
env = QuadrupedVecEnv(num_envs=n, auto_reset=False, task_env="NO_TASK", observation_space_mode="ENCODER", enable_springs=True, enable_action_filter=False,
                      isRLGymInterface=False, motor_control_mode="TORQUE", env_randomizer_mode="NONE", noise=False)
env.reset_tensor()
s = env.get_state().clone()
idx = torch.arange(0, n, every, device=s.device)
s[idx, 2] = 0.12
s[idx, 3:7] = torch.tensor(Rot.from_euler("x", 1.45).as_quat(), dtype=torch.float32, device=s.device)
s[idx, 13:25] = torch.tensor(np.tile([0.0, 1.2, -2.4], 4), dtype=torch.float32, device=s.device)
env.set_state(s)
g = torch.Generator(device="cuda").manual_seed(0)
for _ in range(20):
    env.step_tensor((torch.rand((n, 12), generator=g, device="cuda") - 0.5) * TQ)
out = (C.c_uint64 * 48)()
env.lib.qs_debug_phases(out, 1)
c0 = env.counter("limit_path_substeps")
steps = 50
for _ in range(steps):
    env.step_tensor((torch.rand((n, 12), generator=g, device="cuda") - 0.5) * TQ)
env.lib.qs_debug_phases(out, 0)
print(f"workgroup 0 (environment 0 on its side, one robot in {every}), {steps} env steps x 10 substeps; many-rows wave-substeps per step: {(env.counter('limit_path_substeps') - c0) / steps:.0f}")
tot = sum(out[k] for k in NAMES)
print(f"{'phase':44s} cycles/substep   share")
for k in NAMES:
    print(f"{NAMES[k]:44s} {out[k] / (steps * 10):12.0f}   {100 * out[k] / tot:5.1f} %")
print(f"{'substep total':44s} {tot / (steps * 10):12.0f}")
print("cycles of substep k of the env step: " + ", ".join(f"{out[16 + k] / steps:.0f}" for k in range(10)))
if out[0]:
    print(f"all workgroups: {out[0]} many-rows solves, {out[46] / out[0]:.1f} sweeps, {out[47] / out[0]:.1f} live rows, {out[30] / out[0]:.1f} contact points per solve")
