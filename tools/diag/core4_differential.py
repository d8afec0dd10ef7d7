"""Runs a library built with -DQS_DBG_CORE4 (QS_LIB_PATH): the many-rows solver's core<0, 4> and core<0, 6> on the same rows of every small
solve; bits 40.. of the self-narrow counter = sum of the contact points of the solves whose impulses differ in any bit (0 = none differ; the low
bits also count the real self-narrow events).
    QS_HIPCC_EXTRA=-DQS_DBG_CORE4 QS_BUILD_OUT=$PWD/tools/bin/dbg4.so python quadruped-springs_amd/build.py --force
    QS_LIB_PATH=$PWD/tools/bin/dbg4.so python tools/diag/core4_differential.py"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np, torch
from scipy.spatial.transform import Rotation as Rot
from qs_amd.vec_env import QuadrupedVecEnv
n = 2048
for fm in ("cone", "pyramid"):
  for thr in (0.0, 1e-7):
    env = QuadrupedVecEnv(num_envs=n, auto_reset=False, task_env="NO_TASK", observation_space_mode="ENCODER", enable_springs=True, enable_action_filter=False,
                      isRLGymInterface=False, motor_control_mode="TORQUE", env_randomizer_mode="NONE", noise=False, friction_model=fm, solver_residual_threshold=thr)
    env.reset_tensor()
    s = env.get_state().clone()
    idx = torch.arange(0, n, 4, device=s.device)
    s[idx, 2] = 0.12
    s[idx, 3:7] = torch.tensor(Rot.from_euler("x", 1.45).as_quat(), dtype=torch.float32, device=s.device)
    s[idx, 13:25] = torch.tensor(np.tile([0.0, 1.2, -2.4], 4), dtype=torch.float32, device=s.device)
    env.set_state(s)
    g = torch.Generator(device="cuda").manual_seed(0)
    for k in range(60):
        env.step_tensor((torch.rand((n, 12), generator=g, device="cuda") - 0.5) * 4)
    c = env.counter("self_narrow_substeps")
    print(fm, thr, "rare wave-substeps", env.counter("limit_path_substeps"), "mismatching small solves", c & 0xFFFFF, "sweep-count mismatches", (c >> 20) & 0xFFFFF, "sum mB", c >> 40, flush=True)
    env.close()
