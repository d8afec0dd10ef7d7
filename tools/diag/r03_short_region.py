import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import torch
from qs_amd.vec_env import QuadrupedVecEnv
n = 8192
env = QuadrupedVecEnv(num_envs=n, auto_reset=True, seed=5, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", action_space_mode="SYMMETRIC",
                      motor_control_mode="PD", enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", info_fields=False)
env.reset_tensor()
acts = torch.rand((64, n, 6), device="cuda") * 2 - 1
for i in range(600): env.step_tensor(acts[i % 64])
torch.cuda.synchronize()
def region(K, timing, gap):
    for i in range(5): env.step_tensor(acts[i % 64])
    torch.cuda.synchronize()
    if gap: 
        for k in ("settle_substeps", "resets", "lookahead_served", "lookahead_settled", "reset_stalls", "limit_path_substeps", "self_narrow_substeps"): env.counter(k)
    if timing: env.enable_timing(True)
    t0 = time.perf_counter()
    for i in range(K): env.step_tensor(acts[i % 64])
    km = env.last_step_kernel_ms() if timing else 0.0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if timing: env.enable_timing(False)
    return el / K * 1e6, km * 1e3
for K in (20, 100, 1000):
    for timing in (0, 1):
        for gap in (0, 1):
            r = [region(K, timing, gap) for _ in range(5)]
            print(f"K {K:5d} timing {timing} counter-reads-before {gap}: us/step " + " ".join(f"{x[0]:.1f}" for x in r) + ("  kernel_us " + " ".join(f"{x[1]:.1f}" for x in r) if timing else ""))
