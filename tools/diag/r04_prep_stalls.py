import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import torch
from qs_amd.vec_env import QuadrupedVecEnv
n = 8192
env = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_lookahead=16, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True,
                      enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1234, info_fields=False)
env.reset_tensor()
g = torch.Generator(device="cuda").manual_seed(1234)
acts = torch.rand((64, n, 6), generator=g, device="cuda") * 2 - 1
ids = torch.arange(n, device="cuda")
groups = 125
print("after first reset: stalls", env.counter("reset_stalls"), "served", env.counter("lookahead_served"), "resets", env.counter("resets"))
for gidx in range(groups):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.reset_tensor((ids % groups == gidx).to(torch.uint8))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for i in range(1000 // groups):
        env.step_tensor(acts[i % 64])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if gidx < 6 or gidx % 25 == 0 or (t2 - t1) > 0.005 or (t1 - t0) > 0.005:
        print(f"group {gidx}: reset {1e3 * (t1 - t0):.2f} ms, 8 steps {1e3 * (t2 - t1):.2f} ms, stalls {env.counter('reset_stalls')} served {env.counter('lookahead_served')} resets {env.counter('resets')}")
print("end: stalls", env.counter("reset_stalls"), "served", env.counter("lookahead_served"), "resets", env.counter("resets"))
