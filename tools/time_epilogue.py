import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import torch
from qs_amd.vec_env import QuadrupedVecEnv
def run(tag, **kw):
    env = QuadrupedVecEnv(num_envs=8192, auto_reset=True, reset_lookahead=8, enable_springs=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1, **kw)
    env.reset_tensor()
    a = torch.rand((16, 8192, env.action_dim), device="cuda") * 2 - 1
    for i in range(30): env.step_tensor(a[i % 16])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(400): env.step_tensor(a[i % 16])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 400
    print(f"{tag:60s} {dt*1e3:.4f} ms/step")
    env.close()
run("default (JUMPING_IN_PLACE, PPO_BASIC, filter, noise)", task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_action_filter=True)
run("noise off", task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_action_filter=True, noise=False)
run("noise off, filter off", task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_action_filter=False, noise=False)
run("NO_TASK, ENCODER, noise off, filter off", task_env="NO_TASK", observation_space_mode="ENCODER", enable_action_filter=False, noise=False)
run("JUMPING_IN_PLACE_PPO (dense reward), PPO_BASIC", task_env="JUMPING_IN_PLACE_PPO", observation_space_mode="PPO_BASIC", enable_action_filter=True)
