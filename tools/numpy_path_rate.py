import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np, torch
from qs_amd.vec_env import QuadrupedVecEnv
env = QuadrupedVecEnv(num_envs=8192, auto_reset=True, reset_lookahead=8, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                      enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1)
env.reset()
a = np.random.default_rng(0).uniform(-1, 1, size=(8192, 6)).astype(np.float32)
for i in range(20): env.step(a)
t0 = time.perf_counter()
for i in range(300): env.step(a)
dt = (time.perf_counter() - t0) / 300
print(f"numpy VecEnv.step path: {dt*1e3:.3f} ms/step = {8192/dt/1e6:.1f} M env-steps/s")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(100): env.step(a)
pr.disable(); pstats.Stats(pr).sort_stats("cumtime").print_stats(12)
