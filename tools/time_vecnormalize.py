import sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import torch
from qs_amd import DeviceVecNormalize, QuadrupedVecEnv
venv = QuadrupedVecEnv(num_envs=8192, device=0, auto_reset=True, reset_lookahead=8, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                       enable_springs=True, enable_action_filter=True)
env = DeviceVecNormalize(venv)
obs = env.reset_tensor()
a = torch.rand((8192, 6), device="cuda") * 2 - 1
def timeit(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("raw step_tensor        %.3f ms" % timeit(lambda: venv.step_tensor(a)))
print("normalised step_tensor %.3f ms" % timeit(lambda: env.step_tensor(a)))
env.training = False
print("normalised, eval mode  %.3f ms" % timeit(lambda: env.step_tensor(a)))
pol = lambda o: torch.tanh(o[:, :6] * 3.0)
print("policy only            %.3f ms" % timeit(lambda: pol(obs)))
