import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import torch, ctypes as C
from qs_amd import config as qc
orig = qc.build_config
for iters in (30, 0, 10, 60):
    def bc(**kw):
        cfg, meta = orig(**kw); cfg.solver_iters = iters; return cfg, meta
    import qs_amd.vec_env as ve
    ve.build_config = bc
    env = ve.QuadrupedVecEnv(num_envs=8192, auto_reset=True, reset_lookahead=8, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                             enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1)
    env.reset_tensor()
    a = torch.rand((16, 8192, 6), device="cuda") * 2 - 1
    for i in range(30): env.step_tensor(a[i % 16])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(300): env.step_tensor(a[i % 16])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    print(f"solver_iters={iters}: {dt*1e3:.4f} ms/step")
    env.close()
