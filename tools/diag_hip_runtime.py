import sys, os, subprocess
sys.path.insert(0, 'quadruped-springs_amd')
case = sys.argv[1]
def maps():
    return sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l))
import ctypes as C
def create():
    from qs_amd import lib
    from qs_amd.config import build_config
    cfg, _ = build_config(n_envs=16, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC")
    h = C.c_void_p()
    rc = lib.load().qs_create(C.byref(cfg), 0, C.byref(h))
    print(case, "qs_create rc", rc, lib.load().qs_last_error())
if case == "A":
    import torch; torch.cuda.set_device(0); x = torch.zeros(1, device="cuda"); create()
elif case == "B":
    import torch; print("avail", torch.cuda.is_available()); create()
elif case == "C":
    from qs_amd import lib; lib.load(); import torch; print("avail", torch.cuda.is_available()); create()
elif case == "D":
    create()
print(case, maps())
