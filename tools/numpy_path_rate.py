#!/usr/bin/env python3
"""Rate of the SB3 numpy path (VecEnv.step on host arrays: load_model.py:113-133) at N = 8192, and where its time goes.
usage: python tools/numpy_path_rate.py [out.json]"""
import cProfile
import json
import os
import pstats
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "quadruped-springs_amd"))
import numpy as np
import torch
from qs_amd.vec_env import QuadrupedVecEnv
BC = {"true": True, "auto": "auto", "false": False}[__import__("os").environ.get("QS_BODY_CONTACTS", "true").lower()]   # the links' contact response: the default (True) or QS_BODY_CONTACTS=auto

N = 8192
out = {}
for name, kw in (("default", {}), ("views_no_info_block", dict(copy_outputs=False, info_fields=False))):
    env = QuadrupedVecEnv(num_envs=N, auto_reset=True, reset_lookahead=16, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                          enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=1, body_contacts=BC, **kw)
    env.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(64, N, 6)).astype(np.float32)
    for i in range(1200):          # spread the episode phases (untimed)
        env.step(acts[i % 64])
    steps, dones, term = 1000, 0, 0
    t0 = time.perf_counter()
    for i in range(steps):
        obs, rew, done, infos = env.step(acts[i % 64])
    dt = (time.perf_counter() - t0) / steps
    for i in range(200):
        obs, rew, done, infos = env.step(acts[i % 64])
        dones += int(done.sum()); term += sum(1 for k in np.flatnonzero(done) if "terminal_observation" in infos[k])
    # the device-tensor path of the same handle, for comparison
    a_dev = torch.as_tensor(acts, device=env.device)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for i in range(steps):
        env.step_tensor(a_dev[i % 64])
    torch.cuda.synchronize(); dt_dev = (time.perf_counter() - t1) / steps
    print(f"{name}: numpy VecEnv.step path {dt * 1e3:.3f} ms/step = {N / dt / 1e6:.1f} M env-steps/s; device-tensor path {dt_dev * 1e3:.3f} ms/step = "
          f"{N / dt_dev / 1e6:.1f} M; {dones} episode ends in 200 steps, {term} with terminal_observation; stalls {env.counter('reset_stalls')}")
    out[name] = dict(ms_per_step=dt * 1e3, env_steps_per_s=N / dt, device_tensor_env_steps_per_s=N / dt_dev, keywords=kw)
    if name == "default":
        # where the time goes: step_async (copy into pinned staging + enqueue), the wait inside step_wait, the host-side rest of step_wait
        import ctypes as C
        from qs_amd import lib as L
        tb = tw = tp = 0.0
        for i in range(300):
            t0 = time.perf_counter(); env.step_async(acts[i % 64]); t1 = time.perf_counter()
            res = L.HostResult(); L.check(env.lib.qs_host_step_end(env.h, C.byref(res))); t2 = time.perf_counter()
            env.lib.qs_host_step_begin(env.h, acts[(i + 1) % 64].ctypes.data_as(C.c_void_p))     # (keeps the begin / end pairing for the real step_wait)
            env.step_wait(); t3 = time.perf_counter()
            tb += t1 - t0; tw += t2 - t1
        for i in range(300):
            env.step_async(acts[i % 64]); t0 = time.perf_counter(); env.step_wait(); tp += time.perf_counter() - t0
        print(f"   step_async {tb / 300 * 1e6:.1f} us, wait for kernel + copies {tw / 300 * 1e6:.1f} us, whole step_wait {tp / 300 * 1e6:.1f} us per step")
        out[name]["breakdown_us"] = dict(step_async=tb / 300 * 1e6, wait=tw / 300 * 1e6, step_wait=tp / 300 * 1e6)
        pr = cProfile.Profile(); pr.enable()
        for i in range(200):
            env.step(acts[i % 64])
        pr.disable(); pstats.Stats(pr).sort_stats("cumtime").print_stats(10)
    env.close()
if len(sys.argv) > 1:
    out["what"] = "QuadrupedVecEnv.step(numpy actions) -> numpy obs / rewards / dones / infos at N = 8192, jump-in-place, U(-1,1) actions, reset_lookahead 16"
    json.dump(out, open(sys.argv[1], "w"), indent=1)
