"""TEST-ONLY stand-in for QuadrupedVecEnv in `bench.py --backend gloo` (tests/test_bench_launch.py): the surface bench.py's timed loop
uses -- reset_tensor, step_tensor, counters, timing -- on top of the CPU oracle, so that the multi-rank code of the benchmark (launcher,
gloo process group in place of RCCL, barrier, MAX / MIN over ranks, rank 0's JSON line, the other ranks' stdout) runs on a box without
GPUs.  Nothing it measures means anything."""
import types

import numpy as np
import torch


class StandInVecEnv:
    def __init__(self, num_envs=1, device=0, auto_reset=True, reset_lookahead=None, env_id_offset=0, seed=0, **kw):
        from oracle.qso import Oracle
        from qs_amd.config import build_config
        kw = dict(kw, settle_steps=100)                     # (a settle of 2500 substeps per reset would only make the test slow)
        cfg, _ = build_config(n_envs=num_envs, auto_reset=auto_reset, env_id_offset=env_id_offset, seed=seed, **kw)
        self.o = Oracle(cfg)
        self.num_envs, self.action_dim, self.obs_dim, self.device = num_envs, cfg.action_dim, cfg.obs_dim, torch.device("cpu")
        self.cfg = types.SimpleNamespace(settle_steps=cfg.settle_steps, reset_lookahead=int(reset_lookahead or 0))
        self.env_id_offset, self._steps, self._resets = env_id_offset, 0, 0

    def reset_tensor(self, mask=None):
        return torch.from_numpy(self.o.reset(None if mask is None else mask.numpy()))

    def step_tensor(self, a):
        obs, rew, done, trunc = self.o.step(a.numpy())
        self._steps += 1; self._resets += int(done.sum())
        return torch.from_numpy(obs), torch.from_numpy(rew), torch.from_numpy(done), torch.from_numpy(trunc)

    def settle_lanes(self, on=True):
        pass

    def counters_snapshot(self):     # index = QuadrupedVecEnv.COUNTERS: settle substeps, resets, served, settled, limit path, self narrow, stalls
        return torch.tensor([self._resets * self.cfg.settle_steps, self._resets, 0, 0, 0, 0, 0, 0], dtype=torch.int64)

    def counter(self, name):
        return {"settle_substeps": self._resets * self.cfg.settle_steps, "resets": self._resets}.get(name, 0)

    def enable_timing(self, on=True):
        if on is True:
            self._t0 = self._steps

    def last_step_kernel_ms(self):
        return 1.0                   # (no kernel, no kernel time; the roofline fields of the line are meaningless in this mode)

    def close(self):
        self.o.close()
