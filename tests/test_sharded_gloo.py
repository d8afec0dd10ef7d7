"""N > 1 path on CPU: world_size-2 gloo processes, each owning half of the environments, must reproduce the single-process
result bit for bit (global-id keyed RNG, no cross-environment state).  The local environment here is the CPU oracle
wrapped in the `step_tensor` / `reset_tensor` interface of QuadrupedVecEnv (tests only)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True,
          env_randomizer_mode="GROUND_RANDOMIZER", seed=11, noise=True, auto_reset=True, settle_steps=300)
N_GLOBAL, STEPS = 4, 12


class OracleEnv:
    """CPU stand-in with the device-tensor interface of QuadrupedVecEnv."""

    def __init__(self, n, offset):
        from oracle.qso import Oracle
        from qs_amd.config import build_config
        cfg, _ = build_config(n_envs=n, env_id_offset=offset, **KW)
        self.o = Oracle(cfg)
        self.num_envs, self.action_dim, self.obs_dim, self.device = n, cfg.action_dim, cfg.obs_dim, torch.device("cpu")

    def reset_tensor(self):
        return torch.from_numpy(self.o.reset())

    def step_tensor(self, a):
        obs, rew, done, trunc = self.o.step(a.numpy())
        return torch.from_numpy(obs), torch.from_numpy(rew), torch.from_numpy(done), torch.from_numpy(trunc)


class FusedOracleEnv(OracleEnv):
    """... and with QuadrupedVecEnv.step_fused: the rows [obs | reward | done + 2 truncated] written into the caller's slice, which is
    the branch of ShardedVecEnv.step the GPU build takes (in-place all-gather of the learner's buffer)."""

    def step_fused(self, a, out):
        obs, rew, done, trunc = self.o.step(a.numpy())
        out[:, : self.obs_dim] = torch.from_numpy(obs)
        out[:, self.obs_dim] = torch.from_numpy(rew)
        out[:, self.obs_dim + 1] = torch.from_numpy(done.astype(np.float32) + 2 * trunc.astype(np.float32))
        return out


def actions():
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, size=(STEPS, N_GLOBAL, 6)).astype(np.float32)
    a[4:8, :2] = [0.0, -1.0, 1.0, 0.0, -1.0, 1.0]
    return a


def worker(rank, world, port, out, fused=False):
    for p in (REPO, os.path.join(REPO, "quadruped-springs_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qs_amd.sharded import ShardedVecEnv
    n = N_GLOBAL // world
    from qs_amd.sharded import decode_flags
    env = ShardedVecEnv((FusedOracleEnv if fused else OracleEnv)(n, ShardedVecEnv.env_id_offset(n)), learner_rank=0)
    res = [env.reset().clone()]
    roll = torch.zeros((3, N_GLOBAL, env.obs_dim + 2))
    for k, a in enumerate(actions()):
        if fused and k % 2:      # the learner's own rollout row as the gather buffer, flags decoded by the reader
            g = env.step(torch.from_numpy(a) if rank == 0 else None, out=roll[k % 3], unpack=False)
            assert g.data_ptr() == roll[k % 3].data_ptr()
            o, r = g[:, : env.obs_dim], g[:, env.obs_dim]
            d, t = decode_flags(g[:, env.obs_dim + 1])
        else:
            o, r, d, t = env.step(torch.from_numpy(a) if rank == 0 else None)
        res.append(torch.cat([o, r[:, None], d[:, None].float(), t[:, None].float()], dim=1).clone())
    if rank == 1:  # every rank holds the full gathered result
        torch.save(torch.stack([x if x.shape[1] == res[1].shape[1] else torch.nn.functional.pad(x, (0, 3)) for x in res]), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("fused", [False, True])
def test_two_ranks_equal_one(tmp_path, fused):
    out = str(tmp_path / "r1.pt")
    port = 29500 + os.getpid() % 2000 + (7 if fused else 0)
    mp.spawn(worker, args=(2, port, out, fused), nprocs=2, join=True)
    got = torch.load(out).numpy()
    single = OracleEnv(N_GLOBAL, 0)
    ref = [np.pad(single.reset_tensor().numpy(), ((0, 0), (0, 3)))]
    for a in actions():
        o, r, d, t = single.step_tensor(torch.from_numpy(a))
        ref.append(np.concatenate([o.numpy(), r.numpy()[:, None], d.numpy()[:, None].astype(np.float32), t.numpy()[:, None].astype(np.float32)], axis=1))
    assert np.array_equal(got, np.stack(ref))
