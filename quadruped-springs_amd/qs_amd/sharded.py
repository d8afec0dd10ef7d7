"""Environment sharding over the GPUs of one node.

Environments are independent, so the simulation data path needs no collective: rank r owns the global environments
[r*n, (r+1)*n) and its Philox streams are keyed by the global id (`env_id_offset`), which makes every result independent
of the number of ranks.  What a centralised learner needs per step is:
  * actions:  broadcast of the global [N, d] batch from the learner rank (each rank keeps its slice), and
  * results:  ONE all-gather of a fused [n, o+2] float32 buffer (obs | reward | done+2*truncated) per rank.
Backend "nccl" is RCCL over xGMI on the GPU box; the same code runs on "gloo" for the CPU tests, where the local
environment is any object with `step_tensor` / `reset_tensor` (tests plug in the CPU oracle).
"""
import torch
import torch.distributed as dist


class ShardedVecEnv:
    def __init__(self, local_env, learner_rank=0, group=None):
        self.env = local_env
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.learner_rank = learner_rank
        self.n_local = local_env.num_envs
        self.num_envs = self.n_local * self.world
        self.action_dim, self.obs_dim = local_env.action_dim, local_env.obs_dim
        self.device = local_env.device
        self._fused = torch.zeros((self.n_local, self.obs_dim + 2), dtype=torch.float32, device=self.device)
        self._gathered = torch.zeros((self.num_envs, self.obs_dim + 2), dtype=torch.float32, device=self.device)
        self._actions = torch.zeros((self.num_envs, self.action_dim), dtype=torch.float32, device=self.device)

    @staticmethod
    def env_id_offset(n_local, rank=None):
        return n_local * (dist.get_rank() if rank is None else rank)

    def _unpack(self):
        g = self._gathered
        flags = g[:, self.obs_dim + 1]
        # done = flag 1 or 3, truncated = flag 3 (done without termination, gym_env.py:246); obs / reward are views, not copies
        return g[:, : self.obs_dim], g[:, self.obs_dim], flags > 0.5, flags > 2.5

    def _gather(self, obs, rew=None, done=None, trunc=None):
        f = self._fused
        f[:, : self.obs_dim] = obs
        f[:, self.obs_dim] = 0 if rew is None else rew
        f[:, self.obs_dim + 1] = 0 if done is None else done.to(torch.float32) + 2 * trunc.to(torch.float32)
        dist.all_gather_into_tensor(self._gathered, f, group=self.group)
        return self._unpack()

    def reset(self):
        return self._gather(self.env.reset_tensor())[0]

    def step(self, actions=None):
        """`actions`: the global [N, d] batch on the learner rank (ignored elsewhere).  Returns the global
        (obs [N,o], rew [N], done [N], truncated [N]) on every rank (views of the gathered buffer, valid until the next call)."""
        buf = actions if (self.rank == self.learner_rank and actions.is_contiguous() and actions.dtype == torch.float32
                          and actions.device == self._actions.device) else self._actions
        if self.rank == self.learner_rank and buf is self._actions:
            self._actions.copy_(actions)
        dist.broadcast(buf, src=self.learner_rank, group=self.group)
        lo = self.rank * self.n_local
        mine = buf[lo: lo + self.n_local]          # a row slice: contiguous
        if hasattr(self.env, "step_fused"):        # the step kernel writes the fused row itself (qs_step_fused)
            self.env.step_fused(mine, self._fused)
            dist.all_gather_into_tensor(self._gathered, self._fused, group=self.group)
            return self._unpack()
        obs, rew, done, trunc = self.env.step_tensor(mine.contiguous())
        return self._gather(obs, rew, done, trunc)
