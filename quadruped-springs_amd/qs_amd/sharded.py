"""Environment sharding over the GPUs of one node.

Environments are independent, so the simulation data path needs no collective: rank r owns the global environments
[r*n, (r+1)*n) and its Philox streams are keyed by the global id (`env_id_offset`), which makes every result independent
of the number of ranks.  What a centralised learner needs per step is:
  * actions:  broadcast of the global [N, d] batch from the learner rank (each rank keeps its slice), and
  * results:  ONE all-gather of a fused [n, o+2] float32 buffer (obs | reward | done+2*truncated) per rank.
Nothing else is launched: the step kernel writes this rank's rows straight into its slice of the gathered buffer (which may be
the learner's own rollout row, `out=`), the all-gather runs in place on that buffer, and on one rank both collectives are skipped.
Backend "nccl" is RCCL over xGMI on the GPU box; the same code runs on "gloo" for the CPU tests, where the local
environment is any object with `step_tensor` / `reset_tensor` (tests plug in the CPU oracle).
"""
import torch
import torch.distributed as dist

from .config import decode_flags  # noqa: F401  (re-exported: the decoder of the fused rows' flag column)


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
        self._gathered = torch.zeros((self.num_envs, self.obs_dim + 2), dtype=torch.float32, device=self.device)
        self._actions = torch.zeros((self.num_envs, self.action_dim), dtype=torch.float32, device=self.device)
        self.lo = self.rank * self.n_local

    @staticmethod
    def env_id_offset(n_local, rank=None):
        return n_local * (dist.get_rank() if rank is None else rank)

    def _unpack(self, g):
        done, trunc = decode_flags(g[:, self.obs_dim + 1])
        return g[:, : self.obs_dim], g[:, self.obs_dim], done, trunc   # obs / reward are views of the gathered buffer, not copies

    def _all_gather(self, g):
        if self.world > 1:   # in place: this rank's rows already sit in its slice of g
            dist.all_gather_into_tensor(g, g[self.lo: self.lo + self.n_local], group=self.group)

    def _check_out(self, out):
        if out is None:
            return self._gathered
        if tuple(out.shape) != (self.num_envs, self.obs_dim + 2) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self._gathered.device:
            raise ValueError(f"out must be a contiguous float32 tensor of shape {(self.num_envs, self.obs_dim + 2)} on {self._gathered.device}")
        return out

    def reset(self):
        g = self._gathered
        mine = g[self.lo: self.lo + self.n_local]
        mine[:, : self.obs_dim] = self.env.reset_tensor()
        mine[:, self.obs_dim:] = 0
        self._all_gather(g)
        return self._unpack(g)[0]

    def step(self, actions=None, out=None, unpack=True):
        """`actions`: the global [N, d] batch on the learner rank (ignored elsewhere).  `out` (optional): a contiguous float32
        [N, o + 2] tensor that receives the gathered rows (e.g. one row block of the learner's rollout buffer) instead of the
        internal buffer.  Returns the global (obs [N,o], rew [N], done [N], truncated [N]) on every rank: views of / computed
        from that buffer, valid until it is written again.  `unpack=False` returns the gathered [N, o + 2] buffer itself and launches
        nothing for the flags (two comparison kernels otherwise): a learner that stores the fused rows decodes them when it reads them
        (`decode_flags`)."""
        g = self._check_out(out)
        learner = self.rank == self.learner_rank
        direct = learner and actions.is_contiguous() and actions.dtype == torch.float32 and actions.device == self._actions.device
        buf = actions if direct else self._actions
        if learner and not direct:
            self._actions.copy_(actions)
        if self.world > 1:
            dist.broadcast(buf, src=self.learner_rank, group=self.group)
        mine = buf[self.lo: self.lo + self.n_local]          # a row slice: contiguous
        rows = g[self.lo: self.lo + self.n_local]
        if hasattr(self.env, "step_fused"):        # the step kernel writes the fused rows itself (qs_step_fused), in place
            self.env.step_fused(mine, rows)
        else:
            obs, rew, done, trunc = self.env.step_tensor(mine.contiguous())
            rows[:, : self.obs_dim] = obs
            rows[:, self.obs_dim] = rew
            rows[:, self.obs_dim + 1] = done.to(torch.float32) + 2 * trunc.to(torch.float32)
        self._all_gather(g)
        return self._unpack(g) if unpack else g
