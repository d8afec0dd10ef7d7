"""Reference-state initialisation for the batched environment.

The reference's ReferenceStateInitializationWrapper (env/wrappers/reference_state_initialization_wrapper.py:9-43) wraps ONE
environment of a DEMO task: every reset places the robot in a random row of the demonstration and starts the demo counter
there; every sixth reset draws from the first fifth of the demonstration, the others from all but its last five rows.  This
is the same rule for N environments at once: the environments that finished are re-seated by one masked qs_reset_to +
qs_set_demo_counter.  (On the N = 1 QuadrupedGymEnv view the reference's own wrapper class works unchanged.)"""
import numpy as np

from .config import DEMO_FILES

TASK_ALLOWED = ("JUMPING_IN_PLACE_DEMO", "JUMPING_FORWARD_DEMO", "BACKFLIP_DEMO")   # reference_state_initialization_wrapper.py:6


class ReferenceStateInitVecEnv:
    """venv: a QuadrupedVecEnv of a DEMO task.  Everything not defined here is the wrapped environment's."""

    counter_reset_period = 5

    def __init__(self, venv, seed=None):
        self.venv = venv
        self.enable_wrapper = venv.meta["task_env"] in TASK_ALLOWED
        if venv.meta["task_env"] in DEMO_FILES and venv.demo_list is None:
            raise ValueError("the wrapped environment holds no demonstration")
        self._rng = np.random.default_rng(seed)
        self._counter = np.zeros(venv.num_envs, dtype=np.int64)
        self.random_el = np.zeros(venv.num_envs, dtype=np.int64)

    def __getattr__(self, name):
        return getattr(self.venv, name)

    def compute_random_el(self, idx):
        """reference_state_initialization_wrapper.py:35-43 for the environments `idx`."""
        L = self.venv.demo_length
        short = self._counter[idx] == self.counter_reset_period
        self._counter[idx] = np.where(short, 0, self._counter[idx] + 1)
        limit = np.where(short, L // 5, L - 5)
        return self._rng.integers(0, np.maximum(limit, 1))

    def _reseat(self, mask):
        v = self.venv
        idx = np.nonzero(mask)[0]
        self.random_el[idx] = self.compute_random_el(idx)
        states = v.demo_states(v.demo_list[self.random_el], v.action_dim)
        obs = v.reset_tensor(mask=mask.astype(np.uint8), states=states)
        v.set_demo_counter(self.random_el.astype(np.int32), mask=mask.astype(np.uint8))
        return obs

    def reset_tensor(self):
        if not self.enable_wrapper:
            return self.venv.reset_tensor()
        return self._reseat(np.ones(self.venv.num_envs, dtype=bool))

    def reset(self):
        return self.reset_tensor().cpu().numpy().copy()

    def step_tensor(self, actions):
        """(obs, rew, done, truncated) like QuadrupedVecEnv.step_tensor; the observation rows of finished environments are those
        after their re-seating, and `terminal_obs` holds the batch of observations the step itself produced."""
        obs, rew, done, trunc = self.venv.step_tensor(actions)
        if self.enable_wrapper:
            d = done.bool().cpu().numpy()
            if d.any():
                self.terminal_obs = (self.venv.get_info("terminal_obs") if self.venv.cfg.auto_reset else obs).clone()
                obs = self._reseat(d)
        return obs, rew, done, trunc

    def step(self, actions):
        obs, rew, done, infos = self.venv.step(actions)
        if self.enable_wrapper and done.any():
            if not self.venv.cfg.auto_reset:
                for i in np.nonzero(done)[0]:
                    infos[i]["terminal_observation"] = obs[i].copy()
            obs = obs.copy()
            new = self._reseat(done).cpu().numpy()
            obs[done] = new[done]
        return obs, rew, done, infos
