"""DeviceVecNormalize: SB3's VecNormalize (stable_baselines3 1.5.1a7, common/vec_env/vec_normalize.py) with its running
statistics and the normalisation on the GPU, for consumers that keep observations on the device.

The reference applies VecNormalize at load_model.py:109-137 / get_demonstrations.py:71 (`VecNormalize.load(stats, env)`, then
`training = False`, `norm_reward = False`).  SB3's own numpy VecNormalize also wraps QuadrupedVecEnv unchanged; this class is
for `step_tensor` users.  Statistics are float64 on the device; `get_stats` / `set_stats` / `save` / `load` exchange them as the
attributes SB3 pickles (obs_rms.mean/var/count, ret_rms.mean/var/count, clip_obs, clip_reward, gamma, epsilon)."""
import ctypes as C

import numpy as np

from . import lib as _lib


class DeviceVecNormalize:
    def __init__(self, venv, training=True, norm_obs=True, norm_reward=True, clip_obs=10.0, clip_reward=10.0, gamma=0.99, epsilon=1e-8):
        self.venv = venv
        self.torch = venv.torch
        self.lib = _lib.load()
        self.num_envs, self.obs_dim, self.action_dim, self.device = venv.num_envs, venv.obs_dim, venv.action_dim, venv.device
        self.observation_space, self.action_space = venv.observation_space, venv.action_space
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon
        self.h = C.c_void_p()
        _lib.check(self.lib.qs_norm_create(self.num_envs, self.obs_dim, clip_obs, clip_reward, gamma, epsilon, self.device.index or 0, C.byref(self.h)))
        t = self.torch
        self.old_obs = t.zeros((self.num_envs, self.obs_dim), dtype=t.float32, device=self.device)
        self.old_reward = t.zeros(self.num_envs, dtype=t.float32, device=self.device)
        self._term = t.zeros((self.num_envs, self.obs_dim), dtype=t.float32, device=self.device)

    def _stream(self):
        _lib.check(self.lib.qs_norm_set_stream(self.h, C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)))

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    # ---- device path
    def reset_tensor(self, mask=None):
        obs = self.venv.reset_tensor(mask)
        self.old_obs.copy_(obs)
        self._stream()
        _lib.check(self.lib.qs_norm_reset(self.h, self._p(obs), int(self.training), int(self.norm_obs)))
        return obs

    def step_tensor(self, actions):
        """-> (obs, rew, done, truncated): normalised in place in the environment's reused buffers; the raw values of this step
        stay in `old_obs` / `old_reward` (VecNormalize.get_original_obs / get_original_reward)."""
        obs, rew, done, trunc = self.venv.step_tensor(actions)
        self._stream()
        _lib.check(self.lib.qs_norm_step(self.h, self._p(obs), self._p(rew), self._p(done), None, int(self.training), int(self.norm_obs),
                                         int(self.norm_reward), self._p(self.old_obs), self._p(self.old_reward)))
        return obs, rew, done, trunc

    def normalize_obs(self, obs):
        """normalize_obs on a host array with the current statistics (vec_normalize.py:_normalize_obs)."""
        if not self.norm_obs:
            return obs
        s = self.get_stats()
        return np.clip((obs - s["obs_mean"]) / np.sqrt(s["obs_var"] + self.epsilon), -self.clip_obs, self.clip_obs).astype(np.float32)

    def normalize_reward(self, reward):
        """normalize_reward on a host array with the current statistics (vec_normalize.py:normalize_reward)."""
        if not self.norm_reward:
            return reward
        s = self.get_stats()
        return np.clip(reward / np.sqrt(s["ret_var"] + self.epsilon), -self.clip_reward, self.clip_reward)

    def unnormalize_obs(self, obs):
        if not self.norm_obs:
            return obs
        s = self.get_stats()
        return obs * np.sqrt(s["obs_var"] + self.epsilon) + s["obs_mean"]

    def unnormalize_reward(self, reward):
        if not self.norm_reward:
            return reward
        return reward * np.sqrt(self.get_stats()["ret_var"] + self.epsilon)

    @property
    def obs_rms(self):
        """a snapshot of the running statistics under the names SB3 uses (RunningMeanStd.mean / .var / .count); set_stats writes them"""
        from types import SimpleNamespace
        s = self.get_stats()
        return SimpleNamespace(mean=s["obs_mean"], var=s["obs_var"], count=s["obs_count"])

    @property
    def ret_rms(self):
        from types import SimpleNamespace
        s = self.get_stats()
        return SimpleNamespace(mean=s["ret_mean"], var=s["ret_var"], count=s["ret_count"])

    def get_original_obs(self):
        return self.old_obs.cpu().numpy().copy()

    def get_original_reward(self):
        return self.old_reward.cpu().numpy().copy()

    # ---- SB3 VecEnv surface (numpy): load_model.py:109-137 -- env = VecNormalize.load(stats, env); env.step(numpy actions)
    def reset(self):
        return self.reset_tensor().cpu().numpy().copy()

    def step_async(self, actions):
        """The wrapped environment's host path with this normalisation attached (qs_host_set_norm): the step writes its result block on the
        device, VecNormalize.step_wait's work runs on it in place (one more launch), ONE copy brings observations, rewards, flags and the
        step's terminal observations -- normalised -- to page-locked host memory."""
        v = self.venv
        v._stream()
        _lib.check(self.lib.qs_host_set_norm(v.h, self.h, int(self.training), int(self.norm_obs), int(self.norm_reward), self._p(self.old_obs),
                                             self._p(self.old_reward)))
        v._terminal_hook = self.normalize_obs      # (only used when more episodes end in one step than the compact list holds)
        try:
            v.step_async(actions)
        except Exception:
            self._detach()
            raise

    def step_wait(self):
        """-> (obs, rewards, dones, infos) as VecNormalize.step_wait returns them: normalised observations / rewards, and
        infos[i]["terminal_observation"] normalised with the statistics of this step (vec_normalize.py step_wait)."""
        try:
            return self.venv.step_wait()
        finally:
            self._detach()

    def _detach(self):
        v = self.venv
        v._terminal_hook = None
        if v.h:
            self.lib.qs_host_set_norm(v.h, None, 0, 0, 0, None, None)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    # ---- statistics
    def get_stats(self):
        o = self.obs_dim
        mean, var = np.zeros(o), np.zeros(o)
        oc, rm, rv, rc = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        _lib.check(self.lib.qs_norm_get_stats(self.h, mean.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p), C.byref(oc), C.byref(rm),
                                              C.byref(rv), C.byref(rc)))
        return dict(obs_mean=mean, obs_var=var, obs_count=oc.value, ret_mean=rm.value, ret_var=rv.value, ret_count=rc.value)

    def set_stats(self, obs_mean, obs_var, obs_count, ret_mean=0.0, ret_var=1.0, ret_count=1e-4):
        m, v = np.ascontiguousarray(obs_mean, np.float64), np.ascontiguousarray(obs_var, np.float64)
        assert m.shape == (self.obs_dim,) and v.shape == (self.obs_dim,)
        _lib.check(self.lib.qs_norm_set_stats(self.h, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), float(obs_count), float(ret_mean),
                                              float(ret_var), float(ret_count)))

    def save(self, path):
        np.savez(path, clip_obs=self.clip_obs, clip_reward=self.clip_reward, gamma=self.gamma, epsilon=self.epsilon, norm_obs=self.norm_obs,
                 norm_reward=self.norm_reward, **self.get_stats())

    @classmethod
    def load(cls, path, venv):
        """`path`: an .npz written by save(), or an SB3 VecNormalize pickle (needs stable_baselines3 importable)."""
        if str(path).endswith(".npz"):
            z = np.load(path)
            self = cls(venv, clip_obs=float(z["clip_obs"]), clip_reward=float(z["clip_reward"]), gamma=float(z["gamma"]), epsilon=float(z["epsilon"]),
                       norm_obs=bool(z["norm_obs"]), norm_reward=bool(z["norm_reward"]))
            self.set_stats(z["obs_mean"], z["obs_var"], float(z["obs_count"]), float(z["ret_mean"]), float(z["ret_var"]), float(z["ret_count"]))
            return self
        import pickle
        with open(path, "rb") as f:
            sb3 = pickle.load(f)
        self = cls(venv, clip_obs=sb3.clip_obs, clip_reward=sb3.clip_reward, gamma=sb3.gamma, epsilon=sb3.epsilon, norm_obs=sb3.norm_obs,
                   norm_reward=sb3.norm_reward)
        self.set_stats(sb3.obs_rms.mean, sb3.obs_rms.var, sb3.obs_rms.count, sb3.ret_rms.mean, sb3.ret_rms.var, sb3.ret_rms.count)
        return self

    def close(self):
        if self.h:
            self._detach()          # (the environment's host path must not keep a pointer to the statistics that go away now)
            self.lib.qs_norm_destroy(self.h)
            self.h = None
        self.venv.close()

    def __getattr__(self, name):   # everything else is the wrapped environment's
        return getattr(self.venv, name)
