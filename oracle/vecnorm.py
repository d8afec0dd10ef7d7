"""ORACLE (test infrastructure, not product code): numpy restatement of SB3's VecNormalize on arrays.

stable_baselines3 == 1.5.1a7 is pinned by the reference (setup.py:11) but is absent from /root/reference and from this image,
so its published algorithm is restated here: common/running_mean_std.py (RunningMeanStd.__init__, update,
update_from_moments) and common/vec_env/vec_normalize.py (VecNormalize.reset, step_wait, _update_reward, _normalize_obs,
normalize_obs, normalize_reward).  The reference's call sites: load_model.py:109-137, get_demonstrations.py:71.
Parity unpinned against SB3 itself (cannot be imported here); the formulas are its documented parallel-variance update."""
import numpy as np


class RunningMeanStd:
    def __init__(self, epsilon=1e-4, shape=(), moments_dtype=None):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = epsilon
        # SB3 takes np.mean / np.var of the array as it comes: float32 observations are summed row after row in float32
        # (error ~ N * 6e-8 relative).  moments_dtype=np.float64 gives the same algorithm with exact batch moments, which is
        # what the device kernels compute; the two agree to float32 accuracy wherever var + epsilon does not amplify it.
        self.moments_dtype = moments_dtype

    def update(self, arr):
        if self.moments_dtype is not None:
            arr = np.asarray(arr, self.moments_dtype)
        self.update_from_moments(np.mean(arr, axis=0), np.var(arr, axis=0), arr.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot_count = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot_count
        m_a = self.var * self.count
        m_b = batch_var * batch_count
        m_2 = m_a + m_b + np.square(delta) * self.count * batch_count / (self.count + batch_count)
        self.mean, self.var, self.count = new_mean, m_2 / (self.count + batch_count), batch_count + self.count


class VecNormalizeRef:
    def __init__(self, n_envs, obs_dim, training=True, norm_obs=True, norm_reward=True, clip_obs=10.0, clip_reward=10.0, gamma=0.99, epsilon=1e-8,
                 moments_dtype=None):
        self.obs_rms, self.ret_rms = RunningMeanStd(shape=(obs_dim,), moments_dtype=moments_dtype), RunningMeanStd(shape=(), moments_dtype=moments_dtype)
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon
        self.returns = np.zeros(n_envs)

    def normalize_obs(self, obs):
        if not self.norm_obs:
            return obs
        return np.clip((obs - self.obs_rms.mean) / np.sqrt(self.obs_rms.var + self.epsilon), -self.clip_obs, self.clip_obs).astype(np.float32)

    def normalize_reward(self, reward):
        if self.norm_reward:
            reward = np.clip(reward / np.sqrt(self.ret_rms.var + self.epsilon), -self.clip_reward, self.clip_reward)
        return reward

    def reset(self, obs):
        self.returns = np.zeros(len(obs))
        if self.training and self.norm_obs:
            self.obs_rms.update(obs)
        return self.normalize_obs(obs)

    def step(self, obs, rewards, dones, term_obs=None):
        if self.training and self.norm_obs:
            self.obs_rms.update(obs)
        obs = self.normalize_obs(obs)
        if self.training:
            self.returns = self.returns * self.gamma + rewards
            self.ret_rms.update(self.returns)
        rewards = self.normalize_reward(rewards)
        if term_obs is not None:
            term_obs = self.normalize_obs(term_obs)
        self.returns[dones.astype(bool)] = 0
        return obs, rewards, term_obs
