"""gym / stable-baselines3 are optional (absent in the build image): use them when importable, else small stand-ins
that carry the same attributes the reference and SB3 read (low, high, shape, dtype, sample)."""
import numpy as np

try:  # pragma: no cover - exercised only where gym is installed
    from gym import Env as GymEnv
    from gym import spaces as _spaces

    Box = _spaces.Box
except Exception:  # noqa: BLE001
    class GymEnv:  # minimal gym.Env duck type
        metadata = {}

        @property
        def unwrapped(self):
            return self

    class Box:
        def __init__(self, low, high, dtype=np.float32):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.dtype = np.dtype(dtype)
            self.shape = self.low.shape

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

try:  # pragma: no cover
    from stable_baselines3.common.vec_env import VecEnv as SB3VecEnv
except Exception:  # noqa: BLE001
    SB3VecEnv = object
