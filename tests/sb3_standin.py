"""TEST-ONLY stand-in for `stable_baselines3.common.vec_env` (SB3 1.5.1a7 is what the reference pins through rl-baselines3-zoo; absent from this
image and from the GPU box).  It restates the INTERFACE of SB3's VecEnv as load_model.py:109-137 relies on it -- the constructor's signature
and attributes, the abstract method set, step() = step_async + step_wait -- so that the test can prove that qs_amd.QuadrupedVecEnv, which
subclasses the real class when SB3 is importable (qs_amd/spaces.py), is instantiable under the ABC's rules and calls the base constructor.
install() puts it into sys.modules BEFORE qs_amd is imported."""
import sys
import types
from abc import ABC, abstractmethod


class VecEnv(ABC):
    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space
        self.base_constructor_ran = True           # (the stand-in's own marker)

    @abstractmethod
    def reset(self): ...

    @abstractmethod
    def step_async(self, actions): ...

    @abstractmethod
    def step_wait(self): ...

    @abstractmethod
    def close(self): ...

    @abstractmethod
    def get_attr(self, attr_name, indices=None): ...

    @abstractmethod
    def set_attr(self, attr_name, value, indices=None): ...

    @abstractmethod
    def env_method(self, method_name, *method_args, indices=None, **method_kwargs): ...

    @abstractmethod
    def env_is_wrapped(self, wrapper_class, indices=None): ...

    @abstractmethod
    def seed(self, seed=None): ...

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def get_images(self):
        raise NotImplementedError

    def render(self, mode="human"):
        raise NotImplementedError

    @property
    def unwrapped(self):
        return self


def install():
    sb3 = types.ModuleType("stable_baselines3")
    common = types.ModuleType("stable_baselines3.common")
    vec = types.ModuleType("stable_baselines3.common.vec_env")
    vec.VecEnv = VecEnv
    sb3.common, common.vec_env = common, vec
    sys.modules.update({"stable_baselines3": sb3, "stable_baselines3.common": common, "stable_baselines3.common.vec_env": vec})
    return VecEnv
