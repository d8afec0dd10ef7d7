"""QuadrupedVecEnv as a stable_baselines3 VecEnv (load_model.py:109-137) without SB3 in the image: a test-only stand-in of the ABC
(tests/sb3_standin.py) goes into sys.modules before qs_amd is imported, in a fresh interpreter."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRELUDE = f"""
import sys
sys.path[:0] = [{REPO!r}, {os.path.join(REPO, 'tests')!r}, {os.path.join(REPO, 'quadruped-springs_amd')!r}]
import sb3_standin
VecEnv = sb3_standin.install()
import numpy as np
from qs_amd.vec_env import QuadrupedVecEnv
from qs_amd import spaces
assert spaces.SB3VecEnv is VecEnv and issubclass(QuadrupedVecEnv, VecEnv)
"""


def run(body):
    r = subprocess.run([sys.executable, "-c", PRELUDE + body], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def test_subclass_is_concrete_and_calls_the_base_constructor():
    out = run("""
assert QuadrupedVecEnv.__abstractmethods__ == frozenset(), sorted(QuadrupedVecEnv.__abstractmethods__)
# the constructor's call of VecEnv.__init__ (QuadrupedVecEnv._init_vec_env_base), on an instance without a device handle
e = QuadrupedVecEnv.__new__(QuadrupedVecEnv)
e.num_envs, e.observation_space, e.action_space = 7, spaces.Box(-np.ones(3), np.ones(3)), spaces.Box(-np.ones(2), np.ones(2))
e._init_vec_env_base()
assert e.base_constructor_ran and e.num_envs == 7 and e.action_space.shape == (2,)
e._closed = True
print("ok")
""")
    assert "ok" in out


@pytest.mark.gpu
def test_instantiates_and_steps_as_a_vec_env():
    out = run("""
env = QuadrupedVecEnv(num_envs=32, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True,
                      env_randomizer_mode="GROUND_RANDOMIZER", seed=3)
assert isinstance(env, VecEnv) and env.base_constructor_ran and env.num_envs == 32
obs = env.reset()
o1, r1, d1, i1 = env.step(np.zeros((32, 6), np.float32))          # VecEnv.step of the base class: step_async + step_wait
assert o1.shape == (32, env.observation_space.shape[0]) and r1.shape == (32,) and len(i1) == 32
# seed(): a new handle under the new seed; the same seed gives the same first observations, another seed other ones (GROUND_RANDOMIZER + sensor noise)
assert env.seed(11) == [11 + i for i in range(32)]
a = env.reset().copy()
env.seed(11); b = env.reset().copy()
env.seed(12); c = env.reset().copy()
assert np.array_equal(a, b) and not np.array_equal(a, c)
assert env.seed(None) == [None] * 32
env.close()
print("ok")
""")
    assert "ok" in out
