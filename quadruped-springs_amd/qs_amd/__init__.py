"""MI355X-native batched Go1 + PEA simulation step behind the QuadrupedGymEnv / VecEnv API of
francescovezzi/quadruped-springs.  The step itself is hand-written HIP (csrc/), reached through the C ABI of
include/qs_amd.h; there is no CPU path."""
from .config import build_config  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not need torch / a GPU
    if name == "QuadrupedVecEnv":
        from .vec_env import QuadrupedVecEnv
        return QuadrupedVecEnv
    if name == "QuadrupedGymEnv":
        from .env.quadruped_gym_env import QuadrupedGymEnv
        return QuadrupedGymEnv
    if name == "DeviceVecNormalize":
        from .vec_normalize import DeviceVecNormalize
        return DeviceVecNormalize
    if name == "ShardedVecEnv":
        from .sharded import ShardedVecEnv
        return ShardedVecEnv
    if name == "ReferenceStateInitVecEnv":
        from .rsi import ReferenceStateInitVecEnv
        return ReferenceStateInitVecEnv
    raise AttributeError(name)


def _register_gym_id():
    """quadruped_spring/__init__.py:3-12 registers "QuadrupedSpring-v0" with gym; do the same where gym is importable (it is not in
    the build image).  The keyword set is the reference's, including its observation_space_mode "ARS_HEIGHT", which no longer exists in
    sensor_collection.py -- gym.make of this id fails there with the unknown-key message, and here too."""
    try:
        from gym.envs.registration import register
    except Exception:  # noqa: BLE001
        return
    try:
        register(id="QuadrupedSpring-v0", entry_point="qs_amd.env.quadruped_gym_env:QuadrupedGymEnv",
                 kwargs={"motor_control_mode": "PD", "task_env": "JUMPING_IN_PLACE", "observation_space_mode": "ARS_HEIGHT",
                         "action_space_mode": "SYMMETRIC"})
    except Exception:  # noqa: BLE001  (already registered)
        pass


_register_gym_id()
