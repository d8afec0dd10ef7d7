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
    raise AttributeError(name)
