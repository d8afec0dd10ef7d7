"""N = 1 view of the batched environment behind the reference's QuadrupedGymEnv interface (see quadruped_gym_env.py)."""
