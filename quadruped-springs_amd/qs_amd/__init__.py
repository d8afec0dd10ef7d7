"""MI355X-native batched Go1 + PEA simulation step behind the QuadrupedGymEnv / VecEnv API."""
