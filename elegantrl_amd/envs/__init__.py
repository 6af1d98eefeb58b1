from .vec_envs import PendulumVecEnv, SynVecEnv
