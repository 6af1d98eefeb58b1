from .vec_envs import CartPoleVecEnv, PendulumVecEnv, SynVecEnv
