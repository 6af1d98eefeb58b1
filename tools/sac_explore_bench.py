#!/usr/bin/env python3
"""Config 3's off-policy rollout alone: 64 steps x 64 envs of AgentSAC.explore_env (one explore-action launch, one state-row copy, one env
step per time step; the interpreter enqueues them)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402
from elegantrl_amd.agents import AgentSAC  # noqa: E402
from elegantrl_amd.envs import SynVecEnv  # noqa: E402
from elegantrl_amd.train import Config  # noqa: E402

N, S, A, H = 64, 11, 3, 64
args = Config(AgentSAC, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 1000, "state_dim": S, "action_dim": A, "if_discrete": False})
args.net_dims, args.horizon_len, args.batch_size, args.gpu_id = [256, 256], H, 256, 0
agent = AgentSAC(args.net_dims, S, A, gpu_id=0, args=args)
env = SynVecEnv(N, S, A, max_step=1000, gpu_id=0, seed=0)
agent.last_state = env.reset()[0]
for _ in range(5):
    agent.explore_env(env, H)
th.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    agent.explore_env(env, H)
t1 = time.perf_counter()
th.cuda.synchronize()
t2 = time.perf_counter()
print(f"AgentSAC.explore_env, {H} steps x {N} envs: host returned after {(t1 - t0) / 20 * 1e3:.2f} ms, GPU done after {(t2 - t0) / 20 * 1e3:.2f} ms per rollout "
      f"({(t2 - t0) / 20 / H * 1e6:.1f} us per time step)")
