#!/usr/bin/env python3
"""Host-side cost of one PPO iteration's launches (enqueue time without waiting for the GPU) vs GPU time."""
import os
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd.agents import AgentPPO  # noqa: E402
from elegantrl_amd.envs import SynVecEnv  # noqa: E402
from elegantrl_amd.train import Config  # noqa: E402

N, S, A, H, B, U = 4096, 64, 8, 32, 16384, 40
args = Config(AgentPPO, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 1000, "state_dim": S, "action_dim": A,
                                    "if_discrete": False})
args.net_dims = [128, 128]
args.horizon_len, args.batch_size, args.repeat_times = H, B, U * B / H
args.gpu_id = 0
agent = AgentPPO(args.net_dims, S, A, gpu_id=0, args=args)
env = SynVecEnv(N, S, A, max_step=1000, gpu_id=0, seed=0)
agent.last_state = env.reset()[0]
for _ in range(3):
    items = agent.explore_env(env, H)
    agent.update_net(list(items))
th.cuda.synchronize()
for name, fn in (("explore_env", lambda: agent.explore_env(env, H)), ("update_net", lambda: agent.update_net(list(items)))):
    hs, gs = [], []
    for _ in range(5):
        th.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        t1 = time.perf_counter()
        th.cuda.synchronize()
        t2 = time.perf_counter()
        if name == "explore_env":
            items = out
        hs.append(t1 - t0)
        gs.append(t2 - t0)
    print(f"{name}: host enqueue {min(hs) * 1e3:.3f} ms, until GPU idle {min(gs) * 1e3:.3f} ms   (cores: {os.cpu_count()}, "
          f"affinity {len(os.sched_getaffinity(0))})")
