#!/usr/bin/env python3
"""BASELINE config 2 timing (AgentPPO, vectorised Pendulum-v1, 4096 envs, H = 200, net [128, 64]; not the bench line):
per-stage wall time of one iteration.  Run on the GPU box:  python tools/c2_bench.py"""
import json
import os
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd.agents import AgentPPO  # noqa: E402
from elegantrl_amd.envs import PendulumVecEnv  # noqa: E402
from elegantrl_amd.train import Config  # noqa: E402

N, H, B = 4096, 200, 16384
args = Config(AgentPPO, PendulumVecEnv, {"env_name": "Pendulum-v1", "num_envs": N, "max_step": 200, "state_dim": 3, "action_dim": 1,
                                         "if_discrete": False})
args.net_dims = [128, 64]
args.horizon_len, args.batch_size, args.repeat_times = H, B, 16 * B / H * (H * N / B) / (H * N / B)   # 16 "epochs" in the reference's units
args.repeat_times = 40 * B / H          # 40 minibatches of 16384 per iteration, as config 4
args.gamma, args.reward_scale, args.learning_rate, args.gpu_id = 0.97, 2 ** -2, 4e-4, 0
agent = AgentPPO(args.net_dims, 3, 1, gpu_id=0, args=args)
env = PendulumVecEnv(N, max_step=200, gpu_id=0, seed=0)
agent.last_state = env.reset()[0]


def timed(fn):
    th.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    th.cuda.synchronize()
    return out, time.perf_counter() - t0


for _ in range(2):
    items = agent.explore_env(env, H)
    agent.update_net(list(items))
res = {}
items, res["explore_env_ms"] = timed(lambda: agent.explore_env(env, H))
_, res["update_net_ms"] = timed(lambda: agent.update_net(list(items)))
values, res["value_prepass_ms"] = timed(lambda: agent.get_values(items[0]))
_, res["gae_200x4096_ms"] = timed(lambda: agent._gae(items[3].clone(), items[4].clone(), items[5], values))
res = {k: round(v * 1e3, 3) for k, v in res.items()}
res["env_steps_per_s"] = round(N * H / ((res["explore_env_ms"] + res["update_net_ms"]) * 1e-3))
res["fused_path"] = agent._fused
print(json.dumps(res))
