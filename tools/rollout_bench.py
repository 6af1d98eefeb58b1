#!/usr/bin/env python3
"""HIP-event timing of AgentPPO.explore_env: the persistent H-step rollout (csrc/rollout_fused.hip) vs the per-step launches,
at the BASELINE shapes.  Prints one JSON line per case: us per explore_env call, us per step, env-steps/s of the rollout alone,
and the fp32 MFMA fraction of the fused kernel (algorithmic flops = actor + critic forward + SynVecEnv map per env-step)."""
import json
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd.agents import AgentPPO            # noqa: E402
from elegantrl_amd.envs import PendulumVecEnv, SynVecEnv   # noqa: E402
from elegantrl_amd.train import Config                # noqa: E402

PEAK = 157.3e12
CASES = [("c4", "syn", 4096, 64, 8, (128, 128), 32), ("c5", "syn", 8192, 60, 8, (128, 128), 32),
         ("c2", "pendulum", 4096, 3, 1, (128, 64), 200)]


def run(kind, N, S, A, net, H, fused, iters=20):
    args = Config(AgentPPO, None, {"env_name": kind, "num_envs": N, "max_step": 1000, "state_dim": S, "action_dim": A,
                                   "if_discrete": False})
    args.net_dims, args.fused_rollout = list(net), fused
    th.manual_seed(0)
    agent = AgentPPO(args.net_dims, S, A, gpu_id=0, args=args)
    env = PendulumVecEnv(N, gpu_id=0) if kind == "pendulum" else SynVecEnv(N, S, A, gpu_id=0)
    agent.last_state = env.reset()[0]
    for _ in range(3):
        agent.explore_env(env, H)
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        agent.explore_env(env, H)
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


if __name__ == "__main__":
    for tag, kind, N, S, A, net, H in CASES:
        h1, h2 = net
        flops = 2 * (S * h1 + h1 * h2 + h2 * A) + 2 * (S * h1 + h1 * h2 + h2) + (2 * (S * S + A * S) if kind == "syn" else 0)
        out = {"case": tag, "N": N, "S": S, "A": A, "net": net, "H": H}
        for fused in (True, False):
            us = run(kind, N, S, A, net, H, fused)
            key = "fused" if fused else "per_step"
            out[key] = {"us_per_rollout": round(us, 1), "us_per_step": round(us / H, 2), "env_steps_per_s": round(N * H / us * 1e6)}
            if fused:      # H + 1 critic passes, H actor / env passes: count H full steps
                out[key]["mfma_frac"] = round(flops * N * H / (us * 1e-6) / PEAK, 4)
        print(json.dumps(out), flush=True)
