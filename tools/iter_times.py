#!/usr/bin/env python3
"""per-iteration wall times of the config-4 loop (explore_env / update_net separately, host clocks around a synchronize): finds
one-off stalls (allocator growth, first-touch) that an average hides.  ERL_FUSED_GAE=0|1"""
import os
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd.agents import AgentPPO  # noqa: E402
from elegantrl_amd.envs import SynVecEnv  # noqa: E402
from elegantrl_amd.train import Config  # noqa: E402

N, S, A, H, B, U = 4096, 64, 8, 32, 16384, 40
args = Config(AgentPPO, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 1000, "state_dim": S, "action_dim": A, "if_discrete": False})
args.horizon_len, args.batch_size, args.repeat_times, args.gpu_id = H, B, U * B / H, 0
th.manual_seed(0)
agent = AgentPPO(args.net_dims, S, A, gpu_id=0, args=args)
env = SynVecEnv(N, S, A, max_step=1000, gpu_id=0, seed=0)
agent.last_state = env.reset()[0]
rows, allocs, hosts = [], [], []
bench_like = os.environ.get("BENCH_LIKE") == "1"      # no syncs between the two calls, event brackets, K6 sampling: what bench.py does
from elegantrl_amd import _hip  # noqa: E402
for it in range(int(os.environ.get("ITERS", 30))):
    if it == 3 and os.environ.get("GC_FREEZE") == "1":
        import gc
        gc.collect()
        gc.freeze()
    if bench_like and it == 3:
        _hip.k6_null_bracket_us(200)
        _hip.k6_timing_enable(16)
    if bench_like:
        e0, e1, e2 = (th.cuda.Event(enable_timing=True) for _ in range(3))
        h0 = time.perf_counter()
        e0.record()
        items = agent.explore_env(env, H)
        e1.record()
        h1 = time.perf_counter()
        agent.update_net(list(items))
        e2.record()
        th.cuda.synchronize()
        rows.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
        hosts.append((h1 - h0) * 1e3)
    else:
        th.cuda.synchronize()
        t0 = time.perf_counter()
        items = agent.explore_env(env, H)
        th.cuda.synchronize()
        t1 = time.perf_counter()
        agent.update_net(list(items))
        th.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    allocs.append(th.cuda.memory_stats()["num_device_alloc"])
print("device allocs after each iteration:", allocs)
if hosts:
    print("host ms inside explore_env:", " ".join(f"{h:.2f}" for h in hosts))
print("fused_gae", agent.fused_gae, "explore ms:", " ".join(f"{a:.2f}" for a, _ in rows))
print("update ms:", " ".join(f"{b:.2f}" for _, b in rows))
print("reserved MB", th.cuda.memory_reserved() >> 20, "allocs", th.cuda.memory_stats()["num_device_alloc"])
