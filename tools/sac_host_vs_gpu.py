#!/usr/bin/env python3
"""Config 3 (bench.py --config c3 setup): is the SAC step bound by the host enqueueing ~52 launches per update or by the GPU
executing them?  320 x [sample(256) + update] with no synchronisation: time until the host returns vs until the GPU is done."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402
from elegantrl_amd import ops
from elegantrl_amd.agents import AgentSAC
from elegantrl_amd.envs import SynVecEnv
from elegantrl_amd.train import Config, ReplayBuffer
dev = th.device("cuda:0")
N, S, A, H, B, UPD, NET = 64, 11, 3, 64, 256, 64, [256, 256]
max_size = 1_000_000 // N
args = Config(AgentSAC, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 1000, "state_dim": S, "action_dim": A, "if_discrete": False})
args.net_dims, args.horizon_len, args.batch_size = NET, H, B
args.repeat_times = UPD * B / max_size
args.gpu_id, args.random_seed = 0, 0
agent = AgentSAC(args.net_dims, S, A, gpu_id=0, args=args)
env = SynVecEnv(N, S, A, max_step=1000, gpu_id=0, seed=0)
agent.last_state = env.reset()[0]
buf = ReplayBuffer(max_size=max_size, state_dim=S, action_dim=A, gpu_id=0, num_seqs=N)
g = th.Generator(device=dev).manual_seed(1)
for _ in range(2):
    buf.update((th.randn((max_size // 2 + 7, N, S), device=dev, generator=g), th.randn((max_size // 2 + 7, N, A), device=dev, generator=g).tanh(),
                th.randn((max_size // 2 + 7, N), device=dev, generator=g), th.rand((max_size // 2 + 7, N), device=dev, generator=g) < 0.99,
                th.rand((max_size // 2 + 7, N), device=dev, generator=g) < 0.995))
for _ in range(3):
    agent.update_net(buf)
th.cuda.synchronize()
objs = th.zeros((320, 2), dtype=th.float32, device=dev)
t0 = time.perf_counter()
for t in range(320):
    agent._update_on_batch(buf.sample(256, reuse=True), objs[t])
t1 = time.perf_counter()
th.cuda.synchronize()
t2 = time.perf_counter()
print(f"320 x [sample + update] without a sync: host returned after {(t1-t0)*1e3:.1f} ms, GPU done after {(t2-t0)*1e3:.1f} ms -> {(t2-t0)/320*1e6:.0f} us per update")
