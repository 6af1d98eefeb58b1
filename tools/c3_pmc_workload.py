#!/usr/bin/env python3
"""Workloads for config 3's counter passes (rocprofv3 --kernel-trace --pmc <one counter set per pass>; SURVEY.md 8d: "report both
algorithmic and FETCH_SIZE bytes" for the replay gather):
    C3_PART=step  a few fused SAC update steps at config 3's shapes (B = 256, net [256, 256], 4 critics, S = 11, A = 3): critic_tile_kernel<MODE, ..>,
                  actor_fwd_kernel, dw_table_kernel, ... -- summarise with PMC_KEEP_TEMPLATE=1 to keep the three critic passes apart
    C3_PART=k9    replay_sample_rows_kernel (C3_RING=planar: replay_sample_kernel) alone, CASES x 5 launches in a fixed order: (num_seqs, B) in [(64, 256), (64, 4096), (64, 2^20),
                  (1, 256), (1, 4096), (1, 2^20)] on a 1e6-transition ring -- summarise with PMC_CASES=replay_sample_kernel:5:<names>
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o p -- python tools/c3_pmc_workload.py"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ERL_QUIET", "1")
from elegantrl_amd import ops  # noqa: E402
from elegantrl_amd.agents import AgentSAC  # noqa: E402
from elegantrl_amd.train import Config, ReplayBuffer  # noqa: E402

dev = th.device("cuda:0")
S, A = 11, 3
g = th.Generator(device=dev).manual_seed(1)
part = os.environ.get("C3_PART", "step")
K9_CASES = [(64, 256), (64, 4096), (64, 1 << 20), (1, 256), (1, 4096), (1, 1 << 20)]


def ring(num_seqs):
    rows = 1_000_000 // num_seqs
    a = Config()
    a.replay_interleaved = os.environ.get("C3_RING", "interleaved") != "planar"     # C3_RING=planar: the reference's five tensors (round 5's layout)
    buf = ReplayBuffer(max_size=rows, state_dim=S, action_dim=A, gpu_id=0, num_seqs=num_seqs, args=a)
    n = rows - 1
    buf.update((th.randn((n, num_seqs, S), device=dev, generator=g), th.randn((n, num_seqs, A), device=dev, generator=g).tanh(),
                th.randn((n, num_seqs), device=dev, generator=g), th.rand((n, num_seqs), device=dev, generator=g) < 0.99,
                th.rand((n, num_seqs), device=dev, generator=g) < 0.995))
    return buf


if part == "k9":
    rings = {64: ring(64), 1: ring(1)}
    th.cuda.synchronize()
    for seqs, B in K9_CASES:
        r = rings[seqs]
        idx = th.randint((r.cur_size - 1) * seqs, (B,), device=dev, generator=g)
        for _ in range(5):
            r.sample(B, ids=idx, reuse=True)
        th.cuda.synchronize()
    print("cases:", ",".join(f"seqs{s}_B{b}" for s, b in K9_CASES))
else:
    N, B, NET = 64, 256, [256, 256]
    args = Config(AgentSAC, None, {"env_name": "x", "num_envs": N, "max_step": 1000, "state_dim": S, "action_dim": A, "if_discrete": False})
    args.net_dims, args.batch_size = NET, B
    buf = ring(N)
    args.repeat_times = 8 * B / buf.cur_size                   # 8 update steps
    th.manual_seed(0)
    agent = AgentSAC(NET, S, A, gpu_id=0, args=args)
    for _ in range(2):
        agent.update_net(buf)
    th.cuda.synchronize()
