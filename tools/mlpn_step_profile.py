#!/usr/bin/env python3
"""The layered (generic-shape) PPO minibatch alone, for rocprofv3:  rocprofv3 --kernel-trace --stats -d out -o mlpn -- python tools/mlpn_step_profile.py [h1,h2,...]"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import ops  # noqa: E402

dev = th.device("cuda:0")
N, S, A, H, B = 4096, 64, 8, 32, 16384
hid = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "128,128").split(",")]
g = th.Generator(device=dev).manual_seed(0)
avg, std = th.zeros(S, device=dev), th.ones(S, device=dev)
states = th.randn((H, N, S), device=dev, generator=g)
actions = th.randn((H, N, A), device=dev, generator=g)
logprobs = th.randn((H, N), device=dev, generator=g) - 8
adv, ret = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
um = th.rand((H, N), device=dev, generator=g) < 0.995
ids = th.randint(H * N, (B,), device=dev, generator=g)
spn = ops.MlpSpecN([S, *hid, A], True)
pcn = ops.MlpSpecN([S, *hid, 1], False).count
fl = th.randn(spn.count + pcn, device=dev, generator=g) * 0.05
gout = th.empty(spn.count + pcn + 4, device=dev)
for _ in range(10):
    ops.mlpn_ppo_step(fl[:spn.count], fl[spn.count:], avg, std, avg, std, spn, states, actions, um, logprobs, adv, ret, ids, 0.25, 0.001,
                      1.0 / B, gout)
th.cuda.synchronize()
e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.mlpn_ppo_step(fl[:spn.count], fl[spn.count:], avg, std, avg, std, spn, states, actions, um, logprobs, adv, ret, ids, 0.25, 0.001,
                      1.0 / B, gout)
e1.record()
th.cuda.synchronize()
print(f"mlpn_ppo_step {hid}: {e0.elapsed_time(e1) * 50:.1f} us per minibatch")
