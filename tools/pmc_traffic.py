#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter set per pass):
runs the GAE look-back scan at 2048 x 4096 and one K6 minibatch a few times each.
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o pmc -- python tools/pmc_traffic.py"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import ops  # noqa: E402

dev = th.device("cuda:0")
g = th.Generator(device=dev).manual_seed(0)
H, N = 2048, 4096
r, v = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
u = th.rand((H, N), device=dev, generator=g) < 0.99
m = th.rand((H, N), device=dev, generator=g) < 0.995
nv = th.randn(N, device=dev, generator=g)
adv, ret = th.empty_like(r), th.empty_like(r)
for _ in range(5):
    ops.gae_scan(r, u, m, v, nv, 0.99, 0.95, mutate=False, algo="lookback", adv=adv, ret=ret)
th.cuda.synchronize()

N, S, A, H, B, h1, h2 = 4096, 64, 8, 32, 16384, 128, 128
sa, sc = ops.MlpSpec(S, h1, h2, A, True), ops.MlpSpec(S, h1, h2, 1, False)
Pa, Pc = sa.count, sc.count
flat = th.randn(Pa + Pc, device=dev, generator=g) * 0.05
avg, std = th.zeros(S, device=dev), th.ones(S, device=dev)
states = th.randn((H, N, S), device=dev, generator=g)
actions = th.randn((H, N, A), device=dev, generator=g)
logprobs = th.randn((H, N), device=dev, generator=g) - 8
adv2, ret2 = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
um = th.rand((H, N), device=dev, generator=g) < 0.995
ids = th.randint(H * N, (B,), device=dev, generator=g)
stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)
slabs = th.empty((n_slabs, stride), device=dev)
# through the C update loop (5 minibatches, lr = 0): that is how the agent launches the minibatch kernel -- the split-arithmetic
# kernel gets its W2 images from the loop; K6_ARITH=f32 measures the fp32-MFMA kernel instead
if os.environ.get("K6_ARITH"):
    ops.ppo_set_arith(os.environ["K6_ARITH"])
m1, m2, rows = th.zeros_like(flat), th.zeros_like(flat), th.zeros((5, stride), device=dev)
ops.ppo_update(flat, m1, m2, avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs, adv2, ret2, ids.repeat(5, 1), 0.25, 0.001,
               slabs, rows, 1, 0.0, 3.0)
th.cuda.synchronize()
