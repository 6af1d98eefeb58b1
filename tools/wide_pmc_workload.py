#!/usr/bin/env python3
"""Workload for the PMC / kernel-trace passes over the (256, h2[, h3]) minibatch kernels: a few minibatches of each at B = 16384 (S = 64, A = 8).
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o p -- python tools/wide_pmc_workload.py"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import ops  # noqa: E402

dev = th.device("cuda:0")
N, S, A, H, B, T = 4096, int(os.environ.get("WD_S", "64")), int(os.environ.get("WD_A", "8")), 32, 16384, 6
ONLY = os.environ.get("WD_ONLY")          # "128" / "64": that (256, h2) kernel alone; "3": the three-layer ones alone
g = th.Generator(device=dev).manual_seed(0)
avg, std = th.zeros(S, device=dev), th.ones(S, device=dev)
states = th.randn((H, N, S), device=dev, generator=g)
actions = th.randn((H, N, A), device=dev, generator=g)
logprobs = th.randn((H, N), device=dev, generator=g) - 8
adv, ret = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
um = th.rand((H, N), device=dev, generator=g) < 0.995
ids = th.randint(H * N, (T, B), device=dev, generator=g)
for h2 in ((128, 64) if ONLY is None else ((int(ONLY),) if ONLY != "3" else ())):
    Pa, Pc = ops.MlpSpec(S, 256, h2, A, True).count, ops.MlpSpec(S, 256, h2, 1, False).count
    fl = th.randn(Pa + Pc, device=dev, generator=g) * 0.05
    stride, n_slabs = ops.ppo_slab_stride(S, 256, h2, A), ops.ppo_num_slabs(B)
    slabs, rows = th.empty((n_slabs, stride), device=dev), th.empty((T, stride), device=dev)
    m1, m2 = th.zeros_like(fl), th.zeros_like(fl)
    ops.ppo_update(fl, m1, m2, avg, std, avg, std, S, 256, h2, A, states, actions, um, logprobs, adv, ret, ids, 0.25, 0.001, slabs, rows, 1, 1e-4, 3.0)
for h3 in ((64, 128) if ONLY in (None, "3") else ()):
    spec = ops.MlpSpecN([S, 256, 128, h3, A], True)
    Pa, Pc = spec.count, ops.MlpSpecN([S, 256, 128, h3, 1], False).count
    fl = th.randn(Pa + Pc, device=dev, generator=g) * 0.05
    g3 = th.empty(Pa + Pc + 4, device=dev)
    for k in range(T):
        ops.mlpn_ppo_step(fl[:Pa], fl[Pa:], avg, std, avg, std, spec, states, actions, um, logprobs, adv, ret, ids[k], 0.25, 0.001, 1.0 / B, g3)
th.cuda.synchronize()
