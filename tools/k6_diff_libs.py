#!/usr/bin/env python3
"""K6 debugging aid: run the stand-alone minibatch kernel (erl_ppo_step_f32) of the library named by ERL_HIP_LIB on a fixed seeded
case and save the reduced gradient row; with two .npy files as arguments, print the largest difference per parameter block.
    ERL_HIP_LIB=.../liberl_hip_a.so python tools/k6_diff_libs.py a.npy; ERL_HIP_LIB=... python tools/k6_diff_libs.py b.npy
    python tools/k6_diff_libs.py a.npy b.npy"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S, h1, h2, A, B, H, N = (int(x) for x in os.environ.get("K6_CASE", "64,128,128,8,64,9,50").split(","))


def blocks(out):
    o, res = 0, []
    for nm, n in (("W1", h1 * S), ("b1", h1), ("W2", h2 * h1), ("b2", h2), ("W3", out * h2), ("b3", out)) + ((("std", out),) if out == A else ()):
        res.append((nm, o, n))
        o += n
    return res, o


if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    ba, Pa = blocks(A)
    bc, Pc = blocks(1)
    for net, off, bl in (("actor", 0, ba), ("critic", Pa, bc)):
        for nm, o, n in bl:
            x, y = a[off + o:off + o + n], b[off + o:off + o + n]
            print(f"{net:6s} {nm:3s} max|a-b| {np.abs(x - y).max():.3e}  max|a| {np.abs(x).max():.3e}  first bad {int(np.argmax(np.abs(x - y) > 1e-6 * (1e-9 + np.abs(x).max())))}")
    print("logs", a[Pa + Pc:Pa + Pc + 4], b[Pa + Pc:Pa + Pc + 4])
    sys.exit(0)

import torch as th
from elegantrl_amd import ops
dev = th.device("cuda:0")
g = th.Generator(device=dev).manual_seed(0)
sa, sc = ops.MlpSpec(S, h1, h2, A, True), ops.MlpSpec(S, h1, h2, 1, False)
Pa = sa.count
flat = th.randn(Pa + sc.count, device=dev, generator=g) * 0.1
avg, std = th.randn(S, device=dev, generator=g) * 0.1, th.rand(S, device=dev, generator=g) + 0.5
states = th.randn((H, N, S), device=dev, generator=g)
actions = th.randn((H, N, A), device=dev, generator=g)
logprobs = th.randn((H, N), device=dev, generator=g) - 8
adv, ret = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
um = th.rand((H, N), device=dev, generator=g) < 0.9
ids = th.randint(H * N, (B,), device=dev, generator=g)
stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)
slabs = th.full((n_slabs, stride), float("nan"), device=dev)
if os.environ.get("K6_LOOP"):
    m1, m2, rows = th.zeros_like(flat), th.zeros_like(flat), th.zeros((1, stride), device=dev)
    ops.ppo_update(flat, m1, m2, avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs, adv, ret, ids.view(1, B), 0.25, 0.001, slabs, rows, 1, 0.0, 3.0)
    out = rows[0]
else:
    ops.ppo_step(flat[:Pa], flat[Pa:], avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs, adv, ret, ids, 0.25, 0.001, 1.0 / B, slabs, n_slabs)
    out = th.zeros(stride, device=dev)
    ops.grad_reduce(slabs, n_slabs, stride, out)
np.save(sys.argv[1], out.cpu().numpy())
