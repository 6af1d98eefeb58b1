#!/usr/bin/env python3
"""Diagnostics (round 5): the minibatch kernel inside the C update loop with EVERY workgroup on one network's code path (a -DERL_K6_EXP=16
build, ERL_K6_ONLY_NET=0: the actor's, 1: the critic's, unset: the normal launch).  The kernel is 110 KB of straight-line code, ~55 KB per
network, and a pair of CUs shares a 64 KB instruction cache: if a box's slow workgroups are the ones whose CU pair runs BOTH paths, the
single-path launches run at the fast boxes' speed there.  Gradients are meaningless in the single-path modes (timing only).
    ERL_HIP_LIB=elegantrl_amd/lib/liberl_hip_e16.so ERL_K6_ONLY_NET=0 python tools/k6_only_net.py"""
import json
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ERL_QUIET", "1")
from elegantrl_amd import _hip, ops  # noqa: E402

dev = th.device("cuda:0")
N, S, A, H, B, h1, h2 = 4096, 64, 8, 32, 16384, 128, 128
g = th.Generator(device=dev).manual_seed(0)
sa, sc = ops.MlpSpec(S, h1, h2, A, True), ops.MlpSpec(S, h1, h2, 1, False)
flat = th.randn(sa.count + sc.count, device=dev, generator=g) * 0.05
avg, std = th.zeros(S, device=dev), th.ones(S, device=dev)
states = th.randn((H, N, S), device=dev, generator=g)
actions = th.randn((H, N, A), device=dev, generator=g)
logprobs = th.randn((H, N), device=dev, generator=g) - 8
adv, ret = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
um = th.rand((H, N), device=dev, generator=g) < 0.995
ids = th.randint(H * N, (40, B), device=dev, generator=g)
stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)
slabs = th.empty((n_slabs, stride), device=dev)
m1, m2, rows = th.zeros_like(flat), th.zeros_like(flat), th.zeros((40, stride), device=dev)
for rep in range(4):
    if rep == 3:
        _hip.k6_timing_enable(4)
    ops.ppo_update(flat, m1, m2, avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs, adv, ret, ids, 0.25, 0.001, slabs, rows, 1 + 40 * rep, 0.0, 3.0)
th.cuda.synchronize()
_hip.k6_timing_enable(False)
_hip.k6_timing_read2()
c = _hip.k6_timing_clocks(False)
w = _hip.k6_wg_summary(_hip.k6_timing_last_records(False)) or {}
print(json.dumps({"only_net": os.environ.get("ERL_K6_ONLY_NET"), "lib": os.path.basename(os.environ.get("ERL_HIP_LIB", "liberl_hip.so")),
                  "us_span": round(c["span_us"] or 0, 2), "workgroup_us": round(c["workgroup_us"], 2), "shader_mhz": round(c["shader_mhz"], 1),
                  "dur_us": w.get("dur_us"), "table": w.get("table")}))
