#!/usr/bin/env python3
"""A/B of the minibatch kernel's workgroup map inside one process (round 5): ERL_K6_WG_MAP=0 (network = blockIdx.y: both networks' code paths
on every XCD) against 1 (XCDs 0-3 the actor's workgroups, 4-7 the critic's; ppo_step.h k6_wg_map), alternating, the kernel inside the C
update loop of config 4's shape.  Prints one JSON line per pass: the unbracketed span of the sampled launches, the workgroup durations,
which networks ran on which XCD, and -- for the first pass of each map -- a checksum of the updated parameters (the maps must agree bit
for bit: the slab a workgroup writes does not depend on where it ran).
    python tools/k6_wg_map_ab.py [passes]"""
import json
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ERL_QUIET", "1")
from elegantrl_amd import _hip, ops  # noqa: E402

dev = th.device("cuda:0")
N, S, A, H, B, h1, h2 = 4096, 64, 8, 32, 16384, 128, 128
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
MAPS = (0, 1, 2)
g = th.Generator(device=dev).manual_seed(0)
sa, sc = ops.MlpSpec(S, h1, h2, A, True), ops.MlpSpec(S, h1, h2, 1, False)
flat0 = th.randn(sa.count + sc.count, device=dev, generator=g) * 0.05
avg, std = th.zeros(S, device=dev), th.ones(S, device=dev)
states = th.randn((H, N, S), device=dev, generator=g)
actions = th.randn((H, N, A), device=dev, generator=g)
logprobs = th.randn((H, N), device=dev, generator=g) - 8
adv, ret = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
um = th.rand((H, N), device=dev, generator=g) < 0.995
ids = th.randint(H * N, (40, B), device=dev, generator=g)
stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)
slabs = th.empty((n_slabs, stride), device=dev)
rows = th.zeros((40, stride), device=dev)


def one_pass(wg_map: int, timed: bool):
    os.environ["ERL_K6_WG_MAP"] = str(wg_map)
    flat, m1, m2 = flat0.clone(), th.zeros_like(flat0), th.zeros_like(flat0)
    if timed:
        _hip.k6_timing_enable(4)
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    ops.ppo_update(flat, m1, m2, avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs, adv, ret, ids, 0.25, 0.001, slabs, rows, 1, 0.0, 3.0)
    e1.record()
    th.cuda.synchronize()
    out = {"wg_map": wg_map, "loop_ms_40_minibatches": round(e0.elapsed_time(e1), 4), "params_checksum": float(flat.double().sum().item()),
           "params_absmax": float(flat.abs().max().item())}
    if timed:
        _hip.k6_timing_enable(False)
        _hip.k6_timing_read2()
        c = _hip.k6_timing_clocks(False)
        w = _hip.k6_wg_summary(_hip.k6_timing_last_records(False)) or {}
        out.update({"us_span": round(c["span_us"] or 0, 2), "workgroup_us": round(c["workgroup_us"], 2), "shader_mhz": round(c["shader_mhz"], 1),
                    "dur_us": w.get("dur_us"), "dur_us_mean_by_xcc": w.get("dur_us_mean_by_xcc")})
    return out


for m in MAPS:
    one_pass(m, False)                                       # warm
for p in range(passes):
    for m in MAPS:
        print(json.dumps(one_pass(m, True)), flush=True)
# the same loop with NO launch sampled: does a launch that leaves per-workgroup records run as long as one that does not?  (half of the
# launches of a timed pass above are sampled; on the pool's slow boxes the bench's sampled launches came out ~9 us above the loop's mean)
for p in range(2):
    for m in MAPS:
        print(json.dumps(dict(one_pass(m, False), sampled=False)), flush=True)
