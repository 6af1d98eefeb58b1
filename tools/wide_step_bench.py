#!/usr/bin/env python3
"""The fused PPO minibatch kernel for net_dims = (256, h2) (csrc/ppo_step_wd_impl.h) next to the layered path it replaces, at a full-size
minibatch (B = 16384 of 32 x 4096, S = 64, A = 8):  python tools/wide_step_bench.py [h2]  (rocprofv3 --kernel-trace --stats around it
gives the kernel times)."""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import _hip, ops  # noqa: E402

dev = th.device("cuda:0")
N, S, A, H, B, T = 4096, 64, 8, 32, 16384, 40
h1, h2 = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = th.Generator(device=dev).manual_seed(0)
avg, std = th.zeros(S, device=dev), th.ones(S, device=dev)
states = th.randn((H, N, S), device=dev, generator=g)
actions = th.randn((H, N, A), device=dev, generator=g)
logprobs = th.randn((H, N), device=dev, generator=g) - 8
adv, ret = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
um = th.rand((H, N), device=dev, generator=g) < 0.995
ids = th.randint(H * N, (T, B), device=dev, generator=g)
spn = ops.MlpSpecN([S, h1, h2, A], True)
Pa, Pc = spn.count, ops.MlpSpecN([S, h1, h2, 1], False).count
fl0 = th.randn(Pa + Pc, device=dev, generator=g) * 0.05
stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)
slabs, rows = th.empty((n_slabs, stride), device=dev), th.empty((T, stride), device=dev)
e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)


def fused():
    fl, m1, m2 = fl0.clone(), th.zeros_like(fl0), th.zeros_like(fl0)
    ops.ppo_update(fl, m1, m2, avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs, adv, ret, ids, 0.25, 0.001, slabs, rows, 1, 1e-4, 3.0)


def layered():
    fl, m1, m2 = fl0.clone(), th.zeros_like(fl0), th.zeros_like(fl0)
    gout = th.empty(Pa + Pc + 4, device=dev)
    for k in range(T):
        ops.mlpn_ppo_step(fl[:Pa], fl[Pa:], avg, std, avg, std, spn, states, actions, um, logprobs, adv, ret, ids[k], 0.25, 0.001, 1.0 / B, gout)
        ops.clip_adam(fl, gout, m1, m2, [(0, Pa), (Pa, Pc)], k + 1, 1e-4, 3.0)


for name, fn in (("fused minibatch kernel + slab reduction + clip/Adam (C loop)", fused), ("layered step + clip/Adam (python loop)", layered)):
    fn()
    th.cuda.synchronize()
    e0.record()
    fn()
    e1.record()
    th.cuda.synchronize()
    print(f"net ({h1},{h2}) S={S} A={A} B={B}: {name}: {e0.elapsed_time(e1) * 1000 / T:.1f} us per minibatch")
_hip.k6_timing_enable(1)
fused()
th.cuda.synchronize()
ev, sp, n = _hip.k6_timing_read2()
print(f"minibatch kernel alone: {sp / n * 1e6:.1f} us by its own device clock, {ev / n * 1e6:.1f} us inside a HIP-event bracket ({n} launches)")
_hip.k6_timing_enable(0)

# three hidden layers, (256, 128, h3): the fused kernel sits behind erl_mlpn_ppo_step_f32 (ERL_WIDE_FUSED=0 in the environment = the layered step)
for h3 in (64, 128):
    spec3 = ops.MlpSpecN([S, 256, 128, h3, A], True)
    Pa3, Pc3 = spec3.count, ops.MlpSpecN([S, 256, 128, h3, 1], False).count
    fl3 = th.randn(Pa3 + Pc3, device=dev, generator=g) * 0.05
    g3 = th.empty(Pa3 + Pc3 + 4, device=dev)
    run3 = lambda k: ops.mlpn_ppo_step(fl3[:Pa3], fl3[Pa3:], avg, std, avg, std, spec3, states, actions, um, logprobs, adv, ret, ids[k], 0.25, 0.001,  # noqa: E731
                                       1.0 / B, g3)
    for k in range(5):
        run3(k)
    th.cuda.synchronize()
    e0.record()
    for k in range(T):
        run3(k)
    e1.record()
    th.cuda.synchronize()
    mode = "layered step (ERL_WIDE_FUSED=0)" if os.environ.get("ERL_WIDE_FUSED") == "0" else "fused kernel + image build + slab reduction"
    print(f"net (256,128,{h3}) S={S} A={A} B={B}: erl_mlpn_ppo_step_f32, {mode}: {e0.elapsed_time(e1) * 1000 / T:.1f} us per minibatch (no optimiser)")
