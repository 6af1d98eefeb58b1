#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the BASELINE config-4 shapes (HIP events, best of 3 x 20 launches).
Run on the GPU box:  python tools/kernel_bench.py"""
import json
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import ops  # noqa: E402

dev = th.device("cuda:0")
N, S, A, H, B, h1, h2 = 4096, 64, 8, 32, 16384, 128, 128


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        th.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3 / iters)
    return best * 1e6


def main():
    g = th.Generator(device=dev).manual_seed(0)
    sa, sc = ops.MlpSpec(S, h1, h2, A, True), ops.MlpSpec(S, h1, h2, 1, False)
    Pa, Pc = sa.count, sc.count
    flat = th.randn(Pa + Pc, device=dev, generator=g) * 0.05
    avg, std = th.zeros(S, device=dev), th.ones(S, device=dev)
    states = th.randn((H, N, S), device=dev, generator=g)
    actions = th.randn((H, N, A), device=dev, generator=g)
    logprobs = th.randn((H, N), device=dev, generator=g) - 8
    adv = th.randn((H, N), device=dev, generator=g)
    ret = th.randn((H, N), device=dev, generator=g)
    um = th.rand((H, N), device=dev, generator=g) < 0.995
    ids = th.randint(H * N, (B,), device=dev, generator=g)
    stride = ops.ppo_slab_stride(S, h1, h2, A)
    n_slabs = ops.ppo_num_slabs(B)
    slabs = th.empty((n_slabs, stride), device=dev)
    grads = th.empty(stride, device=dev)
    m1, m2 = th.zeros_like(flat), th.zeros_like(flat)
    out = {}
    out["ppo_step"] = timeit(lambda: ops.ppo_step(flat[:Pa], flat[Pa:], avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs,
                                                   adv, ret, ids, 0.25, 0.001, 1.0 / B, slabs, n_slabs))
    out["grad_reduce"] = timeit(lambda: ops.grad_reduce(slabs, n_slabs, stride, grads))
    p_adam = flat.clone()
    out["clip_adam"] = timeit(lambda: ops.clip_adam(p_adam, grads, m1, m2, [(0, Pa), (Pa, Pc)], 5, 6e-5, 3.0))
    state = th.randn((N, S), device=dev, generator=g)
    env_a = th.empty((N, A), device=dev)
    out["rollout_step"] = timeit(lambda: ops.rollout_step(flat[:Pa], sa, avg, std, state, seed=1, counter=2, out_state=states[0],
                                                           out_action=actions[0], out_logprob=logprobs[0], out_env_action=env_a))
    vals = th.empty((H, N), device=dev)
    out["value_forward_HxN"] = timeit(lambda: ops.value_forward(flat[Pa:], sc, avg, std, states, out=vals))
    out["value_forward_N"] = timeit(lambda: ops.value_forward(flat[Pa:], sc, avg, std, state))
    flops = 2 * (S * h1 + h1 * h2) * 2 + 2 * (h1 * h2) + 0  # rough; the bench uses its own formula
    # config 3 replay ring: 1e6 x (S=11, A=3), num_seqs 1
    M, S3, A3 = 1_000_000, 11, 3
    rs, ra = th.randn((M, 1, S3), device=dev, generator=g), th.randn((M, 1, A3), device=dev, generator=g)
    rr, ru, rm = (th.rand((M, 1), device=dev, generator=g) for _ in range(3))
    for Bq in (256, 4096, 65536, 1048576):
        rid = th.randint(M - 1, (Bq,), device=dev, generator=g)
        out[f"replay_sample_B{Bq}"] = timeit(lambda: ops.replay_sample(rs, ra, rr, ru, rm, rid, M - 1))
    add = 4096
    items = [th.randn((add, 1, S3), device=dev), th.randn((add, 1, A3), device=dev), th.randn((add, 1), device=dev),
             th.rand((add, 1), device=dev) > 0.5, th.rand((add, 1), device=dev) > 0.5]
    out["replay_write_4096rows"] = timeit(lambda: ops.replay_write(rs, ra, rr, ru, rm, items, M - 1000))
    # generic-shape path (own MFMA GEMMs) on the config-4 buffers
    for tag, hid in (("128x128", [128, 128]), ("256x128", [256, 128]), ("256x128x64", [256, 128, 64])):
        spn = ops.MlpSpecN([S, *hid, A], True)
        pcn = ops.MlpSpecN([S, *hid, 1], False).count
        fl = th.randn(spn.count + pcn, device=dev, generator=g) * 0.05
        gout = th.empty(spn.count + pcn + 4, device=dev)
        out[f"mlpn_ppo_step_{tag}"] = timeit(lambda: ops.mlpn_ppo_step(fl[:spn.count], fl[spn.count:], avg, std, avg, std, spn, states,
                                                                       actions, um, logprobs, adv, ret, ids, 0.25, 0.001, 1.0 / B, gout),
                                             iters=10)
    # SAC update step (config 3 shapes): one erl_sac_update_f32 call
    for tag, hid, Bq in (("64x32_B256", [64, 32], 256), ("256x256_B256", [256, 256], 256), ("256x256_B1024", [256, 256], 1024)):
        spec = ops.SacSpec(S3, A3, hid, 4)
        pa = th.randn(spec.actor_count, device=dev, generator=g) * 0.1
        pc = th.randn(spec.critic_count, device=dev, generator=g) * 0.1
        pt, al = pc.clone(), th.full((1,), -1.0, device=dev)
        mom = [th.zeros_like(pa), th.zeros_like(pa), th.zeros_like(pc), th.zeros_like(pc), th.zeros(1, device=dev), th.zeros(1, device=dev)]
        rid = th.randint(M - 1, (Bq,), device=dev, generator=g)
        batch, _ = ops.replay_sample(rs, ra, rr, ru, rm, rid, M - 1)
        objs = th.zeros(2, device=dev)
        out[f"sac_update_{tag}"] = timeit(lambda: ops.sac_update(spec, pa, pc, pt, al, mom, batch, 3, gamma=0.99, target_entropy=1.1,
                                                                  tau=5e-3, lr=1e-4, max_norm=3.0, objs_out=objs), iters=10)
    out = {k: round(v, 2) for k, v in out.items()}
    print(json.dumps(out))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/kernel_bench.json", "w"))


if __name__ == "__main__":
    main()
