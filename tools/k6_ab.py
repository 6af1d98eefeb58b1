#!/usr/bin/env python3
"""K6 (erl_ppo_step_f32) alone at the BASELINE config-4 minibatch (B = 16384, S = 64, A = 8, net [128,128]):
HIP-event time per launch over back-to-back launches.  A/B the two kernel forms on the same box:
    ERL_K6_FORM=8 python tools/k6_ab.py ; python tools/k6_ab.py"""
import json
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import ops  # noqa: E402

dev = th.device("cuda:0")
N, S, A, H, B, h1, h2 = 4096, int(os.environ.get("K6_S", 64)), 8, 32, 16384, 128, 128


def main():
    g = th.Generator(device=dev).manual_seed(0)
    sa, sc = ops.MlpSpec(S, h1, h2, A, True), ops.MlpSpec(S, h1, h2, 1, False)
    Pa = sa.count
    flat = th.randn(Pa + sc.count, device=dev, generator=g) * 0.05
    avg, std = th.zeros(S, device=dev), th.ones(S, device=dev)
    states = th.randn((H, N, S), device=dev, generator=g)
    actions = th.randn((H, N, A), device=dev, generator=g)
    logprobs = th.randn((H, N), device=dev, generator=g) - 8
    adv = th.randn((H, N), device=dev, generator=g)
    ret = th.randn((H, N), device=dev, generator=g)
    um = th.rand((H, N), device=dev, generator=g) < 0.995
    ids = th.randint(H * N, (B,), device=dev, generator=g)
    # K6_ROTATE=1: another minibatch of the permutation on every launch, as in the PPO loop (the gathered rows are then cold in
    # L2); K6_ROTATE=2 additionally streams a slab-sized buffer through the caches between launches (what the tail does)
    # K6_BETWEEN=adam|reduce|both: run the loop's other kernels between launches and time K6 alone with per-launch events
    between = os.environ.get("K6_BETWEEN", "")
    rotate = int(os.environ.get("K6_ROTATE", 0))
    idsets = [th.randint(H * N, (B,), device=dev, generator=g) for _ in range(8)] if rotate else [ids]
    scrub = th.empty(26 << 18, device=dev) if rotate == 2 else None
    call = [0]
    stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)
    slabs = th.empty((n_slabs, stride), device=dev)
    def run():
        call[0] += 1
        if scrub is not None:
            scrub.add_(1.0)
        ops.ppo_step(flat[:Pa], flat[Pa:], avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs, adv, ret,
                     idsets[call[0] % len(idsets)], 0.25, 0.001, 1.0 / B, slabs, n_slabs)
    if between:
        fgrad = th.zeros(stride, device=dev)
        m1, m2 = th.zeros_like(flat), th.zeros_like(flat)
        groups = [(0, Pa), (Pa, sc.count)]
        for _ in range(300):
            run()
        th.cuda.synchronize()
        ev = []
        for it in range(400):
            if between in ("reduce", "both"):
                ops.grad_reduce(slabs, n_slabs, stride, fgrad)
            if between in ("adam", "both"):
                ops.clip_adam(flat, fgrad, m1, m2, groups, 5 + it, 1e-6, 3.0)
            e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            ev.append((e0, e1))
        th.cuda.synchronize()
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[100:])
        print(json.dumps({"between": between, "k6_event_us_median": round(t[len(t) // 2], 2), "p10": round(t[len(t) // 10], 2)}))
        return
    for _ in range(10):
        run()
    th.cuda.synchronize()
    best, tot = 1e9, []
    for _ in range(int(os.environ.get("K6_REPS", 5))):
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        th.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / 50
        tot.append(round(t, 2))
        best = min(best, t)
    flops = 2 * B * sum(3 * (S * h1 + h1 * h2 + h2 * o) - 2 * S * h1 for o in (A, 1))
    tail = sorted(tot[len(tot) // 2:])
    print(json.dumps({"form": os.environ.get("ERL_K6_FORM", "auto"), "S": S, "us": tot if len(tot) <= 8 else tot[:3] + ["..."] + tot[-3:],
                      "best_us": round(best, 2), "median_second_half_us": tail[len(tail) // 2],
                      "checksum": float(slabs.double().sum())}))


if __name__ == "__main__":
    main()
