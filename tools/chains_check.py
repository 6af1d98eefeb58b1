#!/usr/bin/env python3
"""The two-chain update loop (ERL_PPO_CHAINS=2: one chain of half-chip launches per network on two streams) against the one-chain
loop at the BASELINE config-4 minibatch shape: parameters, Adam moments and gradient rows must be BIT-IDENTICAL; then the loop's time
per minibatch by HIP events, alternating the two forms in one process.    python tools/chains_check.py [update_times]"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import ops  # noqa: E402

dev = th.device("cuda:0")
S, h1, h2, A = (int(x) for x in os.environ.get("K6_SHAPE", "64,128,128,8").split(","))
N, H, B = 4096, 32, int(os.environ.get("K6_B", 16384))
UT = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def main():
    g = th.Generator(device=dev).manual_seed(0)
    sa, sc = ops.MlpSpec(S, h1, h2, A, True), ops.MlpSpec(S, h1, h2, 1, False)
    Pa, Pc = sa.count, sc.count
    flat0 = th.randn(Pa + Pc, device=dev, generator=g) * 0.05
    avg, std = th.zeros(S, device=dev), th.ones(S, device=dev)
    states = th.randn((H, N, S), device=dev, generator=g)
    actions = th.randn((H, N, A), device=dev, generator=g)
    logprobs = th.randn((H, N), device=dev, generator=g) - 8
    adv = th.randn((H, N), device=dev, generator=g)
    ret = th.randn((H, N), device=dev, generator=g)
    um = th.rand((H, N), device=dev, generator=g) < 0.995
    ids = th.randint(H * N, (UT, B), device=dev, generator=g)
    stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)

    def run(chains, flat, m1, m2, slabs, rows):
        os.environ["ERL_PPO_CHAINS"] = str(chains)
        ops.ppo_update(flat, m1, m2, avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs, adv, ret, ids, 0.25, 0.001, slabs, rows, 1,
                       3e-4, 3.0)

    def fresh():
        return flat0.clone(), th.zeros_like(flat0), th.zeros_like(flat0), th.empty((n_slabs, stride), device=dev), th.zeros((UT, stride), device=dev)

    run(1, *fresh())                      # (the first full-chip launch measures the workgroup maps)
    th.cuda.synchronize()
    out = {}
    for chains in (1, 2, 1, 2):
        bufs = fresh()
        run(chains, *bufs)
        th.cuda.synchronize()
        out.setdefault(chains, []).append(bufs)
    ok = True
    for name, i in (("params", 0), ("exp_avg", 1), ("exp_avg_sq", 2), ("gradient rows", 4)):
        a, b = out[1][0][i], out[2][0][i]
        same = th.equal(a.view(th.int32), b.view(th.int32))
        rep = th.equal(out[2][0][i].view(th.int32), out[2][1][i].view(th.int32))
        ok = ok and same and rep
        print(f"{name:14s} one chain == two chains bit for bit: {same}   two chains repeatable: {rep}   max |diff| {float((a - b).abs().max()):.3e}")
    print("moved:", float((out[1][0][0] - flat0).abs().max()))
    for rep in range(3):
        for chains in (1, 2):
            bufs = fresh()
            run(chains, *bufs)
            th.cuda.synchronize()
            e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(chains, *bufs)
            e1.record()
            th.cuda.synchronize()
            print(f"rep {rep} chains {chains}: {e0.elapsed_time(e1) / 5 / UT * 1e3:.2f} us per minibatch ({UT} minibatches per loop)")
    print("BITWISE", "OK" if ok else "MISMATCH")


if __name__ == "__main__":
    main()
