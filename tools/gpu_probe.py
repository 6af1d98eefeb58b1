"""Micro-benchmarks of the individual kernels on the GPU box (writes gpurun_out/probe.json).
Not part of the product or the tests; used while developing to decide what to optimise next."""
import json
import os
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import _hip, ops  # noqa: E402

dev = th.device("cuda:0")
out = {"device": th.cuda.get_device_name(0)}


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def section(name):
    def deco(fn):
        try:
            t0 = time.time()
            out[name] = fn()
            print(f"[{name}] ok in {time.time() - t0:.1f}s: {json.dumps(out[name])[:600]}", flush=True)
        except Exception as e:  # keep going: one call should yield as much information as possible
            out[name] = {"error": repr(e)}
            print(f"[{name}] FAILED: {e!r}", flush=True)
        return fn
    return deco


@section("info")
def _():
    cu, lds = _hip.device_info()
    return {"cus": cu, "lds": lds, "mfma_selftest_err": _hip.selftest_mfma()}


@section("gae")
def _():
    res = []
    for H, N in [(32, 4096), (128, 4096), (200, 4096), (1024, 4096), (2048, 4096), (32, 32768), (4096, 4096)]:
        g = th.Generator(device=dev).manual_seed(0)
        r = th.randn((H, N), device=dev, generator=g)
        v = th.randn((H, N), device=dev, generator=g)
        u = th.rand((H, N), device=dev, generator=g) < 0.99
        m = th.rand((H, N), device=dev, generator=g) < 0.995
        nv = th.randn(N, device=dev, generator=g)
        adv, ret = th.empty_like(r), th.empty_like(r)
        for algo in ("exact", "chunked"):
            if algo == "exact" and H * N > (1 << 23):
                continue
            t = timeit(lambda: ops.gae_scan(r, u, m, v, nv, 0.99, 0.95, algo=algo, mutate=False, adv=adv, ret=ret))
            res.append({"H": H, "N": N, "algo": algo, "us": t * 1e6, "GBps": 18.0 * H * N / t / 1e9})
    return res


@section("stream_copy_ceiling")
def _():
    n = 1 << 28
    a = th.empty(n, dtype=th.float32, device=dev)
    b = th.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    return {"bytes": 8 * n, "GBps": 8.0 * n / t / 1e9}


def make_net(S, h1, h2, out, with_std):
    spec = ops.MlpSpec(S, h1, h2, out, with_std)
    P = th.randn(spec.count, device=dev) * 0.05
    return spec, P, th.zeros(S, device=dev), th.ones(S, device=dev)


@section("value_forward")
def _():
    S, h1, h2 = 64, 128, 128
    spec, P, avg, sd = make_net(S, h1, h2, 1, False)
    res = []
    for rows in (4096, 131072, 1 << 20):
        x = th.randn((rows, S), device=dev)
        o = th.empty(rows, device=dev)
        t = timeit(lambda: ops.value_forward(P, spec, avg, sd, x, out=o))
        fl = 2.0 * rows * (S * h1 + h1 * h2 + h2)
        res.append({"rows": rows, "us": t * 1e6, "TFLOPs": fl / t / 1e12})
    return res


@section("rollout_step")
def _():
    S, h1, h2, A, N = 64, 128, 128, 8, 4096
    spec, P, avg, sd = make_net(S, h1, h2, A, True)
    x = th.randn((N, S), device=dev)
    o_s, o_a, o_e, o_l = th.empty((N, S), device=dev), th.empty((N, A), device=dev), th.empty((N, A), device=dev), th.empty(N, device=dev)
    t = timeit(lambda: ops.rollout_step(P, spec, avg, sd, x, seed=1, counter=2, out_state=o_s, out_action=o_a, out_logprob=o_l,
                                        out_env_action=o_e), iters=50)
    return {"N": N, "us": t * 1e6}


@section("ppo_step")
def _():
    S, h1, h2, A, H, N, B = 64, 128, 128, 8, 32, 4096, 16384
    sa, Pa_, avg, sd = make_net(S, h1, h2, A, True)
    sc, Pc_, _, _ = make_net(S, h1, h2, 1, False)
    stride = ops.ppo_slab_stride(S, h1, h2, A)
    states = th.randn((H, N, S), device=dev)
    actions = th.randn((H, N, A), device=dev)
    um = th.rand((H, N), device=dev) < 0.99
    lp = th.randn((H, N), device=dev) - 8
    adv, rs = th.randn((H, N), device=dev), th.randn((H, N), device=dev)
    ids = th.randint(H * N, (B,), device=dev)
    res = []
    for n_slabs in (64, 128, 256):
        slabs = th.empty((n_slabs, stride), device=dev)
        flat = th.empty(stride, device=dev)
        t1 = timeit(lambda: ops.ppo_step(Pa_, Pc_, avg, sd, avg, sd, S, h1, h2, A, states, actions, um, lp, adv, rs, ids, 0.25,
                                         1e-3, 1.0 / B, slabs, n_slabs))
        t2 = timeit(lambda: ops.grad_reduce(slabs, n_slabs, stride, flat))
        fl = B * 2.0 * 3 * ((S * h1 + h1 * h2 + h2 * A) + (S * h1 + h1 * h2 + h2))
        res.append({"n_slabs": n_slabs, "step_us": t1 * 1e6, "reduce_us": t2 * 1e6, "TFLOPs_step": fl / t1 / 1e12})
    P = th.cat([Pa_, Pc_])
    M1, M2 = th.zeros_like(P), th.zeros_like(P)
    t3 = timeit(lambda: ops.clip_adam(P, flat, M1, M2, [(0, sa.count), (sa.count, sc.count)], 1, 1e-4, 3.0))
    res.append({"adam_us": t3 * 1e6})
    return res


@section("gather_replay")
def _():
    H, N, S, A, B = 32, 4096, 64, 8, 16384
    states, actions = th.randn((H, N, S), device=dev), th.randn((H, N, A), device=dev)
    um = th.rand((H, N), device=dev) < 0.99
    lp, adv, rs = (th.randn((H, N), device=dev) for _ in range(3))
    ids = th.randint(H * N, (B,), device=dev)
    t = timeit(lambda: ops.ppo_gather(states, actions, um, lp, adv, rs, ids))
    res = {"ppo_gather_us": t * 1e6, "ppo_gather_GBps": B * (2 * (4 * S + 4 * A + 13) + 8) / t / 1e9}
    max_size, S, A = 1000000, 11, 3
    bs, ba = th.randn((max_size, 1, S), device=dev), th.randn((max_size, 1, A), device=dev)
    br, bu, bm = (th.randn((max_size, 1), device=dev) for _ in range(3))
    for B in (256, 4096, 65536):
        ids = th.randint(max_size - 1, (B,), device=dev)
        t = timeit(lambda: ops.replay_sample(bs, ba, br, bu, bm, ids, max_size - 1))
        res[f"replay_sample_B{B}_us"] = t * 1e6
        res[f"replay_sample_B{B}_GBps"] = B * (2 * (2 * S + A + 3) * 4 + 8) / t / 1e9
    return res


os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/probe.json", "w") as f:
    json.dump(out, f, indent=1)
print("wrote gpurun_out/probe.json")
