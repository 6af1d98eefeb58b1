#!/usr/bin/env python3
"""One-workgroup GAE scan (csrc/gae_lookback.hip, gae_tall_kernel): 32 envs per workgroup against 16 (ERL_GAE_TALL_ENVS, read at every
launch), warm / cold behind writes / cold behind reads, by the kernel's own device-clock span; results checked against each other.
    python tools/gae_tall_envs_ab.py > gpurun_out/gae_tall_envs_ab.txt"""
import json
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elegantrl_amd import _hip, ops  # noqa: E402

dev = th.device("cuda:0")
FLUSH = th.zeros(640 << 20, dtype=th.uint8, device=dev)


def span(run, n, pre=None):
    _hip.kernel_span_enable(True)
    for _ in range(n):
        if pre is not None:
            pre()
        run()
    th.cuda.synchronize()
    us, _ = _hip.kernel_span_read(_hip.SPAN_GAE)
    _hip.kernel_span_enable(False)
    return us


def measure(H, N):
    g = th.Generator(device=dev).manual_seed(0)
    r, v = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
    u = th.rand((H, N), device=dev, generator=g) < 0.99
    m = th.rand((H, N), device=dev, generator=g) < 0.995
    nv = th.randn(N, device=dev, generator=g)
    outs = {}
    for envs in (32, 16, 32, 16):
        os.environ["ERL_GAE_TALL_ENVS"] = str(envs)
        adv, ret = th.empty_like(r), th.empty_like(r)
        run = lambda: ops.gae_scan(r, u, m, v, nv, 0.99, 0.95, mutate=False, algo="lookback", adv=adv, ret=ret)   # noqa: E731
        for _ in range(5):
            run()
        th.cuda.synchronize()
        warm = span(run, 40)
        cold_w = span(run, 10, lambda: FLUSH.add_(1))
        cold_r = span(run, 10, lambda: FLUSH.view(th.int32).sum())
        _hip.check_async_faults()
        outs[envs] = (adv.clone(), ret.clone())
        b = 18.0 * H * N
        print(json.dumps({"H": H, "N": N, "envs_per_workgroup": envs, "workgroups": -(-N // envs), "kernel_us": round(warm, 2),
                          "frac": round(b / warm / 1e3 / 8000.0, 4), "cold_behind_writes_us": round(cold_w, 2),
                          "cold_behind_writes_frac": round(b / cold_w / 1e3 / 8000.0, 4), "cold_behind_reads_us": round(cold_r, 2),
                          "cold_behind_reads_frac": round(b / cold_r / 1e3 / 8000.0, 4)}), flush=True)
    d = (outs[32][0] - outs[16][0]).abs().max().item()
    print(json.dumps({"H": H, "N": N, "max_abs_diff_adv_32_vs_16": d, "max_abs_adv": outs[32][0].abs().max().item()}), flush=True)


sizes = [(64, 4096), (128, 4096), (200, 4096), (256, 4096), (200, 2048), (128, 1024), (200, 1000), (200, 8192)]
if len(sys.argv) > 1:
    sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for H, N in sizes:
    measure(H, N)
