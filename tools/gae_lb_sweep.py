#!/usr/bin/env python3
"""Tiling sweep of the look-back GAE scan at the sizes BASELINE's configs actually use (round 5: 200 x 4096 ran at 0.16 of the HBM peak,
32 x 4096 at 0.03): kernel_us = the kernel's own device-clock span (erl_kernel_span_*), per (L steps per lane, W waves per workgroup;
the library reads ERL_GAE_LB_L / ERL_GAE_LB_W at every launch), next to the lane-per-env EXACT kernel.
    python tools/gae_lb_sweep.py > gpurun_out/gae_lb_sweep.txt"""
import json
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elegantrl_amd import _hip, ops  # noqa: E402

dev = th.device("cuda:0")
FLUSH = th.zeros(640 << 20, dtype=th.uint8, device=dev)


def measure(H, N, algo, L=None, W=None):
    for k, v in (("ERL_GAE_LB_L", L), ("ERL_GAE_LB_W", W)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    g = th.Generator(device=dev).manual_seed(0)
    r, v = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
    u = th.rand((H, N), device=dev, generator=g) < 0.99
    m = th.rand((H, N), device=dev, generator=g) < 0.995
    nv = th.randn(N, device=dev, generator=g)
    adv, ret = th.empty_like(r), th.empty_like(r)
    run = lambda: ops.gae_scan(r, u, m, v, nv, 0.99, 0.95, mutate=False, algo=algo, adv=adv, ret=ret)   # noqa: E731
    for _ in range(5):
        run()
    th.cuda.synchronize()
    _hip.kernel_span_enable(True)
    for _ in range(40):
        run()
    us, n = _hip.kernel_span_read(_hip.SPAN_GAE)
    _hip.kernel_span_enable(False)
    # cold: a 640 MB buffer rewritten between calls (the warm figure re-reads inputs that fit the 256 MB Infinity Cache)
    _hip.kernel_span_enable(True)
    for _ in range(10):
        FLUSH.add_(1)
        run()
    th.cuda.synchronize()
    cold, n = _hip.kernel_span_read(_hip.SPAN_GAE)
    _hip.kernel_span_enable(False)
    # cold behind READS: the same buffer summed instead of rewritten -- the caches are full of CLEAN lines, nothing is written back while
    # the scan runs (behind writes the scan shares HBM with up to 256 MB of the writer's dirty lines leaving the Infinity Cache)
    _hip.kernel_span_enable(True)
    for _ in range(10):
        FLUSH.view(th.int32).sum()
        run()
    th.cuda.synchronize()
    cold_r, n = _hip.kernel_span_read(_hip.SPAN_GAE)
    _hip.kernel_span_enable(False)
    _hip.check_async_faults()
    print(json.dumps({"H": H, "N": N, "algo": algo, "L": L, "W": W, "kernel_us": round(us, 2), "GBps": round(18.0 * H * N / us / 1e3, 1),
                      "frac": round(18.0 * H * N / us / 1e3 / 8000.0, 4), "cold_kernel_us": round(cold, 2),
                      "cold_frac": round(18.0 * H * N / cold / 1e3 / 8000.0, 4), "cold_behind_reads_us": round(cold_r, 2),
                      "cold_behind_reads_frac": round(18.0 * H * N / cold_r / 1e3 / 8000.0, 4)}), flush=True)


combos = [(None, None), (2, 8), (2, 16), (4, 4), (4, 8), (4, 16), (8, 4), (8, 8), (8, 16), (16, 4), (16, 8)]
sizes = [(32, 4096), (128, 4096), (200, 4096), (32, 32768), (512, 4096), (1024, 4096), (2048, 4096)]
if len(sys.argv) > 1:
    sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for H, N in sizes:
    measure(H, N, "exact")
    for L, W in combos:
        measure(H, N, "lookback", L, W)
