#!/usr/bin/env python3
"""How the small-output GEMM (gemm_small_kernel, the SAC step's 256-row layers) scales with the reduction length: the layered
value pre-pass (normalise + Linear(K, 256) + GELU + Linear(256, 1)) on 256 rows for several K, HIP events over back-to-back
calls.  The K -> 256 layer is the only part that changes."""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import ops  # noqa: E402

dev = th.device("cuda:0")
rows = int(os.environ.get("ROWS", 256))
for K in (16, 64, 128, 256, 512, 1024):
    spec = ops.MlpSpecN([K, 256, 1], False)
    params = th.randn(spec.count, device=dev) * 0.05
    avg, std = th.zeros(K, device=dev), th.ones(K, device=dev)
    x = th.randn(rows, K, device=dev)
    out = th.empty(rows, device=dev)
    for _ in range(20):
        ops.mlpn_value_forward(params, spec, avg, std, x, out)
    th.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            ops.mlpn_value_forward(params, spec, avg, std, x, out)
        e1.record()
        th.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 10)
    print(f"rows {rows}  K {K:5d}: {best:7.2f} us per (normalise + K->256 + 256->1)")
