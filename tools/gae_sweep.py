#!/usr/bin/env python3
"""GAE scan micro-benchmark: every algorithm / look-back tiling on the SURVEY 8d sizes, HIP-event timed,
checked against the exact kernel.  Run on the GPU box:  python tools/gae_sweep.py [--quick]"""
import json
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import ops  # noqa: E402

dev = th.device("cuda:0")


def inputs(H, N):
    g = th.Generator(device=dev).manual_seed(0)
    r, v = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
    u = th.rand((H, N), device=dev, generator=g) < 0.99
    m = th.rand((H, N), device=dev, generator=g) < 0.995
    nv = th.randn(N, device=dev, generator=g)
    return r, u, m, v, nv


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
    return best


def main():
    sizes = [(32, 4096), (200, 4096), (1024, 4096), (2048, 4096), (4096, 4096), (32, 32768), (2048, 16384)]
    tilings = [(4, 2), (4, 4), (4, 8), (8, 2), (8, 4), (8, 8), (8, 16), (16, 2), (16, 4), (16, 8), (2, 4), (2, 8)]
    if "--quick" in sys.argv:
        sizes, tilings = sizes[1:4], [(8, 4), (8, 8), (16, 8)]
    out = []
    for H, N in sizes:
        r, u, m, v, nv = inputs(H, N)
        adv, ret = th.empty_like(r), th.empty_like(r)
        stats = th.zeros(8, dtype=th.float64, device=dev)
        ref, _ = ops.gae_scan(r, u, m, v, nv, 0.99, 0.95, mutate=False, algo="exact")
        ref = ref.clone()
        nbytes = 18 * H * N
        rows = []
        for algo in ("exact", "chunked"):
            if algo == "exact" and H * N > (1 << 24):
                continue
            t = timeit(lambda: ops.gae_scan(r, u, m, v, nv, 0.99, 0.95, mutate=False, algo=algo, adv=adv, ret=ret))
            rows.append((algo, None, None, t))
        for L, W in tilings:
            if L * W > max(4, 2 * H):
                continue
            os.environ["ERL_GAE_LB_L"], os.environ["ERL_GAE_LB_W"] = str(L), str(W)
            t = timeit(lambda: ops.gae_scan(r, u, m, v, nv, 0.99, 0.95, mutate=False, algo="lookback", adv=adv, ret=ret))
            err = ((adv - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
            ts = timeit(lambda: ops.gae_scan(r, u, m, v, nv, 0.99, 0.95, mutate=False, algo="lookback", adv=adv, ret=ret,
                                             stats=stats))
            rows.append(("lookback", L, W, t, err, ts))
        for row in rows:
            t = row[3]
            rec = {"H": H, "N": N, "algo": row[0], "L": row[1], "W": row[2], "us": round(t * 1e6, 2),
                   "GBps": round(nbytes / t / 1e9, 1), "frac_of_8TBps": round(nbytes / t / 8e12, 4)}
            if len(row) > 4:
                rec["max_rel_err_vs_exact"] = float(f"{row[4]:.3e}")
                rec["us_with_stats"] = round(row[5] * 1e6, 2)
            out.append(rec)
            print(json.dumps(rec), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/gae_sweep.json", "w"), indent=1)


if __name__ == "__main__":
    main()
