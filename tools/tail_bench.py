#!/usr/bin/env python3
"""The optimiser tail at config 4 (128 slabs x 50 837 floats): erl_grad_reduce_f32 + erl_clip_adam_f32 against the one-launch
erl_reduce_clip_adam_f32, HIP-event timed back to back (each variant also behind a K6-sized dirty-L2 producer is what bench.py sees)."""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import ops  # noqa: E402

dev = th.device("cuda:0")
Pa, Pc, n_slabs = 25872, 24961, 128
stride = Pa + Pc + 4
groups = [(0, Pa), (Pa, Pc)]
g = th.Generator(device=dev).manual_seed(0)
slabs = th.randn((n_slabs, stride), device=dev, generator=g)
p, m, v, f = (th.randn(Pa + Pc, device=dev, generator=g), th.zeros(Pa + Pc, device=dev), th.zeros(Pa + Pc, device=dev),
              th.empty(stride, device=dev))


def timeit(fn, iters=50):
    for _ in range(5):
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
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def two():
    ops.grad_reduce(slabs, n_slabs, stride, f)
    ops.clip_adam(p, f, m, v, groups, 5, 1e-4, 3.0)


print(f"grad_reduce            {timeit(lambda: ops.grad_reduce(slabs, n_slabs, stride, f)):7.2f} us")
print(f"clip_adam              {timeit(lambda: ops.clip_adam(p, f, m, v, groups, 5, 1e-4, 3.0)):7.2f} us")
print(f"grad_reduce+clip_adam  {timeit(two):7.2f} us")
print(f"reduce_clip_adam       {timeit(lambda: ops.reduce_clip_adam(slabs, n_slabs, stride, f, p, m, v, groups, 5, 1e-4, 3.0)):7.2f} us")
print(f"reduce_clip_adam(grid) {timeit(lambda: ops.reduce_clip_adam(slabs, n_slabs, stride, f, p, m, v, groups, 5, 1e-4, 3.0, grid_wait=True)):7.2f} us")


def partials():
    ops.grad_reduce_partials(slabs, n_slabs, stride, f, groups)
    ops.clip_adam_partials(p, f, m, v, stride, groups, 5, 1e-4, 3.0)


print(f"reduce_partials        {timeit(lambda: ops.grad_reduce_partials(slabs, n_slabs, stride, f, groups)):7.2f} us")
print(f"clip_adam_partials     {timeit(lambda: ops.clip_adam_partials(p, f, m, v, stride, groups, 5, 1e-4, 3.0)):7.2f} us")
print(f"reduce_partials+clip_adam_partials {timeit(partials):7.2f} us   (the default tail since round 3)")
