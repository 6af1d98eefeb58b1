#!/usr/bin/env python3
"""Cycle stamps inside the fused SAC step's tile kernels (ERL_PROFILE build of the library:
   make -C elegantrl_amd/csrc EXTRA=-DERL_PROFILE OUT=../lib/liberl_hip_prof.so OBJDIR=build_prof).  Runs a few config-3 updates
   and prints the phase durations of workgroup (0, 0), wave 0."""
import ctypes
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import _hip  # noqa: E402

_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "liberl_hip_prof.so")
from elegantrl_amd import ops  # noqa: E402

dev = th.device("cuda:0")
S, A, hidden, E, B = 11, 3, (256, 256), 4, 256
spec = ops.SacSpec(S, A, hidden, E)
g = th.Generator(device=dev).manual_seed(0)
pa = th.randn(spec.actor_count, device=dev, generator=g) * 0.05
pc = th.randn(spec.critic_count, device=dev, generator=g) * 0.05
pt = pc.clone()
alpha = th.full((1,), -1.0, device=dev)
mom = [th.zeros_like(pa), th.zeros_like(pa), th.zeros_like(pc), th.zeros_like(pc), th.zeros(1, device=dev), th.zeros(1, device=dev)]
objs = th.zeros(2, device=dev)
batch = [th.randn(B, S, device=dev, generator=g), th.randn(B, A, device=dev, generator=g).tanh(), th.randn(B, device=dev, generator=g),
         th.ones(B, device=dev), th.ones(B, device=dev), th.randn(B, S, device=dev, generator=g)]
for step in range(1, 6):
    ops.sac_update(spec, pa, pc, pt, alpha, mom, batch, step, gamma=0.99, target_entropy=1.0, tau=5e-3, lr=1e-4, max_norm=3.0, objs_out=objs)
th.cuda.synchronize()
lib = _hip.lib()
buf = (ctypes.c_longlong * 256)()
lib.erl_debug_sac_fused_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.erl_debug_sac_fused_profile.restype = ctypes.c_int
assert lib.erl_debug_sac_fused_profile(buf, 256) == 0
NAMES = {0: ["weight loads issued + image clear", "state rows -> LDS", "L1 mma (K = S)", "L1 epilogue + barrier", "L2 mma (256 x 256)",
             "L2 epilogue + barrier", "head (reduction split)", "tanh / log-prob (16 threads)"],
         1: ["state | action rows -> LDS", "encoder mma + epilogue", "decoder forward mma + GELU (+ backward weights requested)", "output layer (reduction split)",
             "q exchange among the slices (split only)", "labels, dq", "dZ1", "dEnc = W1^T dZ1", "dEnc shares: publish, arrive, last one adds (split only)"]}
print("ERL_SAC_TRAIN_SPLIT =", os.environ.get("ERL_SAC_TRAIN_SPLIT", "(default: split)"))
for slot, names in NAMES.items():
    st = [buf[slot * 32 + i] for i in range(len(names) + 1)]
    print(f"slot {slot}: total {st[-1] - st[0]} cycles (s_memtime = 100 MHz ticks x 1? reported raw)")
    for i, n in enumerate(names):
        print(f"  {n:40s} {st[i + 1] - st[i]:8d}")
