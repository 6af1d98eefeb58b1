#!/usr/bin/env python3
"""Phase-level cycle breakdown of the latency-form rollout kernel (needs the ERL_PROFILE build:
   make -C elegantrl_amd/csrc EXTRA=-DERL_PROFILE OUT=../lib/liberl_hip_prof.so OBJDIR=build_prof).
   Run on the GPU box:  python tools/rollout_phase_profile.py"""
import ctypes
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import _hip  # noqa: E402

_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "liberl_hip_prof.so")
from elegantrl_amd import ops  # noqa: E402

dev = th.device("cuda:0")
N, S, A, h1, h2 = 4096, 64, 8, 128, 128
NAMES = ["issue loads + philox, wait for all loads", "normalise", "L1 MFMA + GELU + T1 write", "barrier 1", "L2 + out MFMA + PS write",
         "barrier 2", "reduce + sample + stores (wave 0)"]


def main():
    lib = _hip.lib()
    lib.erl_debug_set_rollout_profile.argtypes = [ctypes.c_void_p]
    lib.erl_debug_set_rollout_profile.restype = None
    g = th.Generator(device=dev).manual_seed(0)
    sa = ops.MlpSpec(S, h1, h2, A, True)
    flat = th.randn(sa.count, device=dev, generator=g) * 0.05
    avg, std = th.zeros(S, device=dev), th.ones(S, device=dev)
    state = th.randn((N, S), device=dev, generator=g)
    o_s, o_a, o_l, o_e = th.empty((N, S), device=dev), th.empty((N, A), device=dev), th.empty(N, device=dev), th.empty((N, A), device=dev)
    prof = th.zeros(8 * 16, dtype=th.int64, device=dev)
    lib.erl_debug_set_rollout_profile(prof.data_ptr())
    for i in range(5):
        ops.rollout_step(flat, sa, avg, std, state, seed=1, counter=i, out_state=o_s, out_action=o_a, out_logprob=o_l, out_env_action=o_e)
    th.cuda.synchronize()
    p = prof.cpu().view(8, 16)
    d = (p[:, 1:8] - p[:, 0:7]).double()
    print("s_memtime ticks per phase, waves 0..7:")
    for i, nm in enumerate(NAMES):
        row = d[:, i] if i < 6 else d[:1, i]
        print(f"  {nm:44s} mean {row.mean():8.0f} min {row.min():8.0f} max {row.max():8.0f}")
    print("  total wave 0:", int(p[0, 7] - p[0, 0]), " start skew across waves:", int(p[:, 0].max() - p[:, 0].min()))


if __name__ == "__main__":
    main()
