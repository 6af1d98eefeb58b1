#!/usr/bin/env python3
"""Phase-level cycle breakdown of the PPO minibatch kernel for net_dims = (256, h2) (csrc/ppo_step_wd_impl.h).  Needs the ERL_PROFILE build:
   make -C elegantrl_amd/csrc EXTRA=-DERL_PROFILE OUT=../lib/liberl_hip_prof.so OBJDIR=build_prof
   python tools/wide_phase_profile.py            (WD_SHAPE="S,h2,A", default 64,128,8)"""
import ctypes
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import _hip  # noqa: E402

_hip.LIB_PATH = os.environ.get("ERL_HIP_PROF_LIB") or os.path.join(os.path.dirname(_hip.LIB_PATH), "liberl_hip_prof.so")
from elegantrl_amd import ops  # noqa: E402

dev = th.device("cuda:0")
S, h2, A = (int(x) for x in os.environ.get("WD_SHAPE", "64,128,8").split(","))
h1, N, H, B = 256, 4096, 32, 16384
NAMES = ["prologue + W1 image + barrier (0a)", "first layer (8 tiles; GELU' out, W2 quarter 0 in)", "wait + barrier (0b)",
         "layer 2 quarter 0 + barrier", "layer 2 quarter 1 + barrier", "layer 2 quarter 2 + barrier", "layer 2 quarter 3",
         "GELU of layer 2, H2 out", "output layer, objective, dZ2", "dZ1 quarter 3 (gate in)", "barrier, stage dZ1 q3, barrier",
         "dW1 quarter 3", "dZ1 quarter 2", "barrier, stage, W2 quarter 1 requested, barrier", "dW1 quarter 2 + wait + barrier",
         "dZ1 quarter 1 (W2 quarter 0 in)", "barrier, stage, barrier", "dW1 quarter 1 + wait + barrier", "dZ1 quarter 0",
         "barrier, stage, barrier", "dW1 quarter 0 + barrier", "H2 back, H2^T staged, barrier", "dW3 + barrier",
         "stage dZ2 image + H1 quarter 0, barrier", "dZ2^T row tile into registers, barrier", "stage H1 q1, q2; dW2 quarter 0; barrier",
         "stage H1 q3; dW2 quarters 1, 2; barrier", "dW2 quarter 3, db2"]
NP = 29


def main():
    lib = _hip.lib()
    lib.erl_debug_set_ppo_profile.argtypes = [ctypes.c_void_p]
    lib.erl_debug_set_ppo_profile.restype = None
    lib.erl_debug_set_ppo_profile_block.argtypes = [ctypes.c_int]
    lib.erl_debug_set_ppo_profile_block.restype = None
    g = th.Generator(device=dev).manual_seed(0)
    Pa, Pc = ops.MlpSpec(S, h1, h2, A, True).count, ops.MlpSpec(S, h1, h2, 1, False).count
    flat = th.randn(Pa + Pc, device=dev, generator=g) * 0.05
    avg, std = th.zeros(S, device=dev), th.ones(S, device=dev)
    states = th.randn((H, N, S), device=dev, generator=g)
    actions = th.randn((H, N, A), device=dev, generator=g)
    logprobs = th.randn((H, N), device=dev, generator=g) - 8
    adv, ret = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
    um = th.rand((H, N), device=dev, generator=g) < 0.995
    ids = th.randint(H * N, (B,), device=dev, generator=g)
    stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)
    slabs = th.empty((n_slabs, stride), device=dev)
    prof = th.zeros(2 * 8 * 32, dtype=th.int64, device=dev)
    lib.erl_debug_set_ppo_profile(prof.data_ptr())
    m1, m2, rows = th.zeros_like(flat), th.zeros_like(flat), th.zeros((1, stride), device=dev)
    run = lambda: ops.ppo_update(flat, m1, m2, avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs, adv, ret,  # noqa: E731
                                 ids.view(1, B), 0.25, 0.001, slabs, rows, 1, 0.0, 3.0)
    print(f"shape S={S} net=({h1},{h2}) A={A} B={B}; launched through erl_ppo_update_dp_f32 (one minibatch, lr = 0)")
    for blk in (1, 37, 64, 127):
        lib.erl_debug_set_ppo_profile_block(blk)
        prof.zero_()
        for _ in range(3):
            run()
        th.cuda.synchronize()
        q = prof.cpu().view(2, 8, 32)
        print(f"workgroup {blk:3d}: actor {int(q[0, 0, NP - 1] - q[0, 0, 0])} cycles, critic {int(q[1, 0, NP - 1] - q[1, 0, 0])} cycles; "
              f"start skew vs critic {int(q[1, 0, 0] - q[0, 0, 0])}")
    lib.erl_debug_set_ppo_profile_block(0)
    prof.zero_()
    for _ in range(5):
        run()
    th.cuda.synchronize()
    full = prof.cpu().view(2, 8, 32)
    p = full[:, :4, :NP]
    for net, name in enumerate(("actor", "critic")):
        d = (p[net, :, 1:] - p[net, :, :-1]).double()
        tot = (p[net, :, NP - 1] - p[net, :, 0]).double()
        print(f"--- {name}: total cycles per wave min/mean/max = {tot.min():.0f} / {tot.mean():.0f} / {tot.max():.0f}")
        for i, nm in enumerate(NAMES[:NP - 1]):
            print(f"  {nm:52s} mean {d[:, i].mean():9.0f}  min {d[:, i].min():9.0f}  max {d[:, i].max():9.0f}  ({100 * d[:, i].mean() / tot.mean():5.1f} %)")


if __name__ == "__main__":
    main()
