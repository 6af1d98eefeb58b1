#!/usr/bin/env python3
"""Phase-level cycle breakdown of the K6 PPO minibatch kernel (needs the ERL_PROFILE build:
   make -C elegantrl_amd/csrc EXTRA=-DERL_PROFILE OUT=../lib/liberl_hip_prof.so OBJDIR=build_prof).
   Run on the GPU box:  python tools/ppo_phase_profile.py"""
import ctypes
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import _hip  # noqa: E402

_hip.LIB_PATH = os.environ.get("ERL_HIP_PROF_LIB") or os.path.join(os.path.dirname(_hip.LIB_PATH), "liberl_hip_prof.so")
from elegantrl_amd import ops  # noqa: E402

dev = th.device("cuda:0")
# shape: K6_SHAPE="S,h1,h2,A" (default config 4: 64,128,128,8; config 2 / Pendulum: 3,128,64,1)
S, h1, h2, A = (int(x) for x in os.environ.get("K6_SHAPE", "64,128,128,8").split(","))
N, H, B = 4096, 32, 16384
NAMES = ["prologue (2 global trips, copies, norm_x)", "barrier0", "L1 fwd", "L2 fwd", "out layer", "objective + dstd partials",
         "dZ2 + dZ1", "barrier1 (wave skew)", "stage dZ1/dY + barrier2", "dW1 + db1 + db3", "barrier3 + stage H2/H1 + barrier4",
         "dW3 + barrier5 + stage dZ2 + barrier6", "dW2 + db2", "loss reduce + logs"]
NP = 14

def main():
    lib = _hip.lib()
    lib.erl_debug_set_ppo_profile.argtypes = [ctypes.c_void_p]
    lib.erl_debug_set_ppo_profile.restype = None
    g = th.Generator(device=dev).manual_seed(0)
    sa, sc = ops.MlpSpec(S, h1, h2, A, True), ops.MlpSpec(S, h1, h2, 1, False)
    Pa, Pc = sa.count, sc.count
    flat = th.randn(Pa + Pc, device=dev, generator=g) * 0.05
    avg, std = th.zeros(S, device=dev), th.ones(S, device=dev)
    print(f"shape S={S} net=[{h1},{h2}] A={A} B={B}")
    states = th.randn((H, N, S), device=dev, generator=g)
    actions = th.randn((H, N, A), device=dev, generator=g)
    logprobs = th.randn((H, N), device=dev, generator=g) - 8
    adv = th.randn((H, N), device=dev, generator=g)
    ret = th.randn((H, N), device=dev, generator=g)
    um = th.rand((H, N), device=dev, generator=g) < 0.995
    ids = th.randint(H * N, (B,), device=dev, generator=g)
    if os.environ.get("K6_IDS") == "contiguous":      # buffer rows t * N + n consecutive: id = n * H + t  (how much of the prologue is the gather?)
        ar = th.arange(B, device=dev)
        ids = (ar % N) * H + ar // N
    stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)
    slabs = th.empty((n_slabs, stride), device=dev)
    prof = th.zeros(2 * 8 * 32, dtype=th.int64, device=dev)
    lib.erl_debug_set_ppo_profile(prof.data_ptr())
    run = lambda: ops.ppo_step(flat[:Pa], flat[Pa:], avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs, adv, ret,  # noqa: E731
                               ids, 0.25, 0.001, 1.0 / B, slabs, n_slabs)
    if os.environ.get("K6_LOOP"):
        # through the C update loop (one minibatch, lr = 0): the split-arithmetic kernel then gets its W2 images from the loop
        m1, m2, rows = th.zeros_like(flat), th.zeros_like(flat), th.zeros((1, stride), device=dev)
        run = lambda: ops.ppo_update(flat, m1, m2, avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs, adv, ret,  # noqa: E731
                                     ids.view(1, B), 0.25, 0.001, slabs, rows, 1, 0.0, 3.0)
        print("minibatch kernel launched through erl_ppo_update_dp_f32 (wall time below includes the optimiser tail)")
    lib.erl_debug_set_ppo_profile_block.argtypes = [ctypes.c_int]
    lib.erl_debug_set_ppo_profile_block.restype = None
    for blk in (1, 37, 64, 127):                      # are the workgroups alike?  (total cycles, wave 0 of each net)
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
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    for _ in range(5):
        run()
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    th.cuda.synchronize()
    print(f"wall time per launch of this (instrumented) build: {e0.elapsed_time(e1) * 50:.1f} us")
    full = prof.cpu().view(2, 8, 32)
    if int(full[0, 0, 16]) != 0:                      # extra prologue stamps of the split-arithmetic kernel (cycles since kernel entry, wave 0..3 mean)
        names = {16: "sample id arrived", 17: "row loads issued", 18: "W2 image pieces issued", 19: "W1 arrived + split into its image",
                 1: "(end of prologue)", 2: "(barrier0 passed)", 20: "own row arrived + normalised", 3: "(end of L1 fwd)"}
        for net, nm in enumerate(("actor", "critic")):
            w = full[net, :4].double()
            print(f"--- {nm}: prologue detail (cycles since entry): " + ", ".join(f"{names[k]} {float((w[:, k] - w[:, 0]).mean()):.0f}" for k in (16, 17, 18, 19, 1, 2, 20, 3)))
    if int(full[0, 0, 21]) != 0:                      # -DERL_PROFILE_FINE builds: tile boundaries inside the second layer and the backward
        fn = {3: "(L2 fwd starts)", 21: "L2 tile 1 starts", 22: "L2 tile 2 starts", 23: "L2 tile 3 starts", 24: "L2 MFMA loop over", 4: "(L2 fwd ends)",
              6: "(backward starts)", 29: "dZ2 formed", 25: "bwd tile 1 starts", 26: "bwd tile 2 starts", 27: "bwd tile 3 starts", 28: "bwd MFMA loop over", 7: "(backward ends)"}
        for net, nm in enumerate(("actor", "critic")):
            w = full[net, :4].double()
            print(f"--- {nm}: fine stamps (cycles since entry): " + ", ".join(f"{v} {float((w[:, k] - w[:, 0]).mean()):.0f}" for k, v in fn.items()))
    names, NPk = NAMES, NP
    if int(full[0, 0, 15]) != 0:                      # weight-gradient order 2 (round 6 default): the stamps 9..15 bracket other work
        names = NAMES[:9] + ["A operand of dW1 (dZ1^T rows) + barrier2b", "dW1 + db1 (the dZ2 image rides)", "barrier3 + A operand of dW2 + stage H1 half + barrier4",
                             "dW2 + db2 (H2^T rides; barriers 5, 6)", "dW3 + db3", "loss reduce + logs"]
        NPk = 16
    p = full[:, :, :NPk]
    p = p[:, (p[0, :, 0] != 0)]                        # the 4-wave form stamps waves 0..3 only
    print(f"{p.shape[1]} waves per workgroup")
    for net, name in enumerate(("actor", "critic")):
        d = (p[net, :, 1:] - p[net, :, :-1]).double()
        tot = (p[net, :, NPk - 1] - p[net, :, 0]).double()
        print(f"--- {name}: total cycles per wave min/mean/max = {tot.min():.0f} / {tot.mean():.0f} / {tot.max():.0f}")
        for i, nm in enumerate(names[:NPk - 1]):
            print(f"  {nm:36s} mean {d[:, i].mean():9.0f}  min {d[:, i].min():9.0f}  max {d[:, i].max():9.0f}  ({100 * d[:, i].mean() / tot.mean():5.1f} %)")


if __name__ == "__main__":
    main()
