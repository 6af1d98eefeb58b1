#!/usr/bin/env python3
"""Phase-level cycle breakdown of one step of the persistent rollout kernel (needs the ERL_PROFILE build:
   make -C elegantrl_amd/csrc EXTRA=-DERL_PROFILE OUT=../lib/liberl_hip_prof.so OBJDIR=build_prof).
   Run on the GPU box:  python tools/rollout_fused_phase_profile.py"""
import ctypes
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import _hip  # noqa: E402

_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "liberl_hip_prof.so")
from elegantrl_amd.agents import AgentPPO            # noqa: E402
from elegantrl_amd.envs import SynVecEnv             # noqa: E402
from elegantrl_amd.train import Config                # noqa: E402

NAMES = ["phase 0: tiles -> regs, L1 both nets, GELU, H1 write", "barrier 1", "phase 1: L2 + out partials", "barrier 2",
         "phase 2: head + env MFMA | value | next draws", "barrier 3", "phase 3: fold, reset, new tiles", "barrier 4"]


def main():
    lib = _hip.lib()
    lib.erl_debug_set_rollout_fused_profile.argtypes = [ctypes.c_void_p]
    lib.erl_debug_set_rollout_fused_profile.restype = None
    N, S, A, H = 4096, 64, 8, 32
    args = Config(AgentPPO, None, {"env_name": "syn", "num_envs": N, "max_step": 1000, "state_dim": S, "action_dim": A,
                                   "if_discrete": False})
    th.manual_seed(0)
    agent = AgentPPO(args.net_dims, S, A, gpu_id=0, args=args)
    env = SynVecEnv(N, S, A, gpu_id=0)
    agent.last_state = env.reset()[0]
    prof = th.zeros(8 * 16, dtype=th.int64, device="cuda:0")
    lib.erl_debug_set_rollout_fused_profile(prof.data_ptr())
    for _ in range(4):
        agent.explore_env(env, H)
    th.cuda.synchronize()
    p = prof.cpu().view(8, 16)
    d = (p[:, 1:9] - p[:, 0:8]).double()
    print("s_memtime ticks per phase of step 5 (workgroup 0), waves 0..7:")
    for i, nm in enumerate(NAMES):
        print(f"  {nm:54s} mean {d[:, i].mean():7.0f} min {d[:, i].min():7.0f} max {d[:, i].max():7.0f}   per wave {[int(x) for x in d[:, i]]}")
    print("  whole step, wave 0:", int(p[0, 8] - p[0, 0]), " start skew across waves:", int(p[:, 0].max() - p[:, 0].min()))


if __name__ == "__main__":
    main()
