#!/usr/bin/env python3
"""Phase-level cycle breakdown of one step of the persistent rollout kernel (workgroup 0, step 5; needs the ERL_PROFILE build:
   make -C elegantrl_amd/csrc EXTRA=-DERL_PROFILE OUT=../lib/liberl_hip_prof.so OBJDIR=build_prof).  Config 4 shapes."""
import ctypes
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import _hip  # noqa: E402

_hip.LIB_PATH = os.environ.get("ERL_HIP_PROF_LIB") or os.path.join(os.path.dirname(_hip.LIB_PATH), "liberl_hip_prof.so")
from elegantrl_amd.agents import AgentPPO  # noqa: E402
from elegantrl_amd.envs import PendulumVecEnv, SynVecEnv  # noqa: E402
from elegantrl_amd.train import Config  # noqa: E402

NAMES = ["state tile -> registers, states[t], L1 of both nets", "barrier 1", "L2 + output partials", "barrier 2",
         "policy head, env step, value head, noise", "barrier 3", "flags, auto-reset, new state tile", "barrier 4"]


def main():
    lib = _hip.lib()
    lib.erl_debug_set_rollout_fused_profile.argtypes = [ctypes.c_void_p]
    lib.erl_debug_set_rollout_fused_profile.restype = None
    pend = os.environ.get("RF_ENV") == "pendulum"             # config 2: Pendulum, net [128, 64], 200 steps
    N, S, A, H = (4096, 3, 1, 200) if pend else (4096, 64, 8, 32)
    args = Config(AgentPPO, PendulumVecEnv if pend else SynVecEnv,
                  {"env_name": "Pendulum-v1" if pend else "SynVecEnv", "num_envs": N, "max_step": 200 if pend else 1000, "state_dim": S,
                   "action_dim": A, "if_discrete": False})
    args.net_dims, args.horizon_len, args.batch_size = ([128, 64] if pend else [128, 128]), H, 16384
    args.gpu_id = 0
    agent = AgentPPO(args.net_dims, S, A, gpu_id=0, args=args)
    env = PendulumVecEnv(N, max_step=200, gpu_id=0, seed=0) if pend else SynVecEnv(N, S, A, max_step=1000, gpu_id=0, seed=0)
    agent.last_state = env.reset()[0]
    prof = th.zeros(8 * 16, dtype=th.int64, device="cuda:0")
    lib.erl_debug_set_rollout_fused_profile(prof.data_ptr())
    for _ in range(3):
        agent.explore_env(env, H)
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        agent.explore_env(env, H)
    e1.record()
    th.cuda.synchronize()
    print(f"explore_env (instrumented build): {e0.elapsed_time(e1) * 100:.1f} us per {H}-step rollout")
    p = prof.cpu().numpy().reshape(8, 16)
    print("s_memtime ticks per phase of one step, waves 0..7:")
    for i, name in enumerate(NAMES):
        d = p[:, i + 1] - p[:, i]
        print(f"  {name:58s} mean {d.mean():8.0f}  min {d.min():7d}  max {d.max():7d}")
    print(f"  one step, wave 0: {p[0, 8] - p[0, 0]} cycles")


if __name__ == "__main__":
    main()
