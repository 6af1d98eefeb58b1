#!/usr/bin/env python3
"""Does PPO learn the same with the split-arithmetic minibatch kernel as with the fp32-MFMA one?  Pendulum-v1 (1024 envs, net [128,64],
the hyper-parameters of tools/ppo_pendulum_runs.py), a few seeds x 80 iterations per arithmetic; prints the evaluation returns along
training and the final ones side by side.  Run on the GPU box:  python tools/k6_arith_learning.py"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import ops, train_agent  # noqa: E402
from elegantrl_amd.agents import AgentPPO  # noqa: E402
from elegantrl_amd.envs import PendulumVecEnv  # noqa: E402
from elegantrl_amd.train import Config  # noqa: E402

SEEDS, ITERS = range(4), 80
final = {}
for arith in ("f32", "split"):
    ops.ppo_set_arith(arith)
    assert ops.ppo_arith_in_use(3, 128, 64, 1) == arith
    for seed in SEEDS:
        args = Config(AgentPPO, PendulumVecEnv, {"env_name": "Pendulum-v1", "num_envs": 1024, "max_step": 200, "state_dim": 3,
                                                 "action_dim": 1, "if_discrete": False})
        args.net_dims = [128, 64]
        args.horizon_len, args.batch_size, args.repeat_times = 200, 4096, 4096 * 16 / 200
        args.gamma, args.reward_scale, args.learning_rate = 0.97, 2 ** -2, 4e-4
        args.break_step, args.eval_per_step, args.eval_times = 200 * ITERS, 200 * 10, 8
        args.cwd, args.gpu_id, args.random_seed = tempfile.mkdtemp(), 0, seed
        with contextlib.redirect_stdout(io.StringIO()):
            train_agent(args, if_single_process=True)
        rec = np.load(os.path.join(args.cwd, "recorder.npy"))
        final[(arith, seed)] = float(rec[-1, 1])
        print(f"{arith:5s} seed {seed}: evaluation return along training {np.round(rec[:, 1], 0).tolist()}", flush=True)
print("final evaluation return (Pendulum-v1, higher is better; a random policy scores about -1200):")
for seed in SEEDS:
    print(f"  seed {seed}: fp32 MFMA {final[('f32', seed)]:8.1f}   split bf16 {final[('split', seed)]:8.1f}")
print(f"  mean   : fp32 MFMA {np.mean([final[('f32', s)] for s in SEEDS]):8.1f}   split bf16 {np.mean([final[('split', s)] for s in SEEDS]):8.1f}")
