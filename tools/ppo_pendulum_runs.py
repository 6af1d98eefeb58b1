import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elegantrl_amd import train_agent
from elegantrl_amd.agents import AgentPPO
from elegantrl_amd.envs import PendulumVecEnv
from elegantrl_amd.train import Config
for seed in range(6):
    for iters in (60, 100):
        args = Config(AgentPPO, PendulumVecEnv, {"env_name": "Pendulum-v1", "num_envs": 1024, "max_step": 200, "state_dim": 3,
                                                 "action_dim": 1, "if_discrete": False})
        args.net_dims = [128, 64]
        args.horizon_len, args.batch_size, args.repeat_times = 200, 4096, 4096 * 16 / 200
        args.gamma, args.reward_scale, args.learning_rate = 0.97, 2 ** -2, 4e-4
        args.break_step, args.eval_per_step, args.eval_times = 200 * iters, 200 * 10, 8
        args.cwd, args.gpu_id, args.random_seed = tempfile.mkdtemp(), 0, seed
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            train_agent(args, if_single_process=True)
        rec = np.load(os.path.join(args.cwd, "recorder.npy"))
        print(seed, iters, np.round(rec[:, 1], 0).tolist(), flush=True)
