"""Shared test helpers (load golden fixtures into oracle structures)."""
import os

import numpy as np

from oracle import ppo_numpy as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def mlp_from(g, prefix, dt=np.float32) -> O.Mlp:
    ws, bs = [], []
    i = 0
    while f"{prefix}.net.{i}.weight" in g:
        ws.append(g[f"{prefix}.net.{i}.weight"].astype(dt))
        bs.append(g[f"{prefix}.net.{i}.bias"].astype(dt))
        i += 2
    asl = g.get(f"{prefix}.action_std_log")
    return O.Mlp(ws, bs, g[f"{prefix}.state_avg"].astype(dt), g[f"{prefix}.state_std"].astype(dt),
                 None if asl is None else asl.reshape(-1).astype(dt))


def hyper(g):
    gamma, lam, clip, lam_ent, lr, max_norm, reward_scale = [float(x) for x in g["hyper"]]
    return dict(gamma=gamma, lam=lam, ratio_clip=clip, lambda_entropy=lam_ent, lr=lr, max_norm=max_norm,
                reward_scale=reward_scale)


def dims(g):
    N, S, A, H, B, n_upd, vtrace, h1, h2 = [int(x) for x in g["dims"]]
    return dict(N=N, S=S, A=A, H=H, B=B, n_upd=n_upd, vtrace=bool(vtrace), h1=h1, h2=h2)


PPO_GOLDENS = ["ppo_small_vtrace.npz", "ppo_small_alt.npz", "ppo_mid_vtrace.npz", "ppo_c4shape.npz"]


# ---- toy envs + actor for the evaluator format fixture (oracle/make_golden.py:make_evaluator and tests/test_evaluator_cpu.py)
class ToyActor:
    """deterministic policy module factory: actor(state) = tanh(state @ w) (fixed weights)."""

    @staticmethod
    def build(state_dim: int, action_dim: int):
        import torch as th
        net = th.nn.Linear(state_dim, action_dim, bias=False)
        with th.no_grad():
            net.weight.copy_(th.linspace(-0.5, 0.5, state_dim * action_dim).reshape(action_dim, state_dim))
        return th.nn.Sequential(net, th.nn.Tanh())


class ToySingleEnv:
    """numpy single env: reward = 1 - |a|, episode ends after `period` steps; the period grows by one per episode."""
    num_envs, state_dim, action_dim, max_step, if_discrete, env_name = 1, 3, 2, 12, False, "ToySingle"

    def __init__(self):
        self.period, self.t = 4, 0

    def reset(self):
        self.t = 0
        return np.full(self.state_dim, 0.1 * self.period, dtype=np.float32), {}

    def step(self, action):
        self.t += 1
        state = np.full(self.state_dim, 0.1 * self.period + 0.01 * self.t, dtype=np.float32)
        done = self.t >= self.period
        if done:
            self.period += 1
        return state, float(1.0 - np.abs(action).mean()), False, bool(done), {}


class ToyVecEnv:
    """CPU-tensor vectorised env with auto-reset: env i ends an episode every 3 + i % 4 steps."""
    state_dim, action_dim, max_step, if_discrete, env_name = 3, 2, 10, False, "ToyVec"

    def __init__(self, num_envs: int = 6):
        import torch as th
        self.num_envs, self.device = num_envs, th.device("cpu")
        self.period = 3 + th.arange(num_envs) % 4
        self.t = th.zeros(num_envs, dtype=th.long)

    def reset(self):
        import torch as th
        self.t.zero_()
        return 0.1 * th.arange(self.num_envs, dtype=th.float32)[:, None].repeat(1, self.state_dim), {}

    def step(self, action):
        import torch as th
        self.t += 1
        done = self.t >= self.period
        reward = 1.0 - action.abs().mean(dim=1) + 0.01 * self.t
        self.t = th.where(done, th.zeros_like(self.t), self.t)
        state = (0.1 * th.arange(self.num_envs, dtype=th.float32) + 0.02 * self.t)[:, None].repeat(1, self.state_dim)
        return state, reward, th.zeros_like(done), done, {}


EVAL_SCHEDULE = [   # (steps, exp_r, logging_tuple) fed to Evaluator.evaluate_and_save, in order
    (100, -1.5, (0.5, 0.25, 0.125, "")),
    (100, -1.0, (0.4, 0.20, 0.100, "")),
    (300, -0.5, (0.3, 0.15, 0.075, "")),
    (50, -0.4, (0.2, 0.10, 0.050, "")),       # below eval_per_step: skipped
    (250, -0.3, (0.1, 0.05, 0.025, "")),
]
