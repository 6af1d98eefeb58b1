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


PPO_GOLDENS = ["ppo_small_vtrace.npz", "ppo_small_alt.npz", "ppo_mid_vtrace.npz"]
