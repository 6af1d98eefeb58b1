"""CPU (torch) port of the reference's PPO actor-learner loop.  ORACLE / BASELINE ONLY (see oracle/__init__.py).

Purpose: (1) the `cpu_baseline` leg of bench.py -- "ElegantRL's own CPU path" is a chain of ATen ops driven
from Python, and this port issues the same op sequence (per-step actor forward + Normal sample/log_prob,
Python-loop GAE, per-minibatch advanced-index gathers, autograd backward, clip_grad_norm_, torch.optim.Adam),
so its throughput on the host cores is a faithful stand-in for the reference when /root/reference is not
present (it never is on the GPU box); (2) a second, autograd-based checker of the manual-backward numpy oracle.

Restates: elegantrl/agents/AgentPPO.py:87-129 (rollout), :135-171 (update_net), :173-205 (objectives),
:207-232 (GAE), elegantrl/agents/AgentBase.py:239-248 (optimizer_backward), :345-365 (MLP + init).
Pinned by tests/test_oracle_golden.py::test_torch_port_* against tests/golden/ppo_*.npz.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch as th
from torch import nn

TEN = th.Tensor


def make_mlp(dims: List[int]) -> nn.Sequential:
    mods: list = []
    for i in range(len(dims) - 1):
        mods.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            mods.append(nn.GELU())
    return nn.Sequential(*mods)


class PortNet(nn.Module):
    def __init__(self, dims: List[int], last_std: float, with_std_log: bool):
        super().__init__()
        self.net = make_mlp(dims)
        nn.init.orthogonal_(self.net[-1].weight, last_std)
        nn.init.constant_(self.net[-1].bias, 1e-6)
        self.state_avg = nn.Parameter(th.zeros(dims[0]), requires_grad=False)
        self.state_std = nn.Parameter(th.ones(dims[0]), requires_grad=False)
        if with_std_log:
            self.action_std_log = nn.Parameter(th.zeros((1, dims[-1])))

    def body(self, state: TEN) -> TEN:
        return self.net((state - self.state_avg) / (self.state_std + 1e-4))


class TorchPortPPO:
    def __init__(self, state_dim: int, action_dim: int, net_dims=(128, 128), *, lr=6e-5, gamma=0.99, lam=0.95,
                 ratio_clip=0.25, lambda_entropy=0.001, max_norm=3.0, reward_scale=1.0, use_v_trace=True):
        self.actor = PortNet([state_dim, *net_dims, action_dim], 0.1, True)
        self.critic = PortNet([state_dim, *net_dims, 1], 0.5, False)
        self.opt_a = th.optim.Adam(self.actor.parameters(), lr)
        self.opt_c = th.optim.Adam(self.critic.parameters(), lr)
        self.gamma, self.lam, self.clip, self.lam_ent = gamma, lam, ratio_clip, lambda_entropy
        self.max_norm, self.reward_scale, self.use_v_trace = max_norm, reward_scale, use_v_trace
        self.last_state: Optional[TEN] = None

    # ---- rollout -------------------------------------------------------------------------------------
    @th.no_grad()
    def explore(self, env, horizon: int):
        n, s_dim = self.last_state.shape
        a_dim = self.actor.action_std_log.shape[1]
        states = th.zeros((horizon, n, s_dim))
        actions = th.zeros((horizon, n, a_dim))
        logprobs = th.zeros((horizon, n))
        rewards = th.zeros((horizon, n))
        terminals = th.zeros((horizon, n), dtype=th.bool)
        truncates = th.zeros((horizon, n), dtype=th.bool)
        state = self.last_state
        for t in range(horizon):
            dist = th.distributions.Normal(self.actor.body(state), self.actor.action_std_log.exp())
            action = dist.sample()
            states[t], actions[t], logprobs[t] = state, action, dist.log_prob(action).sum(1)
            state, rewards[t], terminals[t], truncates[t] = env.step(action.tanh())
        self.last_state = state
        rewards *= self.reward_scale
        return states, actions, logprobs, rewards, ~terminals, ~truncates

    # ---- GAE -----------------------------------------------------------------------------------------
    @th.no_grad()
    def advantages(self, states, rewards, undones, unmasks, values):
        adv = th.empty_like(values)
        trunc = ~unmasks
        if th.any(trunc):
            rewards[trunc] += self.critic.body(states[trunc]).squeeze(1)
            undones[trunc] = False
        masks = undones * self.gamma
        nv = self.critic.body(self.last_state).squeeze(-1)
        a = th.zeros_like(nv)
        horizon = rewards.shape[0]
        if self.use_v_trace:
            for t in range(horizon - 1, -1, -1):
                nv = rewards[t] + masks[t] * nv
                adv[t] = a = nv - values[t] + masks[t] * self.lam * a
                nv = values[t]
        else:
            for t in range(horizon - 1, -1, -1):
                adv[t] = rewards[t] - values[t] + masks[t] * a
                a = values[t] + self.lam * adv[t]
        return adv

    # ---- update --------------------------------------------------------------------------------------
    def _step(self, opt, loss):
        opt.zero_grad()
        loss.backward()
        nn.utils.clip_grad_norm_(opt.param_groups[0]["params"], self.max_norm)
        opt.step()

    def update(self, buffer, batch_size: int, update_times: int, ids: Optional[TEN] = None) -> Tuple[float, float, float]:
        states, actions, logprobs, rewards, undones, unmasks = buffer
        horizon, n = rewards.shape
        with th.no_grad():
            rows = max(1, 1024 // n)
            values = th.cat([self.critic.body(states[i:i + rows]) for i in range(0, horizon, rows)], 0).squeeze(-1)
            adv = self.advantages(states, rewards, undones, unmasks, values)
            rsum = adv + values
            adv = (adv - adv.mean()) / (adv[::4, ::4].std() + 1e-5)
        logs = []
        with th.enable_grad():
            for k in range(update_times):
                idx = th.randint(horizon * n, (batch_size,)) if ids is None else ids[k]
                i0, i1 = th.fmod(idx, horizon), th.div(idx, horizon, rounding_mode="floor")
                s, a, um = states[i0, i1], actions[i0, i1], unmasks[i0, i1]
                lp_old, ad, rs = logprobs[i0, i1], adv[i0, i1], rsum[i0, i1]
                obj_c = ((self.critic.body(s).squeeze(1) - rs) ** 2 * um).mean()
                self._step(self.opt_c, obj_c)
                dist = th.distributions.Normal(self.actor.body(s), self.actor.action_std_log.exp())
                ratio = (dist.log_prob(a).sum(1) - lp_old).exp()
                surrogate = ad * ratio * th.where(ad.gt(0), 1 - self.clip, 1 + self.clip)
                obj_s = (surrogate * um).mean()
                obj_e = (dist.entropy().sum(1) * um).mean()
                self._step(self.opt_a, -(obj_s - obj_e * self.lam_ent))
                logs.append((obj_c.item(), obj_s.item(), obj_e.item()))
        m = th.tensor(logs, dtype=th.float64).mean(0)
        return float(m[0]), float(m[1]), float(m[2])


class TorchSynEnv:
    """CPU twin of elegantrl_amd.envs.SynVecEnv (same maps and rules; resets drawn from a torch generator)."""

    def __init__(self, num_envs=4096, state_dim=64, action_dim=8, max_step=1000, seed=0):
        g = th.Generator().manual_seed(0)
        self.Ws = 0.9 * th.eye(state_dim) + 0.05 * th.randn(state_dim, state_dim, generator=g)
        self.Wa = 0.1 * th.randn(action_dim, state_dim, generator=g)
        self.g = th.Generator().manual_seed(seed)
        self.n, self.s, self.max_step = num_envs, state_dim, max_step
        self.state = th.randn((num_envs, state_dim), generator=self.g)
        self.count = th.zeros(num_envs, dtype=th.int32)

    def reset(self):
        return self.state.clone()

    def step(self, action: TEN):
        s = self.state @ self.Ws + action @ self.Wa
        reward = -(s * s).mean(1) - 0.01 * (action * action).mean(1)
        self.count += 1
        terminal = s.abs().amax(1) > 10.0
        truncate = (self.count >= self.max_step) & ~terminal
        done = terminal | truncate
        if bool(done.any()):
            s = th.where(done[:, None], th.randn((self.n, self.s), generator=self.g), s)
            self.count[done] = 0
        self.state = s
        return s, reward, terminal, truncate


class TorchPendulumEnv:
    """CPU twin of elegantrl_amd.envs.PendulumVecEnv (csrc/envs.hip pendulum_step_kernel): Pendulum-v1 (g = 10, m = l = 1,
    dt = 0.05, |u| <= 2, |theta_dot| <= 8) behind the reference wrapper's scaling (elegantrl/envs/CustomGymEnv.py:42-44: torque =
    2 * action, reward = 0.5 * gym reward), truncation after max_step steps, reset theta ~ U(-pi, pi), theta_dot ~ U(-1, 1)."""

    def __init__(self, num_envs=4096, max_step=200, seed=0):
        import math
        self.pi = math.pi
        self.g = th.Generator().manual_seed(seed)
        self.n, self.max_step = num_envs, max_step
        self.theta = (th.rand(num_envs, generator=self.g) * 2 - 1) * self.pi
        self.theta_dot = th.rand(num_envs, generator=self.g) * 2 - 1
        self.count = th.zeros(num_envs, dtype=th.int32)

    def _obs(self):
        return th.stack((self.theta.cos(), self.theta.sin(), self.theta_dot), dim=1)

    def reset(self):
        return self._obs()

    def step(self, action: TEN):
        u = (2.0 * action.reshape(-1)).clamp(-2.0, 2.0)
        ang = th.remainder(self.theta + self.pi, 2 * self.pi) - self.pi
        cost = ang * ang + 0.1 * self.theta_dot * self.theta_dot + 0.001 * u * u
        nthdot = (self.theta_dot + (15.0 * self.theta.sin() + 3.0 * u) * 0.05).clamp(-8.0, 8.0)
        nth = self.theta + nthdot * 0.05
        self.count += 1
        truncate = self.count >= self.max_step
        if bool(truncate.any()):
            nth = th.where(truncate, (th.rand(self.n, generator=self.g) * 2 - 1) * self.pi, nth)
            nthdot = th.where(truncate, th.rand(self.n, generator=self.g) * 2 - 1, nthdot)
            self.count[truncate] = 0
        self.theta, self.theta_dot = nth, nthdot
        return self._obs(), -0.5 * cost, th.zeros(self.n, dtype=th.bool), truncate
