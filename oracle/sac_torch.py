"""torch (CPU) restatement of the reference's SAC update step.  ORACLE / TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates elegantrl/agents/AgentSAC.py:42-86 (update_objectives), :167-199 (ActorSAC), :243-259 (CriticEnsemble),
elegantrl/agents/AgentBase.py:239-248 (optimizer_backward), :270-278 (soft_update) with every random draw injectable.
Pinned by tests/test_sac.py::test_torch_restatement_replays_the_reference against tests/golden/sac_small.npz (outputs of
the reference's own AgentSAC, oracle/make_golden.py:make_sac); the HIP implementation (csrc/sac.hip) is then checked
against the same fixture.  `ActorFixSAC` / `ModSacStepper` restate AgentModSAC (:89-165, :201-243) the same way, pinned by
tests/golden/sac_mod_small.npz (oracle/make_golden.py:make_sac_mod).
"""
from __future__ import annotations

import math
from copy import deepcopy
from typing import List, Optional, Tuple

import torch as th
from torch import nn

TEN = th.Tensor


def build_mlp(dims: List[int], if_raw_out: bool = True) -> nn.Sequential:
    layers: list = []
    for d_in, d_out in zip(dims[:-1], dims[1:]):
        layers += [nn.Linear(d_in, d_out), nn.GELU()]
    if if_raw_out:
        layers.pop()
    return nn.Sequential(*layers)


def layer_init_with_orthogonal(layer, std: float = 1.0, bias_const: float = 1e-6):
    th.nn.init.orthogonal_(layer.weight, std)
    th.nn.init.constant_(layer.bias, bias_const)


class ActorSAC(nn.Module):
    """state -> encoder MLP (GELU after every layer) -> linear head -> (mean, log_std); action = tanh(mean + std * eps)."""

    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int):
        super().__init__()
        self.state_dim, self.action_dim = state_dim, action_dim
        self.net_s = build_mlp(dims=[state_dim, *net_dims], if_raw_out=False)
        self.net_a = build_mlp(dims=[net_dims[-1], action_dim * 2])
        layer_init_with_orthogonal(self.net_a[-1], std=0.1)

    def _head(self, state: TEN) -> Tuple[TEN, TEN]:
        mean, log_std = self.net_a(self.net_s(state)).chunk(2, dim=1)
        return mean, log_std.clamp(-16, 2).exp()

    def forward(self, state: TEN) -> TEN:
        return self.net_a(self.net_s(state))[:, :self.action_dim].tanh()

    def get_action(self, state: TEN, noise: Optional[TEN] = None) -> TEN:
        mean, std = self._head(state)
        eps = th.randn_like(mean) if noise is None else noise
        return (mean + std * eps).tanh()                       # Normal(mean, std).rsample().tanh()

    def get_action_logprob(self, state: TEN, noise: Optional[TEN] = None) -> Tuple[TEN, TEN]:
        mean, std = self._head(state)
        eps = th.randn_like(mean) if noise is None else noise
        action_tanh = (mean + std * eps).tanh()
        logprob = -std.log() - math.log(math.sqrt(2 * math.pi))          # Normal.log_prob evaluated at the mean (:197)
        logprob = logprob - (-action_tanh.pow(2) + 1.000001).log()       # tanh correction (:198)
        return action_tanh, logprob.sum(1)


class CriticEnsemble(nn.Module):
    """shared (state, action) encoder layer + `num_ensembles` independent Q decoders; forward = ensemble mean."""

    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int, num_ensembles: int = 4):
        super().__init__()
        self.state_dim, self.action_dim = state_dim, action_dim
        self.encoder_sa = build_mlp(dims=[state_dim + action_dim, net_dims[0]])
        self.decoder_qs = []
        for i in range(num_ensembles):
            dec = build_mlp(dims=[*net_dims, 1])
            layer_init_with_orthogonal(dec[-1], std=0.5)
            self.decoder_qs.append(dec)
            setattr(self, f"decoder_q{i:02}", dec)              # registers the parameters under the reference's names

    def get_q_values(self, state: TEN, action: TEN) -> TEN:
        enc = self.encoder_sa(th.cat((state, action), dim=1))
        return th.cat([dec(enc) for dec in self.decoder_qs], dim=-1)

    def forward(self, state: TEN, action: TEN) -> TEN:
        return self.get_q_values(state, action).mean(dim=-1, keepdim=True)



class SacStepper:
    """holds actor / critic / target / alpha and their Adam optimisers; `step` = one update_objectives on a given batch."""

    def __init__(self, net_dims, state_dim, action_dim, num_ensembles, lr, gamma, tau, max_norm):
        self.act = ActorSAC(list(net_dims), state_dim, action_dim)
        self.cri = CriticEnsemble(list(net_dims), state_dim, action_dim, num_ensembles)
        self.cri_target = deepcopy(self.cri)
        self.alpha_log = th.tensor((-1,), dtype=th.float32, requires_grad=True)
        self.gamma, self.tau, self.max_norm = gamma, tau, max_norm
        self.target_entropy = math.log(action_dim)
        self.lr = lr
        self.reset_optimizers()

    def reset_optimizers(self):
        self.act_opt = th.optim.Adam(self.act.parameters(), self.lr)
        self.cri_opt = th.optim.Adam(self.cri.parameters(), self.lr)
        self.alpha_opt = th.optim.Adam((self.alpha_log,), lr=self.lr)

    def _opt(self, opt, obj):
        opt.zero_grad()
        obj.backward()
        th.nn.utils.clip_grad_norm_(parameters=opt.param_groups[0]["params"], max_norm=self.max_norm)
        opt.step()

    def step(self, batch, eps_next: TEN, eps_cur: TEN, is_weight: Optional[TEN] = None, cum_reward: Optional[TEN] = None,
             lambda_fit_cum_r: float = 0.0) -> Tuple[float, float]:
        """`is_weight`: prioritised replay's importance weights (AgentSAC.py:60-62); the per-sample td errors of the step are
        left in `self.td_error`.  `cum_reward` (B,) + `lambda_fit_cum_r`: the fit-the-mean-return term (AgentSAC.py:66-68)."""
        state, action, reward, undone, unmask, next_state = batch
        with th.no_grad():
            next_action, next_logprob = self.act.get_action_logprob(next_state, eps_next)
            next_q = th.min(self.cri_target.get_q_values(next_state, next_action), dim=1)[0]
            q_label = reward + undone * self.gamma * (next_q - next_logprob * self.alpha_log.exp())
        q_values = self.cri.get_q_values(state, action)
        td = ((q_values - q_label.view(-1, 1)) ** 2).mean(dim=1) * unmask
        self.td_error = td.detach().clone()
        obj_critic = td.mean() if is_weight is None else (td * is_weight).mean()
        if lambda_fit_cum_r:
            cum_reward_mean = cum_reward.mean().repeat(q_values.shape[1])
            obj_critic = obj_critic + ((cum_reward_mean - q_values.mean(dim=0)) ** 2).mean() * lambda_fit_cum_r
        self._opt(self.cri_opt, obj_critic)
        with th.no_grad():
            for tar, cur in zip(self.cri_target.parameters(), self.cri.parameters()):
                tar.data.copy_(cur.data * self.tau + tar.data * (1.0 - self.tau))
        action_pg, logprob = self.act.get_action_logprob(state, eps_cur)
        self._opt(self.alpha_opt, (self.alpha_log * (self.target_entropy - logprob).detach()).mean())
        alpha = self.alpha_log.exp().detach()
        with th.no_grad():
            self.alpha_log[:] = self.alpha_log.clamp(-16, 2)
        obj_actor = (self.cri_target(state, action_pg).mean() - logprob * alpha).mean()
        self._opt(self.act_opt, -obj_actor)
        return obj_critic.item(), obj_actor.item()


class ActorFixSAC(nn.Module):
    """elegantrl/agents/AgentSAC.py:201-243: raw last encoder layer, two one-layer decoders, log_std in [-20, 2], the log-prob at the
    sample with the softplus form of the tanh correction"""

    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int):
        super().__init__()
        self.state_dim, self.action_dim = state_dim, action_dim
        self.encoder_s = build_mlp(dims=[state_dim, *net_dims])
        self.decoder_a_avg = build_mlp(dims=[net_dims[-1], action_dim])
        self.decoder_a_std = build_mlp(dims=[net_dims[-1], action_dim])
        self.soft_plus = nn.Softplus()
        layer_init_with_orthogonal(self.decoder_a_avg[-1], std=0.1)
        layer_init_with_orthogonal(self.decoder_a_std[-1], std=0.1)

    def get_action(self, state: TEN, noise: TEN) -> TEN:
        tmp = self.encoder_s(state)
        return (self.decoder_a_avg(tmp) + self.decoder_a_std(tmp).clamp(-20, 2).exp() * noise).tanh()

    def get_action_logprob(self, state: TEN, noise: TEN) -> Tuple[TEN, TEN]:
        tmp = self.encoder_s(state)
        log_std = self.decoder_a_std(tmp).clamp(-20, 2)
        action = self.decoder_a_avg(tmp) + log_std.exp() * noise
        logprob = -log_std - noise.pow(2) * 0.5 - math.log(math.sqrt(2 * math.pi))
        logprob = logprob - (math.log(2.) - action - self.soft_plus(-2. * action)) * 2.
        return action.tanh(), logprob.sum(1)


class ModSacStepper(SacStepper):
    """AgentModSAC.update_objectives (elegantrl/agents/AgentSAC.py:113-159): SacStepper's step with ActorFixSAC, target_entropy =
    -log(action_dim), the two-time-scale rule on the actor and an actor target that follows by soft updates"""

    def __init__(self, net_dims, state_dim, action_dim, num_ensembles, lr, gamma, tau, max_norm):
        super().__init__(net_dims, state_dim, action_dim, num_ensembles, lr, gamma, tau, max_norm)
        self.act = ActorFixSAC(list(net_dims), state_dim, action_dim)
        self.act_target = deepcopy(self.act)
        self.target_entropy = -math.log(action_dim)
        self.critic_value, self.update_a = 1.0, 0
        self.reset_optimizers()

    def step(self, batch, eps_next: TEN, eps_cur: TEN, update_t: int = 0) -> Tuple[float, float]:
        state, action, reward, undone, unmask, next_state = batch
        with th.no_grad():
            next_action, next_logprob = self.act.get_action_logprob(next_state, eps_next)
            next_q = th.min(self.cri_target.get_q_values(next_state, next_action), dim=1)[0]
            q_label = reward + undone * self.gamma * (next_q - next_logprob * self.alpha_log.exp())
        q_values = self.cri.get_q_values(state, action)
        obj_critic = (((q_values - q_label.view(-1, 1)) ** 2).mean(dim=1) * unmask).mean()
        self._opt(self.cri_opt, obj_critic)
        with th.no_grad():
            for tar, cur in zip(self.cri_target.parameters(), self.cri.parameters()):
                tar.data.copy_(cur.data * self.tau + tar.data * (1.0 - self.tau))
        action_pg, logprob = self.act.get_action_logprob(state, eps_cur)
        self._opt(self.alpha_opt, (self.alpha_log * (self.target_entropy - logprob).detach()).mean())
        alpha = self.alpha_log.exp().detach()
        with th.no_grad():
            self.alpha_log[:] = self.alpha_log.clamp(-16, 2)
        reliable_lambda = math.exp(-self.critic_value ** 2)
        self.update_a = 0 if update_t == 0 else self.update_a
        if (self.update_a / (update_t + 1)) < (1 / (2 - reliable_lambda)):
            self.update_a += 1
            obj_actor = (self.cri_target(state, action_pg).mean() - logprob * alpha).mean()
            self._opt(self.act_opt, -obj_actor)
            with th.no_grad():
                for tar, cur in zip(self.act_target.parameters(), self.act.parameters()):
                    tar.data.copy_(cur.data * self.tau + tar.data * (1.0 - self.tau))
            return obj_critic.item(), obj_actor.item()
        return obj_critic.item(), float("nan")
