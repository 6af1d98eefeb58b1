"""`AgentSAC` for the off-policy half of the hot path (BASELINE config 3): the rollout loop feeding `ReplayBuffer.update`
(reference `AgentBase._explore_vec_env`, elegantrl/agents/AgentBase.py:130-170) and the update loop around
`ReplayBuffer.sample` (`AgentBase.update_net` :172-189 + `AgentSAC.update_objectives`, elegantrl/agents/AgentSAC.py:42-86).

What runs where: the ring write and the sample path (index split + six gathers incl. next-state) are the HIP kernels K8 /
K9 behind `ReplayBuffer`; the SAC *update math* (tanh-Gaussian actor, critic ensemble, temperature, soft update, three
Adam steps) is SURVEY.md section 8f row f1 ("next") and is still expressed with torch modules/autograd on the device -- it
is pinned to the reference by tests/golden/sac_*.npz so that the HIP version that replaces it has its oracle ready.

Networks and objectives restate the reference, quirks included: `log_prob` is evaluated at the MEAN of the Gaussian
(AgentSAC.py:197), the tanh correction uses `log(1 - tanh(a)^2 + 1e-6)` (:198), the actor is trained against the TARGET
critic ensemble's mean (:83), `alpha_log` is clamped to [-16, 2] after its own step (:80-81).
"""
from __future__ import annotations

import math
from copy import deepcopy
from typing import List, Optional, Tuple

import torch as th
from torch import nn

from ..train.config import Config
from .AgentBase import AgentBase, build_mlp, layer_init_with_orthogonal

TEN = th.Tensor


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


class AgentSAC(AgentBase):
    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int, gpu_id: int = 0, args: Config = None):
        args = Config() if args is None else args
        super().__init__(net_dims, state_dim, action_dim, gpu_id, args)
        self.if_off_policy = True
        if self.if_discrete:
            raise NotImplementedError("SAC is a continuous-action agent")
        self.num_ensembles = getattr(args, "num_ensembles", 4)
        self._act = ActorSAC(net_dims, state_dim, action_dim).to(self.device)
        self.cri = CriticEnsemble(net_dims, state_dim, action_dim, num_ensembles=self.num_ensembles).to(self.device)
        self.cri_target = deepcopy(self.cri)
        self.act_optimizer = th.optim.Adam(self._act.parameters(), self.learning_rate)
        self.cri_optimizer = th.optim.Adam(self.cri.parameters(), self.learning_rate)
        self.alpha_log = th.tensor((-1,), dtype=th.float32, requires_grad=True, device=self.device)
        self.alpha_optim = th.optim.Adam((self.alpha_log,), lr=self.learning_rate)
        self.target_entropy = math.log(action_dim)               # np.log(action_dim), as the reference (:31)

    def explore_action(self, state: TEN, noise: Optional[TEN] = None) -> TEN:
        return self._act.get_action(state, noise)

    def update_objectives(self, buffer, update_t: int, ids: Optional[TEN] = None,
                          noises: Optional[Tuple[TEN, TEN]] = None) -> Tuple[float, float]:
        """one SAC step.  `ids` / `noises` = (eps for next_state, eps for state) inject the random draws (tests)."""
        assert isinstance(update_t, int)
        if self.if_use_per:
            raise NotImplementedError("prioritised replay is SURVEY.md 8f row f2")
        n_next, n_cur = (None, None) if noises is None else noises
        with th.no_grad():
            state, action, reward, undone, unmask, next_state = buffer.sample(self.batch_size, ids=ids)   # HIP K9
            next_action, next_logprob = self._act.get_action_logprob(next_state, n_next)
            next_q = th.min(self.cri_target.get_q_values(next_state, next_action), dim=1)[0]
            alpha = self.alpha_log.exp()
            q_label = reward + undone * self.gamma * (next_q - next_logprob * alpha)

        q_values = self.cri.get_q_values(state, action)
        q_labels = q_label.view((-1, 1)).repeat(1, q_values.shape[1])
        td_error = self.criterion(q_values, q_labels).mean(dim=1) * unmask
        obj_critic = td_error.mean()
        if self.lambda_fit_cum_r:
            cum_r = buffer.cum_rewards[buffer.ids0, buffer.ids1].detach().mean().repeat(q_values.shape[1])
            obj_critic = obj_critic + self.criterion(cum_r, q_values.mean(dim=0)).mean() * self.lambda_fit_cum_r
        self.optimizer_backward(self.cri_optimizer, obj_critic)
        self.soft_update(self.cri_target, self.cri, self.soft_update_tau)

        action_pg, logprob = self._act.get_action_logprob(state, n_cur)
        obj_alpha = (self.alpha_log * (self.target_entropy - logprob).detach()).mean()
        self.optimizer_backward(self.alpha_optim, obj_alpha)

        alpha = self.alpha_log.exp().detach()
        with th.no_grad():
            self.alpha_log[:] = self.alpha_log.clamp(-16, 2)
        q_value_pg = self.cri_target(state, action_pg).mean()
        obj_actor = (q_value_pg - logprob * alpha).mean()
        self.optimizer_backward(self.act_optimizer, -obj_actor)
        return obj_critic.item(), obj_actor.item()
