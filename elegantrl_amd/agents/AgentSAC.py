"""`AgentSAC` on the HIP path (BASELINE config 3): off-policy rollout feeding `ReplayBuffer.update` (reference
`AgentBase._explore_vec_env`, elegantrl/agents/AgentBase.py:130-170), and the update loop around `ReplayBuffer.sample`
(`AgentBase.update_net` :172-189 + `AgentSAC.update_objectives`, elegantrl/agents/AgentSAC.py:42-86).

Everything numerical runs in liberl_hip.so: ring write / sample are K8 / K9, the exploration action is
`erl_sac_explore_action_f32` and the whole update step after the sample -- tanh-Gaussian actor, critic ensemble, target
min, temperature, soft update, three clip + Adam steps -- is ONE `erl_sac_update_f32` call (csrc/sac.hip).  The
`nn.Module`s below only describe the parameter layout (their tensors are views into the flat blocks the kernels
read/write) and serve the Evaluator's `actor(state)` / checkpoints.  No torch fallback for the update math: the torch
restatement lives in oracle/sac_torch.py as test infrastructure.

Quirks of the reference are reproduced (see csrc/sac.hip): log-prob evaluated at the Gaussian MEAN (:197), tanh correction
`log(1 - tanh^2 + 1e-6)` (:198), actor trained against the TARGET ensemble mean (:83), alpha clamped after its own step.
"""
from __future__ import annotations

import math
import os
from copy import deepcopy
from typing import List, Optional, Tuple

import numpy as np
import torch as th
from torch import nn

from .. import _hip
from ..train.config import Config
from .AgentBase import AgentBase, FlatNet, build_mlp, layer_init_with_orthogonal

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


class ActorFixSAC(nn.Module):
    """AgentModSAC's actor (elegantrl/agents/AgentSAC.py:201-243): encoder build_mlp([S, *net_dims]) whose LAST layer is raw, two
    one-layer decoders for the mean and the log-std (clamped to [-20, 2]), the log-prob AT the sample with the tanh correction in its
    softplus form.  The training path does not call these methods (kernels on the flat block: csrc/sac.hip, ERL_SAC_ACTOR_FIX)."""

    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int):
        super().__init__()
        self.state_dim, self.action_dim = state_dim, action_dim
        self.encoder_s = build_mlp(dims=[state_dim, *net_dims])
        self.decoder_a_avg = build_mlp(dims=[net_dims[-1], action_dim])
        self.decoder_a_std = build_mlp(dims=[net_dims[-1], action_dim])
        self.soft_plus = nn.Softplus()
        layer_init_with_orthogonal(self.decoder_a_avg[-1], std=0.1)
        layer_init_with_orthogonal(self.decoder_a_std[-1], std=0.1)

    def forward(self, state: TEN) -> TEN:
        return self.decoder_a_avg(self.encoder_s(state)).tanh()

    def get_action(self, state: TEN, noise: Optional[TEN] = None, **_kwargs) -> TEN:
        tmp = self.encoder_s(state)
        avg, std = self.decoder_a_avg(tmp), self.decoder_a_std(tmp).clamp(-20, 2).exp()
        eps = th.randn_like(avg) if noise is None else noise
        return (avg + std * eps).tanh()

    def get_action_logprob(self, state: TEN, noise: Optional[TEN] = None) -> Tuple[TEN, TEN]:
        tmp = self.encoder_s(state)
        log_std = self.decoder_a_std(tmp).clamp(-20, 2)
        avg = self.decoder_a_avg(tmp)
        eps = th.randn_like(avg) if noise is None else noise
        action = avg + log_std.exp() * eps
        logprob = -log_std - eps.pow(2) * 0.5 - math.log(math.sqrt(2 * math.pi))
        logprob = logprob - (math.log(2.) - action - self.soft_plus(-2. * action)) * 2.
        return action.tanh(), logprob.sum(1)


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



class _Slices:
    def __init__(self, slices):
        self._s = slices

    def slices(self):
        return self._s


class _FlatAdamState:
    """Adam moments of one parameter block as flat tensors (what `save_or_load_agent` stores for the optimisers)."""

    def __init__(self, numel: int, lr: float, device):
        self.exp_avg = th.zeros(numel, dtype=th.float32, device=device)
        self.exp_avg_sq = th.zeros(numel, dtype=th.float32, device=device)
        self.step_count = 0
        self.param_groups = [{"lr": lr, "betas": (0.9, 0.999), "eps": 1e-8}]


class AgentSAC(AgentBase):
    _actor_class = ActorSAC
    _actor_variant = _hip.SAC_ACTOR_SAC
    _default_ensembles = 4

    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int, gpu_id: int = 0, args: Config = None):
        args = Config() if args is None else args
        super().__init__(net_dims, state_dim, action_dim, gpu_id, args)
        self.if_off_policy = True
        if self.if_discrete:
            raise NotImplementedError("SAC is a continuous-action agent")
        if self.device.type != "cuda":
            raise _hip.HipExtensionError("AgentSAC runs on the HIP kernels only; no GPU is visible and there is no CPU fallback")
        from .. import ops
        self.num_ensembles = getattr(args, "num_ensembles", self._default_ensembles)
        # update_net draws the sample ids of all its steps with one th.randint (False: one draw per step, the reference's call pattern)
        self.sample_ids_ahead = bool(getattr(args, "sample_ids_ahead", True))
        # device-resident envs that offer it: the whole off-policy rollout as one launch (False: the per-step loop)
        self.fused_rollout = bool(getattr(args, "fused_rollout", True))
        # update_net hands the replay ring + the drawn ids to the step instead of sampling first (False: sample, then step)
        self.sample_in_step = bool(getattr(args, "sample_in_step", os.environ.get("ERL_SAC_SAMPLE_IN_STEP", "1") != "0"))
        # round 6: with the sample inside the step, update_net's whole loop is ONE C call (erl_sac_update_ring_loop_f32): bit-identical to the
        # per-step calls; `args.update_loop_in_c = False` / ERL_SAC_LOOP_IN_C=0 keeps one call per step
        self.update_loop_in_c = bool(getattr(args, "update_loop_in_c", os.environ.get("ERL_SAC_LOOP_IN_C", "1") != "0"))
        self._last_state_token = None
        self._spec = ops.SacSpec(state_dim, action_dim, net_dims, self.num_ensembles, actor_variant=self._actor_variant)
        dev, f32 = self.device, th.float32
        self._actor_flat = th.zeros(self._spec.actor_count, dtype=f32, device=dev)
        self._critic_flat = th.zeros(self._spec.critic_count, dtype=f32, device=dev)
        self._target_flat = th.zeros(self._spec.critic_count, dtype=f32, device=dev)
        act = self._actor_class(net_dims, state_dim, action_dim).to(dev)
        cri = CriticEnsemble(net_dims, state_dim, action_dim, num_ensembles=self.num_ensembles).to(dev)
        cri_target = deepcopy(cri)
        self._bind_a = FlatNet(act, _Slices(self._spec.actor_slices()), self._actor_flat)
        self._bind_c = FlatNet(cri, _Slices(self._spec.critic_slices()), self._critic_flat)
        self._bind_t = FlatNet(cri_target, _Slices(self._spec.critic_slices()), self._target_flat)
        self._act, self.cri, self.cri_target = act, cri, cri_target
        self.act_optimizer = _FlatAdamState(self._spec.actor_count, self.learning_rate, dev)
        self.cri_optimizer = _FlatAdamState(self._spec.critic_count, self.learning_rate, dev)
        self.alpha_optim = _FlatAdamState(1, self.learning_rate, dev)
        self.alpha_log = th.full((1,), -1.0, dtype=f32, device=dev)
        self.target_entropy = math.log(action_dim)               # np.log(action_dim), as the reference (:31)
        self._step = 0
        self._objs = th.zeros(2, dtype=f32, device=dev)
        self._td_error = None                      # per-sample td errors of a prioritised step (csrc/sac.hip critic_loss_kernel)
        self.save_attr_names = self.save_attr_names | {"alpha_log", "alpha_optim"}
        import ctypes as _ct
        hid = (_ct.c_int * len(net_dims))(*[int(h) for h in net_dims])
        fused_ok = (not self._actor_variant and len(net_dims) == 2 and os.environ.get("ERL_SAC_FUSED", "1") != "0"
                    and bool(_hip.lib().erl_sac_rollout_synenv_supported(state_dim, action_dim, hid, len(net_dims), 16)))
        self.kernel_path = ("fused SAC step (two hidden layers <= 256 wide, S + A <= 64, A <= 8, batch <= 4096: tile kernels, sample inside the step's "
                            "first launch, one-launch rollout on device-resident envs)" if fused_ok else
                            "layered SAC step (one MFMA GEMM launch per dense layer): " +
                            ("ActorFixSAC / two-time-scale options (AgentModSAC)" if self._actor_variant else
                             f"net_dims {list(net_dims)}, state_dim {state_dim}, action_dim {action_dim} outside the fused step's shapes"))
        if not getattr(args, "quiet", False) and os.environ.get("ERL_QUIET", "0") == "0":
            print(f"| {type(self).__name__}: {self.kernel_path}", flush=True)

    def _on_act_replaced(self):
        if getattr(self, "_bind_a", None) is not None and self._act is not None and not self._bind_a.is_bound(self._act):
            self._act = self._act.to(self.device)
            self._bind_a.bind(self._act)

    def _sync_modules(self):
        for bind, mod in ((self._bind_a, self._act), (self._bind_c, self.cri), (self._bind_t, self.cri_target)):
            if not bind.is_bound(mod):
                bind.bind(mod)

    def save_or_load_agent(self, cwd: str, if_save: bool):
        """AgentBase.save_or_load_agent plus what a resumed SAC run needs beyond the reference's file set: the temperature
        and its optimiser state (`alpha_log`, `alpha_optim`), the Adam step (bias correction) and the Philox counters."""
        if if_save:
            for opt in (self.act_optimizer, self.cri_optimizer, self.alpha_optim):
                opt.step_count = self._step
                opt.rng_counter = self.rng_counter
        super().save_or_load_agent(cwd, if_save)
        if not if_save:
            self.alpha_log = self.alpha_log.to(self.device, th.float32).contiguous()
            for opt in (self.act_optimizer, self.cri_optimizer, self.alpha_optim):
                opt.exp_avg = opt.exp_avg.to(self.device, th.float32).contiguous()
                opt.exp_avg_sq = opt.exp_avg_sq.to(self.device, th.float32).contiguous()
            self._step = int(self.act_optimizer.step_count)
            self.rng_counter = int(getattr(self.act_optimizer, "rng_counter", self.rng_counter))
            if self.cri is not None:
                self.cri = self.cri.to(self.device)
            if self.cri_target is not None:
                self.cri_target = self.cri_target.to(self.device)
            self._on_act_replaced()
            self._sync_modules()

    @_hip.on_device
    def _explore_vec_env(self, env, horizon_len: int, noise: Optional[TEN] = None) -> Tuple[TEN, ...]:
        """AgentBase._explore_vec_env (AgentBase.py:130-170); on a device-resident env that offers `fused_rollout_offpolicy` (SynVecEnv) the
        whole loop is ONE launch (erl_sac_rollout_synenv_f32: same five tensors, final state and env counters, bit for bit, as the
        per-step loop; `args.fused_rollout = False` keeps the loop)."""
        H, N, S, A, dev = horizon_len, self.num_envs, self.state_dim, self.action_dim, self.device
        if not (self.fused_rollout and not self._actor_variant and hasattr(env, "fused_rollout_offpolicy") and getattr(env, "num_envs", None) == N
                and getattr(env, "device", None) == dev and getattr(env, "state_dim", None) == S
                and _hip.lib().erl_sac_rollout_synenv_supported(self._spec.S, self._spec.A, self._spec._c, len(self._spec.hidden), N)):
            self._last_state_token = None
            return super()._explore_vec_env(env, horizon_len, noise)
        self._sync_modules()
        state = self.last_state.to(dev, th.float32)
        assert state.shape == (N, S)
        if state.data_ptr() != env.state.data_ptr():
            tok = self._last_state_token
            # `state` is the copy of the final state the last fused rollout of THIS env wrote, untouched, and the env has not moved since:
            # the env's live buffer already holds it; otherwise the env takes the agent's state (it owns the live buffer)
            same = (tok is not None and tok[0] is self.last_state and tok[1] == self.last_state._version and tok[2] is env
                    and tok[3] == getattr(env, "state_epoch", None) and tok[4] == (id(env.state), env.state._version))
            if not same:
                env.state.copy_(state)
                env.state_epoch += 1
        states = th.empty((H, N, S), dtype=th.float32, device=dev)
        actions = th.empty((H, N, A), dtype=th.float32, device=dev)
        rewards = th.empty((H, N), dtype=th.float32, device=dev)
        undones = th.empty((H, N), dtype=th.bool, device=dev)
        unmasks = th.empty((H, N), dtype=th.bool, device=dev)
        last_out = th.empty((N, S), dtype=th.float32, device=dev)
        env.fused_rollout_offpolicy(self, H, None if noise is None else noise.contiguous(), (states, actions, rewards, undones, unmasks), last_out)
        self.rng_counter += H
        self.last_state = last_out
        self._last_state_token = (last_out, last_out._version, env, getattr(env, "state_epoch", None), (id(env.state), env.state._version))
        return states, actions, rewards, undones, unmasks

    @_hip.on_device
    def explore_action(self, state: TEN, noise: Optional[TEN] = None, out: Optional[TEN] = None, out_state: Optional[TEN] = None) -> TEN:
        """`out` (n, action_dim), contiguous: the kernel writes the action there (the rollout passes its buffer row: no copy);
        `out_state` (n, state_dim), contiguous: the kernel also copies `state` there (the rollout's `states[t] = state`)"""
        from .. import ops
        self._sync_modules()
        action = ops.sac_explore_action(self._spec, self._actor_flat, state.contiguous(), noise=noise, seed=self.rng_seed,
                                        counter=self.rng_counter, out=out, out_state=out_state)
        self.rng_counter += 1
        return action

    def _update_from_ring(self, buffer, ring, ids: TEN, objs_out: TEN, noises=None):
        """ReplayBuffer.sample(ids) and the step from one C call; the buffer's stage / ids0 / ids1 end up as after `buffer.sample`"""
        from .. import ops
        self._step += 1
        arrays, sample_len, stage = ring
        ops.sac_update_from_ring(self._spec, self._actor_flat, self._critic_flat, self._target_flat, self.alpha_log,
                                 (self.act_optimizer.exp_avg, self.act_optimizer.exp_avg_sq, self.cri_optimizer.exp_avg,
                                  self.cri_optimizer.exp_avg_sq, self.alpha_optim.exp_avg, self.alpha_optim.exp_avg_sq),
                                 arrays, ids, sample_len, stage, self._step, gamma=float(self.gamma), target_entropy=float(self.target_entropy),
                                 tau=float(self.soft_update_tau), lr=float(self.learning_rate), max_norm=float(self.clip_grad_norm),
                                 objs_out=objs_out, noises=noises, seed=self.rng_seed + 1, counter=self._step)
        buffer.ids0, buffer.ids1 = stage.ids
        self.act_optimizer.step_count = self.cri_optimizer.step_count = self.alpha_optim.step_count = self._step

    def _step_options(self, update_t: int) -> dict:
        """extra keyword arguments of ops.sac_update for this step (AgentModSAC: its two-time-scale rule and actor target)"""
        return {}

    def _update_on_batch(self, batch, objs_out: TEN, noises=None, is_weight=None, td_error_out=None, buffer=None, update_t: int = 0):
        from .. import ops
        self._step += 1
        cum_reward = None
        if self.lambda_fit_cum_r:          # AgentSAC.py:66-68: the sampled transitions' n-step returns (the sampler left ids0 / ids1)
            cum_reward = buffer.cum_rewards[buffer.ids0, buffer.ids1].to(th.float32).contiguous()
        ops.sac_update(self._spec, self._actor_flat, self._critic_flat, self._target_flat, self.alpha_log,
                       (self.act_optimizer.exp_avg, self.act_optimizer.exp_avg_sq, self.cri_optimizer.exp_avg,
                        self.cri_optimizer.exp_avg_sq, self.alpha_optim.exp_avg, self.alpha_optim.exp_avg_sq),
                       batch, self._step, gamma=float(self.gamma), target_entropy=float(self.target_entropy),
                       tau=float(self.soft_update_tau), lr=float(self.learning_rate), max_norm=float(self.clip_grad_norm),
                       objs_out=objs_out, noises=noises, seed=self.rng_seed + 1, counter=self._step, is_weight=is_weight,
                       td_error_out=td_error_out, cum_reward=cum_reward, lambda_fit_cum_r=float(self.lambda_fit_cum_r or 0.0),
                       **self._step_options(update_t))
        self.act_optimizer.step_count = self.cri_optimizer.step_count = self.alpha_optim.step_count = self._step

    @_hip.on_device
    def update_objectives(self, buffer, update_t: int, ids: Optional[TEN] = None,
                          noises: Optional[Tuple[TEN, TEN]] = None) -> Tuple[float, float]:
        """one SAC step (AgentSAC.py:42-86).  `ids` / `noises` = (eps for next_state, eps for state) inject the random draws."""
        assert isinstance(update_t, int)
        self._sync_modules()
        if self.if_use_per:                                                               # AgentSAC.py:45-47, :60-62
            self._per_step(buffer, self._objs, noises, update_t=update_t)
        else:
            batch = buffer.sample(self.batch_size, ids=ids)                               # HIP K9
            self._update_on_batch(batch, self._objs, noises, buffer=buffer, update_t=update_t)
        oc, oa = self._objs.cpu().tolist()
        _hip.check_async_faults()          # the stream is drained: a skipped optimiser step (grid-wait timeout) raises here
        return oc, oa

    def _per_step(self, buffer, objs_out: TEN, noises=None, update_t: int = 0):
        """one step on a prioritised sample: importance weights into the critic objective, td errors back into the trees"""
        *batch, is_weight, is_index = buffer.sample_for_per(self.batch_size)
        if self._td_error is None or self._td_error.numel() != is_weight.numel():
            self._td_error = th.empty_like(is_weight)
        self._update_on_batch(batch, objs_out, noises, is_weight=is_weight, td_error_out=self._td_error, buffer=buffer, update_t=update_t)
        buffer.td_error_update_for_per(is_index, self._td_error)

    @_hip.on_device
    def update_net(self, buffer) -> Tuple[float, float]:
        """AgentBase.update_net (:172-189) with ONE host sync: the per-step objectives stay on the device until the end."""
        if self.lambda_fit_cum_r:                     # AgentBase.py:176-177 (bootstraps with the current actor: see get_cumulative_rewards)
            buffer.update_cum_rewards(get_cumulative_rewards=self.get_cumulative_rewards)
        self._sync_modules()
        update_times = int(buffer.cur_size * self.repeat_times / self.batch_size)
        if update_times < 1:
            return 0.0, 0.0
        objs = th.zeros((update_times, 2), dtype=th.float32, device=self.device)
        id_rows = None
        if not self.if_use_per and self.sample_ids_ahead:
            # the sample ids of ALL the steps in one th.randint (the reference draws batch_size of them per step, replay_buffer.py:121-122:
            # same distribution, same generator, one launch instead of `update_times`; nothing is written to the buffer inside this loop)
            id_all = th.randint((buffer.cur_size - 1) * buffer.num_seqs, size=(update_times, self.batch_size), requires_grad=False,
                                device=self.device)
            id_rows = id_all.unbind(0)
        # the sample rides in the step's first launch (erl_sac_update_ring_f32) where the buffer is the library's continuous-action ring and
        # nothing needs ids0 / ids1 before the step
        ring = (id_rows is not None and self.sample_in_step and not self.lambda_fit_cum_r and not self._actor_variant
                and getattr(buffer, "ring_for_fused_sample", None) and buffer.ring_for_fused_sample(self.batch_size))
        if ring and self.update_loop_in_c:
            # the whole loop from ONE C call (erl_sac_update_ring_loop_f32, round 6): step t = what _update_from_ring(id_rows[t]) enqueues
            from .. import ops
            arrays, sample_len, stage = ring
            ops.sac_update_ring_loop(self._spec, self._actor_flat, self._critic_flat, self._target_flat, self.alpha_log,
                                     (self.act_optimizer.exp_avg, self.act_optimizer.exp_avg_sq, self.cri_optimizer.exp_avg,
                                      self.cri_optimizer.exp_avg_sq, self.alpha_optim.exp_avg, self.alpha_optim.exp_avg_sq),
                                     arrays, id_all, sample_len, stage, self._step + 1, gamma=float(self.gamma),
                                     target_entropy=float(self.target_entropy), tau=float(self.soft_update_tau), lr=float(self.learning_rate),
                                     max_norm=float(self.clip_grad_norm), objs_all=objs, seed=self.rng_seed + 1, counter0=self._step + 1)
            self._step += update_times
            buffer.ids0, buffer.ids1 = stage.ids
            self.act_optimizer.step_count = self.cri_optimizer.step_count = self.alpha_optim.step_count = self._step
            update_times = 0                              # (nothing left for the Python loop)
        for t in range(update_times):
            if self.if_use_per:
                self._per_step(buffer, objs[t], update_t=t)
            elif ring:
                self._update_from_ring(buffer, ring, id_rows[t], objs[t])
            else:       # (the batch is consumed before the next draw)
                self._update_on_batch(buffer.sample(self.batch_size, ids=None if id_rows is None else id_rows[t], reuse=True), objs[t], buffer=buffer,
                                      update_t=t)
        o = objs.cpu().numpy()
        _hip.check_async_faults()          # the stream is drained: a skipped optimiser step (grid-wait timeout) raises here
        oa = o[:, 1][~np.isnan(o[:, 1])]   # (AgentModSAC: steps whose actor update was skipped log nan, AgentBase.py:186-188)
        return float(np.nanmean(o[:, 0])), float(oa.mean()) if oa.size else 0.0


class AgentModSAC(AgentSAC):
    """Modified SAC (elegantrl/agents/AgentSAC.py:89-165): ActorFixSAC, 8 critics, `target_entropy = -log(action_dim)`, an actor target
    that follows the actor by soft updates, and the "auto two-time-scale update rule": the actor is updated on a step only while
    `update_a / (update_t + 1) < 1 / (2 - reliable_lambda)`, `reliable_lambda = exp(-critic_value ** 2)` (the reference never moves
    `critic_value` off 1.0, so this is 0.61: the actor skips roughly every third step).  The step itself is erl_sac_update_opt_f32
    (csrc/sac.hip, layered path; ErlSacOptions carries what differs from AgentSAC)."""
    _actor_class = ActorFixSAC
    _actor_variant = _hip.SAC_ACTOR_FIX
    _default_ensembles = 8

    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int, gpu_id: int = 0, args: Config = None):
        args = Config() if args is None else args
        super().__init__(net_dims, state_dim, action_dim, gpu_id, args)
        self.target_entropy = getattr(args, "target_entropy", -math.log(action_dim))      # (:107; AgentSAC: +log(action_dim))
        self.critic_tau = getattr(args, "critic_tau", 0.995)
        self.critic_value = 1.0                                  # for reliable_lambda (:110; never updated by the reference either)
        self.update_a = 0                                        # actor updates of the current update_net loop (:111)
        self._actor_step = 0                                     # the actor optimiser's own Adam step (it steps only when updated)
        self._actor_target_flat = self._actor_flat.clone()
        self.act_target = deepcopy(self._act)
        self._bind_at = FlatNet(self.act_target, _Slices(self._spec.actor_slices()), self._actor_target_flat)
        self._last_actor_updated = True

    def _sync_modules(self):
        super()._sync_modules()
        if not self._bind_at.is_bound(self.act_target):
            self._bind_at.bind(self.act_target)

    def _step_options(self, update_t: int) -> dict:
        reliable_lambda = math.exp(-self.critic_value ** 2)
        self.update_a = 0 if update_t == 0 else self.update_a                           # (:150)
        do = (self.update_a / (update_t + 1)) < (1 / (2 - reliable_lambda))              # (:151)
        if do:
            self.update_a += 1
            self._actor_step += 1
        self._last_actor_updated = do
        return dict(update_actor=do, actor_step=max(1, self._actor_step), actor_target=self._actor_target_flat)

    def _update_on_batch(self, *a, **k):
        super()._update_on_batch(*a, **k)
        self.act_optimizer.step_count = self._actor_step         # (th.optim.Adam's own count: the actor steps only when it is updated)

    def save_or_load_agent(self, cwd: str, if_save: bool):
        if if_save:
            self.act_optimizer.actor_step = self._actor_step
        super().save_or_load_agent(cwd, if_save)
        if not if_save:
            self._actor_step = int(getattr(self.act_optimizer, "actor_step", self._step))
            if getattr(self, "act_target", None) is not None:
                self.act_target = self.act_target.to(self.device)
            self._sync_modules()
