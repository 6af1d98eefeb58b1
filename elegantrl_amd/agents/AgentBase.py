"""`AgentBase`: constructor contract, attributes and checkpoint format of elegantrl/agents/AgentBase.py,
plus the flat-parameter plumbing the HIP kernels need.

What is kept from the reference (it is the API run.py / Evaluator drive):
ctor `(net_dims, state_dim, action_dim, gpu_id, args)`, the hyper-parameter attributes read from
`Config`, `explore_env` dispatch (:70-74), settable `act` / `last_state`, `save_or_load_agent`
(:280-297, whole-object `th.save` files), `explore_rate` (run.py:126 needs it) and the torch-module
builders `build_mlp` / `layer_init_with_orthogonal` (:345-365).

What is new: `FlatNet` makes an actor/critic's trainable parameters views into one flat fp32 buffer laid
out as include/erl_hip.h describes, so kernels read/write weights in place while the objects stay
ordinary picklable `nn.Module`s.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch as th
from torch import nn

from .. import _hip
from ..train.config import Config

TEN = th.Tensor


def build_mlp(dims: List[int], activation=None, if_raw_out: bool = True) -> nn.Sequential:
    """Linear -> GELU -> ... -> Linear (no activation after the last layer when if_raw_out)."""
    act = nn.GELU if activation is None else activation
    layers: list = []
    for d_in, d_out in zip(dims[:-1], dims[1:]):
        layers += [nn.Linear(d_in, d_out), act()]
    if if_raw_out:
        layers.pop()
    return nn.Sequential(*layers)


def layer_init_with_orthogonal(layer, std: float = 1.0, bias_const: float = 1e-6):
    th.nn.init.orthogonal_(layer.weight, std)
    th.nn.init.constant_(layer.bias, bias_const)


class FlatNet:
    """Binds an `nn.Module` with sub-module `net` = build_mlp([S, h1, h2, out]) (and optionally
    `action_std_log`) to a slice of a flat fp32 parameter buffer.  After `bind`, each trainable
    parameter's `.data` is a view into the slice, in the order W1,b1,W2,b2,W3,b3[,action_std_log]."""

    def __init__(self, module: nn.Module, spec, flat_slice: TEN):
        self.module = module
        self.spec = spec
        self.flat = flat_slice
        self.bind(module)

    def _named(self, module):
        sd = dict(module.named_parameters())
        return [(name, sd[name], off, shape) for name, off, shape in self.spec.slices()]

    def bind(self, module: nn.Module) -> None:
        """copy the module's current values into the flat slice and re-point the parameters at it."""
        self.module = module
        self._probe = None
        with th.no_grad():
            for name, p, off, shape in self._named(module):
                assert tuple(p.shape) == tuple(shape), f"{name}: {tuple(p.shape)} != {shape}"
                view = self.flat[off:off + p.numel()].view(shape)
                view.copy_(p.data.to(self.flat.device, th.float32))
                p.data = view

    def is_bound(self, module: nn.Module) -> bool:
        """does the module's first parameter still live in the flat block?  (O(1): the owning sub-module and attribute are looked
        up once per bind -- walking named_parameters() on every explore_env / update_net cost ~20 us of interpreter time each,
        with the GPU idle behind the previous iteration's host sync)"""
        if module is not self.module:
            return False
        probe = self._probe
        if probe is None or probe[0] is not module:
            name, _, off, _ = self._named(module)[0]
            owner, _, attr = name.rpartition(".")
            probe = self._probe = (module, module.get_submodule(owner) if owner else module, attr, off)
        p = probe[1]._parameters.get(probe[2])
        return p is not None and p.data_ptr() == self.flat.data_ptr() + 4 * probe[3]


class AgentBase:
    """Hyper-parameter capture + the generic pieces shared by the agents."""

    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int, gpu_id: int = 0, args: Config = None):
        args = Config() if args is None else args
        self.if_discrete: bool = args.if_discrete
        self.if_off_policy: bool = args.if_off_policy
        self.net_dims = net_dims
        self.state_dim = state_dim
        self.action_dim = action_dim

        self.gamma = args.gamma
        self.max_step = args.max_step
        self.num_envs = args.num_envs
        self.batch_size = args.batch_size
        self.repeat_times = args.repeat_times
        self.reward_scale = args.reward_scale
        self.learning_rate = args.learning_rate
        self.clip_grad_norm = args.clip_grad_norm
        self.soft_update_tau = args.soft_update_tau
        self.state_value_tau = args.state_value_tau
        self.buffer_init_size = args.buffer_init_size

        self.explore_noise_std = getattr(args, "explore_noise_std", 0.05)
        self.explore_rate = getattr(args, "explore_rate", 1.0)  # read by run.py:126 in the single-process loop
        self.last_state: Optional[TEN] = None
        self.device = th.device(f"cuda:{gpu_id}" if (th.cuda.is_available() and gpu_id >= 0) else "cpu")

        self._act = None
        self.cri = None
        self.act_target = None
        self.cri_target = None
        self.act_optimizer = None
        self.cri_optimizer = None

        self.criterion = getattr(args, "criterion", th.nn.MSELoss(reduction="none"))
        self.if_vec_env = self.num_envs > 1
        self.if_use_per = getattr(args, "if_use_per", None)
        self.lambda_fit_cum_r = getattr(args, "lambda_fit_cum_r", 0.0)

        self.save_attr_names = {"act", "act_target", "act_optimizer", "cri", "cri_target", "cri_optimizer"}

        # data-parallel context (one process per GPU; see elegantrl_amd/parallel.py)
        self.world_size = int(getattr(args, "world_size", 1))
        self.rank = int(getattr(args, "rank", 0))
        seed = getattr(args, "random_seed", None)
        self.rng_seed = (int(seed) if seed is not None else max(0, gpu_id)) * 1000003 + 7919 * self.rank
        self.rng_counter = 0

    # `act` is settable (run.py:404-407 replaces it with the learner's copy); subclasses re-bind kernels
    @property
    def act(self):
        return self._act

    @act.setter
    def act(self, module):
        self._act = module
        self._on_act_replaced()

    def _on_act_replaced(self):
        pass

    @_hip.on_device
    def explore_env(self, env, horizon_len: int, if_random: bool = False) -> Tuple[TEN, ...]:
        """AgentBase.py:70-74.  `if_random` is accepted for signature parity with the PPO rollout of the reference
        (AgentPPO.py:87), which never reads it either."""
        if self.if_vec_env:
            return self._explore_vec_env(env=env, horizon_len=horizon_len)
        return self._explore_one_env(env=env, horizon_len=horizon_len)

    # spellings used by older releases / the north star
    def explore_vec_env(self, env, horizon_len: int):
        return self._explore_vec_env(env=env, horizon_len=horizon_len)

    def explore_one_env(self, env, horizon_len: int):
        return self._explore_one_env(env=env, horizon_len=horizon_len)

    # ---- off-policy defaults (overridden by the on-policy agents): AgentBase.py:76-189 of the reference ----------
    def explore_action(self, state: TEN) -> TEN:
        return self.act.get_action(state)

    def _explore_vec_env(self, env, horizon_len: int, noise: Optional[TEN] = None) -> Tuple[TEN, ...]:
        """off-policy vectorised rollout -> (states, actions, rewards, undones, unmasks), time-major (H, N, .); the
        already-squashed action is both stored and sent to the env (AgentBase.py:130-170).  GPU-resident envs write
        reward / terminal / truncate straight into row t (`step_into`)."""
        H, N, dev = horizon_len, self.num_envs, self.device
        states = th.zeros((H, N, self.state_dim), dtype=th.float32, device=dev)
        actions = th.zeros((H, N, self.action_dim), dtype=th.float32, device=dev)
        rewards = th.zeros((H, N), dtype=th.float32, device=dev)
        terminals = th.zeros((H, N), dtype=th.bool, device=dev)
        truncates = th.zeros((H, N), dtype=th.bool, device=dev)
        native = hasattr(env, "step_into")
        state = self.last_state.to(dev, th.float32)
        assert state.shape == (N, self.state_dim)
        import inspect
        sig = inspect.signature(self.explore_action).parameters
        into_row = "out" in sig                     # the agent's kernel writes the action into its buffer row
        state_row = into_row and "out_state" in sig     # ... and `states[t] = state` too
        # (the rows as views made once: the loop is bound by the interpreter -- ~45 us per time step for three launches)
        st_rows, ac_rows, rw_rows, te_rows, tr_rows = states.unbind(0), actions.unbind(0), rewards.unbind(0), terminals.unbind(0), truncates.unbind(0)
        for t in range(H):
            if state_row:
                action = self.explore_action(state, None if noise is None else noise[t], out=ac_rows[t], out_state=st_rows[t])
            elif into_row:
                action = self.explore_action(state, None if noise is None else noise[t], out=ac_rows[t])
            else:
                action = self.explore_action(state) if noise is None else self.explore_action(state, noise[t])
                ac_rows[t].copy_(action)
            if not state_row:
                st_rows[t].copy_(state)
            if native:
                state = env.step_into(action if into_row else action.contiguous(), rw_rows[t], te_rows[t], tr_rows[t])
            else:
                state, reward, terminal, truncate, _ = env.step(action)
                state = state.to(dev, th.float32)
                rewards[t] = reward
                terminals[t] = terminal
                truncates[t] = truncate
        self.last_state = state.clone() if native else state
        rewards *= self.reward_scale
        return states, actions, rewards, th.logical_not(terminals), th.logical_not(truncates)

    def _explore_one_env(self, env, horizon_len: int) -> Tuple[TEN, ...]:
        """single numpy env (AgentBase.py:76-128): same 5-tuple with a middle dimension of 1."""
        H, dev = horizon_len, self.device
        states = th.zeros((H, 1, self.state_dim), dtype=th.float32, device=dev)
        actions = th.zeros((H, 1, self.action_dim), dtype=th.float32, device=dev)
        rewards = th.zeros((H, 1), dtype=th.float32, device=dev)
        terminals = th.zeros((H, 1), dtype=th.bool, device=dev)
        truncates = th.zeros((H, 1), dtype=th.bool, device=dev)
        state = self.last_state.to(dev, th.float32).reshape(1, self.state_dim)
        for t in range(H):
            action = self.explore_action(state)
            states[t], actions[t] = state, action
            ary_state, reward, terminal, truncate, _ = env.step(action[0].detach().cpu().numpy())
            if terminal or truncate:
                ary_state, _ = env.reset()
            state = th.as_tensor(ary_state, dtype=th.float32, device=dev).reshape(1, self.state_dim)
            rewards[t, 0], terminals[t, 0], truncates[t, 0] = float(reward), bool(terminal), bool(truncate)
        self.last_state = state
        rewards *= self.reward_scale
        return states, actions, rewards, th.logical_not(terminals), th.logical_not(truncates)

    @_hip.on_device
    def update_net(self, buffer) -> Tuple[float, ...]:
        """off-policy update loop (AgentBase.py:172-189): `update_times = int(cur_size * repeat_times / batch_size)` steps of
        `update_objectives`, each drawing one minibatch through ReplayBuffer.sample (HIP K9)."""
        import numpy as np
        objs_critic, objs_actor = [], []
        if self.lambda_fit_cum_r != 0:
            buffer.update_cum_rewards(get_cumulative_rewards=self.get_cumulative_rewards)
        th.set_grad_enabled(True)
        update_times = int(buffer.cur_size * self.repeat_times / self.batch_size)
        for update_t in range(update_times):
            obj_critic, obj_actor = self.update_objectives(buffer=buffer, update_t=update_t)
            objs_critic.append(obj_critic)
            if isinstance(obj_actor, float):
                objs_actor.append(obj_actor)
        th.set_grad_enabled(False)
        return (float(np.nanmean(objs_critic)) if objs_critic else 0.0, float(np.nanmean(objs_actor)) if objs_actor else 0.0)

    def update_objectives(self, buffer, update_t: int) -> Tuple[float, float]:
        raise NotImplementedError

    @_hip.on_device
    def get_cumulative_rewards(self, rewards: TEN, undones: TEN) -> TEN:
        """n-step discounted return of the newest `add_size` rows of the replay buffer (AgentBase.py:226-237), bootstrapped
        with `cri_target(last_state, act_target(last_state))`; the backward scan runs in erl_cum_rewards_f32 (bit-exact op
        order).  Agents without a target actor (the reference's AgentSAC leaves `act_target = None` and would raise
        'NoneType is not callable' here) bootstrap with the current actor instead."""
        from .. import ops
        if self.device.type != "cuda":
            raise _hip.HipExtensionError("get_cumulative_rewards runs on the HIP kernels only; no GPU is visible")
        last_state = self.last_state.to(self.device, th.float32)
        actor = self.act_target if self.act_target is not None else self.act
        critic = self.cri_target if self.cri_target is not None else self.cri
        with th.no_grad():
            next_value = critic(last_state, actor(last_state)).detach().reshape(-1).to(th.float32).contiguous()
        return ops.cum_rewards(rewards.contiguous(), undones.to(th.float32).contiguous(), next_value, float(self.gamma))

    def optimizer_backward(self, optimizer, objective: TEN):
        """zero_grad, backward, global-norm clip of the optimiser's first param group, step (AgentBase.py:239-248)."""
        optimizer.zero_grad()
        objective.backward()
        th.nn.utils.clip_grad_norm_(parameters=optimizer.param_groups[0]["params"], max_norm=self.clip_grad_norm)
        optimizer.step()

    @staticmethod
    def soft_update(target_net: nn.Module, current_net: nn.Module, tau: float):
        with th.no_grad():
            for tar, cur in zip(target_net.parameters(), current_net.parameters()):
                tar.data.mul_(1.0 - tau).add_(cur.data, alpha=tau)

    def save_or_load_agent(self, cwd: str, if_save: bool):
        """whole-object checkpoint files `{cwd}/{attr}.pth`, as AgentBase.py:280-297."""
        assert self.save_attr_names.issuperset({"act", "act_optimizer"})
        for attr_name in self.save_attr_names:
            path = f"{cwd}/{attr_name}.pth"
            obj = getattr(self, attr_name, None)
            if if_save:
                if obj is not None:
                    th.save(obj, path)
            elif os.path.isfile(path):
                setattr(self, attr_name, th.load(path, map_location=self.device, weights_only=False))
