"""GPU-resident vectorised environments implementing the reference's env protocol
(reset() -> (state, info); step(action) -> (state, reward, terminal, truncate, info); auto-reset;
attributes env_name, num_envs, max_step, state_dim, action_dim, if_discrete -- elegantrl/train/config.py:134-135,
elegantrl/envs/PointChasingEnv.py:84-182 as the in-tree exemplar).

They additionally expose `step_into(action, reward_row, terminal_row, truncate_row) -> state`, which lets the
rollout write step outputs straight into row t of its time-major buffers (no copies, no host sync).
The dynamics run in erl_synenv_step_f32 / erl_pendulum_step_f32 (elegantrl_amd/csrc/envs.hip).
"""
from __future__ import annotations

import math
from typing import Tuple

import torch as th

from .. import _hip

TEN = th.Tensor


def _epilogue_args(epilogue):
    """the nine epilogue arguments of erl_rollout_*_f32 from (last_state_out, advantages, reward_sums, stats, workspace, gamma,
    lambda_gae, use_v_trace); all-NULL when there is none"""
    if epilogue is None:
        return None, None, None, None, None, 0, 0.0, 0.0, 0
    last, adv, ret, stats, ws, gamma, lam, vtrace = epilogue
    p = _hip.ptr
    return (None if last is None else p(last, th.float32), None if adv is None else p(adv, th.float32),
            None if ret is None else p(ret, th.float32), None if stats is None else p(stats, th.float64),
            None if ws is None else p(ws, th.float64), 0 if ws is None else ws.numel() * 8, float(gamma), float(lam), int(bool(vtrace)))


class _GpuVecEnv:
    env_name = "GpuVecEnv"
    if_discrete = False

    def __init__(self, num_envs: int, state_dim: int, action_dim: int, max_step: int, gpu_id: int, seed: int):
        if not th.cuda.is_available() or gpu_id < 0:
            raise RuntimeError(f"{type(self).__name__} is GPU resident (HIP kernels); no device available for gpu_id={gpu_id}")
        self.device = th.device(f"cuda:{gpu_id}")
        self.num_envs, self.state_dim, self.action_dim, self.max_step = num_envs, state_dim, action_dim, max_step
        self.seed = int(seed)
        dev = self.device
        self.step_count = th.zeros(num_envs, dtype=th.int32, device=dev)
        self.episode = th.zeros(num_envs, dtype=th.int32, device=dev)
        self._reward = th.zeros(num_envs, dtype=th.float32, device=dev)
        self._terminal = th.zeros(num_envs, dtype=th.bool, device=dev)
        self._truncate = th.zeros(num_envs, dtype=th.bool, device=dev)
        # bumped whenever the live state changes (reset, any step): lets an agent that rolled this env out know whether its own
        # copy of the last state still IS the env's live state (AgentPPO._explore_vec_env skips the copy-back then)
        self.state_epoch = 0

    @_hip.on_device
    def step(self, action: TEN) -> Tuple[TEN, TEN, TEN, TEN, dict]:
        state = self.step_into(action.contiguous(), self._reward, self._terminal, self._truncate)
        return state.clone(), self._reward.clone(), self._terminal.clone(), self._truncate.clone(), {}

    def close(self):
        pass


class SynVecEnv(_GpuVecEnv):
    """Synthetic continuous-control workload of SURVEY.md 8d: s' = s Ws + a Wa with Ws = 0.9 I + 0.05 G1,
    Wa = 0.1 G2 (G ~ N(0,1), torch.Generator().manual_seed(0)); reward = -mean(s'^2) - 0.01 mean(a^2);
    terminal = max|s'| > 10; truncate at max_step; done rows reset to N(0,1)."""
    env_name = "SynVecEnv"

    def __init__(self, num_envs: int = 4096, state_dim: int = 64, action_dim: int = 8, max_step: int = 1000,
                 gpu_id: int = 0, seed: int = 0, **_):
        super().__init__(num_envs, state_dim, action_dim, max_step, gpu_id, seed)
        g = th.Generator().manual_seed(0)
        self.Ws = (0.9 * th.eye(state_dim) + 0.05 * th.randn(state_dim, state_dim, generator=g)).to(self.device).contiguous()
        self.Wa = (0.1 * th.randn(action_dim, state_dim, generator=g)).to(self.device).contiguous()
        self.state = th.zeros((num_envs, state_dim), dtype=th.float32, device=self.device)

    def reset(self) -> Tuple[TEN, dict]:
        self.state_epoch += 1
        g = th.Generator(device=self.device).manual_seed(self.seed)
        self.state = th.randn((self.num_envs, self.state_dim), device=self.device, generator=g)
        self.step_count.zero_()
        self.episode.zero_()
        return self.state.clone(), {}

    def step_into(self, action: TEN, reward_row: TEN, terminal_row: TEN, truncate_row: TEN) -> TEN:
        from .. import ops
        self.state_epoch += 1
        ops.synenv_step(self.state, action, self.Ws, self.Wa, self.step_count, self.episode, reward_row, terminal_row,
                        truncate_row, self.max_step, self.seed)
        return self.state

    def raw_stepper(self):
        """`step(action_ptr, reward_ptr, terminal_ptr, truncate_ptr, stream)` on raw device pointers (the rollout's
        per-step fast path: no tensor views, no argument checks); the state lives in `self.state` (fixed address)."""
        from .. import _hip
        fn = _hip.lib().erl_synenv_step_f32
        st, ws, wa, sc, ep = (t.data_ptr() for t in (self.state, self.Ws, self.Wa, self.step_count, self.episode))
        n, s, a, ms, seed = self.num_envs, self.state_dim, self.action_dim, self.max_step, self.seed & (2 ** 64 - 1)

        def step(action_ptr, reward_ptr, terminal_ptr, truncate_ptr, stream):
            self.state_epoch += 1
            rc = fn(st, action_ptr, ws, wa, sc, ep, reward_ptr, terminal_ptr, truncate_ptr, n, s, a, ms, seed, stream)
            if rc:
                _hip.check(rc, "erl_synenv_step_f32")
        return step


    def fused_rollout(self, agent, horizon_len: int, noise, bufs, values, next_value, epilogue=None) -> None:
        """all `horizon_len` steps of AgentPPO._explore_vec_env in ONE launch (erl_rollout_synenv_f32); `bufs` = (states,
        actions, logprobs, rewards, undones, unmasks) time-major, written in place; the env's state / counters advance.
        `epilogue` = (last_state_out, advantages, reward_sums, stats, workspace, gamma, lambda_gae, use_v_trace) or None: the
        kernel's optional tail (include/erl_hip.h): the agent's own copy of the final state, get_advantages + its statistics."""
        from .. import _hip
        self.state_epoch += 1
        p, f32 = _hip.ptr, th.float32
        a, c = agent._act, agent.cri
        states, actions, logprobs, rewards, undones, unmasks = bufs
        h1, h2 = agent.net_dims
        _hip.check(_hip.lib().erl_rollout_synenv_f32(
            p(agent._flat_a.flat, f32), p(agent._flat_c.flat, f32), p(a.state_avg.data, f32), p(a.state_std.data, f32),
            p(c.state_avg.data, f32), p(c.state_std.data, f32), self.state_dim, h1, h2, self.action_dim, p(self.state, f32),
            p(self.Ws, f32), p(self.Wa, f32), p(self.step_count, th.int32), p(self.episode, th.int32), self.max_step,
            self.seed & (2 ** 64 - 1), self.num_envs, horizon_len, p(noise, f32), agent.rng_seed & (2 ** 64 - 1),
            agent.rng_counter & (2 ** 64 - 1), float(agent.reward_scale), p(states, f32), p(actions, f32), p(logprobs, f32),
            p(rewards, f32), _hip.flag_ptr(undones), _hip.flag_ptr(unmasks), p(values, f32), p(next_value, f32),
            *_epilogue_args(epilogue), _hip.stream_ptr()), "erl_rollout_synenv_f32")


    def fused_rollout_offpolicy(self, agent, horizon_len: int, noise, bufs, last_state_out) -> None:
        """all `horizon_len` steps of the off-policy AgentBase._explore_vec_env in ONE launch (erl_sac_rollout_synenv_f32): `bufs` = (states,
        actions, rewards, undones, unmasks) time-major, written in place (rewards scaled, flags inverted); `last_state_out` (N, S) or None:
        the agent's own copy of the final state; the env's state / counters advance."""
        from .. import _hip
        self.state_epoch += 1
        p, f32 = _hip.ptr, th.float32
        states, actions, rewards, undones, unmasks = bufs
        spec = agent._spec
        _hip.check(_hip.lib().erl_sac_rollout_synenv_f32(
            p(agent._actor_flat, f32), spec.S, spec.A, spec._c, len(spec.hidden), p(self.state, f32), p(self.Ws, f32), p(self.Wa, f32),
            p(self.step_count, th.int32), p(self.episode, th.int32), self.max_step, self.seed & (2 ** 64 - 1), self.num_envs, horizon_len,
            p(noise, f32), agent.rng_seed & (2 ** 64 - 1), agent.rng_counter & (2 ** 64 - 1), float(agent.reward_scale), p(states, f32),
            p(actions, f32), p(rewards, f32), _hip.flag_ptr(undones), _hip.flag_ptr(unmasks),
            p(last_state_out, f32) if last_state_out is not None else None, _hip.stream_ptr()), "erl_sac_rollout_synenv_f32")


class PendulumVecEnv(_GpuVecEnv):
    """Pendulum-v1 (g=10, m=l=1, dt=0.05, 200-step truncation) behind the reference wrapper's scaling
    (elegantrl/envs/CustomGymEnv.py:42-44: torque = 2*action, reward = 0.5*gym reward)."""
    env_name = "Pendulum-v1"

    def __init__(self, num_envs: int = 4096, max_step: int = 200, gpu_id: int = 0, seed: int = 0, **_):
        super().__init__(num_envs, 3, 1, max_step, gpu_id, seed)
        self.phys = th.zeros((num_envs, 2), dtype=th.float32, device=self.device)
        self.state = th.zeros((num_envs, 3), dtype=th.float32, device=self.device)

    def reset(self) -> Tuple[TEN, dict]:
        self.state_epoch += 1
        g = th.Generator(device=self.device).manual_seed(self.seed)
        u = th.rand((self.num_envs, 2), device=self.device, generator=g) * 2 - 1
        self.phys = th.stack((u[:, 0] * math.pi, u[:, 1]), dim=1).contiguous()
        self.state = th.stack((self.phys[:, 0].cos(), self.phys[:, 0].sin(), self.phys[:, 1]), dim=1).contiguous()
        self.step_count.zero_()
        self.episode.zero_()
        return self.state.clone(), {}

    def step_into(self, action: TEN, reward_row: TEN, terminal_row: TEN, truncate_row: TEN) -> TEN:
        from .. import ops
        self.state_epoch += 1
        ops.pendulum_step(self.phys, self.state, action, self.step_count, self.episode, reward_row, terminal_row,
                          truncate_row, self.max_step, self.seed)
        return self.state

    def raw_stepper(self):
        from .. import _hip
        fn = _hip.lib().erl_pendulum_step_f32
        ph, ob, sc, ep = (t.data_ptr() for t in (self.phys, self.state, self.step_count, self.episode))
        n, ms, seed = self.num_envs, self.max_step, self.seed & (2 ** 64 - 1)

        def step(action_ptr, reward_ptr, terminal_ptr, truncate_ptr, stream):
            self.state_epoch += 1
            rc = fn(ph, ob, action_ptr, sc, ep, reward_ptr, terminal_ptr, truncate_ptr, n, ms, seed, stream)
            if rc:
                _hip.check(rc, "erl_pendulum_step_f32")
        return step


    def fused_rollout_offpolicy(self, agent, horizon_len: int, noise, bufs, last_state_out) -> None:
        """all `horizon_len` steps of the off-policy AgentBase._explore_vec_env in ONE launch (erl_sac_rollout_pendulum_f32); arguments as
        SynVecEnv.fused_rollout_offpolicy."""
        from .. import _hip
        self.state_epoch += 1
        p, f32 = _hip.ptr, th.float32
        states, actions, rewards, undones, unmasks = bufs
        spec = agent._spec
        _hip.check(_hip.lib().erl_sac_rollout_pendulum_f32(
            p(agent._actor_flat, f32), spec._c, len(spec.hidden), p(self.phys, f32), p(self.state, f32), p(self.step_count, th.int32),
            p(self.episode, th.int32), self.max_step, self.seed & (2 ** 64 - 1), self.num_envs, horizon_len, p(noise, f32),
            agent.rng_seed & (2 ** 64 - 1), agent.rng_counter & (2 ** 64 - 1), float(agent.reward_scale), p(states, f32), p(actions, f32),
            p(rewards, f32), _hip.flag_ptr(undones), _hip.flag_ptr(unmasks), p(last_state_out, f32) if last_state_out is not None else None,
            _hip.stream_ptr()), "erl_sac_rollout_pendulum_f32")

    def fused_rollout(self, agent, horizon_len: int, noise, bufs, values, next_value, epilogue=None) -> None:
        """all `horizon_len` steps of AgentPPO._explore_vec_env in ONE launch (erl_rollout_pendulum_f32); `epilogue` as
        SynVecEnv.fused_rollout."""
        from .. import _hip
        self.state_epoch += 1
        p, f32 = _hip.ptr, th.float32
        a, c = agent._act, agent.cri
        states, actions, logprobs, rewards, undones, unmasks = bufs
        h1, h2 = agent.net_dims
        _hip.check(_hip.lib().erl_rollout_pendulum_f32(
            p(agent._flat_a.flat, f32), p(agent._flat_c.flat, f32), p(a.state_avg.data, f32), p(a.state_std.data, f32),
            p(c.state_avg.data, f32), p(c.state_std.data, f32), h1, h2, p(self.phys, f32), p(self.state, f32),
            p(self.step_count, th.int32), p(self.episode, th.int32), self.max_step, self.seed & (2 ** 64 - 1), self.num_envs,
            horizon_len, p(noise, f32), agent.rng_seed & (2 ** 64 - 1), agent.rng_counter & (2 ** 64 - 1),
            float(agent.reward_scale), p(states, f32), p(actions, f32), p(logprobs, f32), p(rewards, f32),
            _hip.flag_ptr(undones), _hip.flag_ptr(unmasks), p(values, f32), p(next_value, f32), *_epilogue_args(epilogue),
            _hip.stream_ptr()), "erl_rollout_pendulum_f32")


class CartPoleVecEnv:
    """CartPole-v1 for N envs, written with plain torch ops on device tensors: an exemplar of "any env honouring the
    reference's vector protocol" for the discrete agents (action (N,) int64, auto-reset, 5-tuple).  gymnasium's physics:
    gravity 9.8, cart 1.0 kg, pole 0.1 kg, half-length 0.5 m, force 10 N, dt 0.02 s (Euler), terminal when |x| > 2.4 or
    |theta| > 12 degrees, reward 1 per step, truncation at max_step (500)."""
    env_name = "CartPole-v1"
    if_discrete = True

    def __init__(self, num_envs: int = 1024, max_step: int = 500, gpu_id: int = 0, seed: int = 0, **_):
        self.device = th.device(f"cuda:{gpu_id}" if (th.cuda.is_available() and gpu_id >= 0) else "cpu")
        self.num_envs, self.state_dim, self.action_dim, self.max_step = num_envs, 4, 2, max_step
        self.gen = th.Generator(device=self.device).manual_seed(int(seed))
        self.state = th.zeros((num_envs, 4), dtype=th.float32, device=self.device)
        self.step_count = th.zeros(num_envs, dtype=th.int32, device=self.device)

    def _fresh(self, n: int) -> TEN:
        return th.rand((n, 4), device=self.device, generator=self.gen) * 0.1 - 0.05

    def reset(self) -> Tuple[TEN, dict]:
        self.state = self._fresh(self.num_envs)
        self.step_count.zero_()
        return self.state.clone(), {}

    def step(self, action: TEN) -> Tuple[TEN, TEN, TEN, TEN, dict]:
        x, x_dot, theta, theta_dot = self.state.unbind(dim=1)
        force = th.where(action.to(self.device).reshape(-1) == 1, 10.0, -10.0).to(th.float32)
        cos, sin = theta.cos(), theta.sin()
        temp = (force + 0.05 * theta_dot * theta_dot * sin) / 1.1                       # polemass_length = 0.05, total mass 1.1
        theta_acc = (9.8 * sin - cos * temp) / (0.5 * (4.0 / 3.0 - 0.1 * cos * cos / 1.1))
        x_acc = temp - 0.05 * theta_acc * cos / 1.1
        state = th.stack((x + 0.02 * x_dot, x_dot + 0.02 * x_acc, theta + 0.02 * theta_dot, theta_dot + 0.02 * theta_acc), dim=1)
        self.step_count += 1
        terminal = (state[:, 0].abs() > 2.4) | (state[:, 2].abs() > 12 * 2 * math.pi / 360)
        truncate = (self.step_count >= self.max_step) & ~terminal
        done = terminal | truncate
        reward = th.ones(self.num_envs, dtype=th.float32, device=self.device)
        state = th.where(done[:, None], self._fresh(self.num_envs), state)
        self.step_count = th.where(done, th.zeros_like(self.step_count), self.step_count)
        self.state = state
        return state.clone(), reward, terminal, truncate, {}
