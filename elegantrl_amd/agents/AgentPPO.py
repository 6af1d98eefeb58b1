"""`AgentPPO` on the HIP hot path: vectorised rollout (K1), value pre-pass (K2), GAE scan (K3), advantage
normalisation (K4), fused gather + PPO objective forward/backward (K5/K6) and clip + Adam (K7).

Drop-in for elegantrl/agents/AgentPPO.py (AgentPPO :12-232, ActorPPO :348-390, CriticPPO :425-441): same
constructor, same `explore_env` / `update_net` / `get_advantages` / `explore_action` signatures and
returned shapes/dtypes, same objective (including its quirks: sign-dependent "clip" scale :199,
penalised entropy :203-204, `[::4, ::4]` std :149, `update_times = int(H * repeat_times / batch_size)`
:159) -- the arithmetic itself runs in liberl_hip.so.  There is no PyTorch fallback for these methods:
without a HIP device / the extension they raise.
"""
from __future__ import annotations

import os
import weakref
from typing import List, Optional, Tuple

import torch as th
from torch import nn

from .. import _hip
from ..train.config import Config
from .AgentBase import AgentBase, FlatNet, build_mlp, layer_init_with_orthogonal

TEN = th.Tensor


class ActorPPO(nn.Module):
    """Gaussian policy head: same parameters/buffers and torch-side methods as the reference so that the
    Evaluator's `actor(state)`, `th.save(actor)` and user code keep working; the training path does not
    call these methods (it uses the fused kernels on the flat parameter buffer)."""

    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int):
        super().__init__()
        self.net = build_mlp(dims=[state_dim, *net_dims, action_dim])
        layer_init_with_orthogonal(self.net[-1], std=0.1)
        self.action_std_log = nn.Parameter(th.zeros((1, action_dim)), requires_grad=True)
        self.ActionDist = th.distributions.normal.Normal
        self.state_avg = nn.Parameter(th.zeros((state_dim,)), requires_grad=False)
        self.state_std = nn.Parameter(th.ones((state_dim,)), requires_grad=False)

    def state_norm(self, state: TEN) -> TEN:
        return (state - self.state_avg) / (self.state_std + 1e-4)

    def forward(self, state: TEN) -> TEN:
        return self.convert_action_for_env(self.net(self.state_norm(state)))

    def get_action(self, state: TEN) -> Tuple[TEN, TEN]:
        dist = self.ActionDist(self.net(self.state_norm(state)), self.action_std_log.exp())
        action = dist.sample()
        return action, dist.log_prob(action).sum(1)

    def get_logprob_entropy(self, state: TEN, action: TEN) -> Tuple[TEN, TEN]:
        dist = self.ActionDist(self.net(self.state_norm(state)), self.action_std_log.exp())
        return dist.log_prob(action).sum(1), dist.entropy().sum(1)

    @staticmethod
    def convert_action_for_env(action: TEN) -> TEN:
        return action.tanh()


class ActorDiscretePPO(ActorPPO):
    """Categorical policy head (elegantrl/agents/AgentPPO.py:393-422): `net` emits logits; `forward` is the greedy action
    the Evaluator uses.  `action_std_log` is inherited but unused, exactly as in the reference (keeps state_dicts alike)."""

    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int):
        super().__init__(net_dims=net_dims, state_dim=state_dim, action_dim=action_dim)
        self.ActionDist = th.distributions.Categorical
        self.soft_max = nn.Softmax(dim=-1)

    def forward(self, state: TEN) -> TEN:
        return self.net(self.state_norm(state)).argmax(dim=1)

    def get_action(self, state: TEN) -> Tuple[TEN, TEN]:
        dist = self.ActionDist(self.soft_max(self.net(self.state_norm(state))))
        action = dist.sample()
        return action, dist.log_prob(action)

    def get_logprob_entropy(self, state: TEN, action: TEN) -> Tuple[TEN, TEN]:
        dist = self.ActionDist(self.soft_max(self.net(self.state_norm(state))))
        return dist.log_prob(action), dist.entropy()

    @staticmethod
    def convert_action_for_env(action: TEN) -> TEN:
        return action.long()


class CriticPPO(nn.Module):
    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int):
        super().__init__()
        assert isinstance(action_dim, int)
        self.net = build_mlp(dims=[state_dim, *net_dims, 1])
        layer_init_with_orthogonal(self.net[-1], std=0.5)
        self.state_avg = nn.Parameter(th.zeros((state_dim,)), requires_grad=False)
        self.state_std = nn.Parameter(th.ones((state_dim,)), requires_grad=False)

    def state_norm(self, state: TEN) -> TEN:
        return (state - self.state_avg) / (self.state_std + 1e-4)

    def forward(self, state: TEN) -> TEN:
        return self.net(self.state_norm(state))


class FlatAdam:
    """Adam state of one network as views into the agent's flat moment buffers (picklable; takes the place
    of the reference's `th.optim.Adam` objects in `save_or_load_agent`)."""

    def __init__(self, exp_avg: TEN, exp_avg_sq: TEN, lr: float):
        self.exp_avg, self.exp_avg_sq, self.lr = exp_avg, exp_avg_sq, lr
        self.step_count = 0
        self.param_groups = [{"lr": lr, "betas": (0.9, 0.999), "eps": 1e-8}]


class PendingLogs:
    """update_net(..., lazy=True): the logged objectives of an update whose kernels are still running.  `result()` -> the three floats
    update_net returns otherwise (and the device-fault check that goes with its host sync).  Two pinned host blocks are used in turn;
    a block whose previous owner has not been read yet is resolved first (its numbers are cached on that owner), so a third
    `update_net(lazy=True)` before the first `result()` cannot overwrite what the first one will return."""

    def __init__(self, agent):
        ring = agent.__dict__.setdefault("_lazy_ring", [])
        if len(ring) < 2:                                   # two pinned blocks, used in turn (allocated once: hipHostMalloc is slow)
            ring.append([th.empty(4, dtype=th.float32, pin_memory=True), th.cuda.Event(), None])
        agent._lazy_turn = (getattr(agent, "_lazy_turn", -1) + 1) % 2
        slot = ring[min(agent._lazy_turn, len(ring) - 1)]
        owner = slot[2]() if slot[2] is not None else None
        if owner is not None and owner._vals is None:       # the block's last owner is still unread: read it before the block is reused
            owner._resolve(check=False)
        self._host, self._event = slot[0], slot[1]
        slot[2] = weakref.ref(self)
        self._host.copy_(agent._logs, non_blocking=True)
        self._event.record()
        self._vals = None

    def _resolve(self, check: bool = True):
        self._event.synchronize()
        self._vals = tuple(self._host[:3].tolist())
        if check:
            _hip.check_async_faults()

    def result(self) -> Tuple[float, float, float]:
        if self._vals is None:
            self._resolve()
        return self._vals


class AgentPPO(AgentBase):
    """PPO + GAE, reference-form objective, HIP kernels."""
    _discrete = False            # AgentDiscretePPO: categorical head on the layered path
    _actor_class = ActorPPO
    supports_lazy_logs = True          # update_net(..., lazy=True) -> PendingLogs (train_agent reads it after the next rollout is enqueued)

    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int, gpu_id: int = 0, args: Config = None):
        args = Config() if args is None else args
        super().__init__(net_dims, state_dim, action_dim, gpu_id, args)
        self.if_off_policy = False
        if bool(self.if_discrete) != self._discrete:
            raise ValueError(f"{type(self).__name__} with if_discrete={self.if_discrete}: use "
                             f"{'AgentDiscretePPO' if self.if_discrete else 'AgentPPO'}")
        # fused register-chained kernels: 2 hidden layers of width 32..128 (multiples of 32); every other build_mlp()
        # shape (the reference's demos go up to (256, 128, 128)) takes the layered generic path (erl_mlpn_*)
        self._fused = (not self._discrete and len(net_dims) == 2
                       and all(32 <= d <= _hip.MAX_HIDDEN and d % 32 == 0 for d in net_dims)
                       and state_dim <= _hip.MAX_STATE_DIM and action_dim <= _hip.MAX_ACTION_DIM)
        # net_dims = (256, 64 | 128) (examples/demo_A2C_PPO.py:117 trains (256, 128)): rollout and value pre-pass on the layered path,
        # the minibatch loop on its own fused kernel (csrc/ppo_step_wd.hip) with the fused path's slabs, optimiser tail and C loop;
        # args.wide_fused = False / ERL_WIDE_FUSED=0 keeps the layered minibatch step (A/B runs, parity tests)
        import os as _os0
        self._wide = (not self._fused and not self._discrete and len(net_dims) == 2 and net_dims[0] == 256 and net_dims[1] in (64, 128)
                      and state_dim <= 64 and action_dim <= 8
                      and bool(getattr(args, "wide_fused", _os0.environ.get("ERL_WIDE_FUSED", "1") != "0")))
        self._fused_update = self._fused or self._wide

        self.ratio_clip = getattr(args, "ratio_clip", 0.25)
        self.lambda_gae_adv = getattr(args, "lambda_gae_adv", 0.95)
        self.lambda_entropy_value = float(getattr(args, "lambda_entropy", 0.001))
        self.lambda_entropy = th.tensor(self.lambda_entropy_value, dtype=th.float32, device=self.device)
        # the reference reads `if_use_v_trace`; its Config sets `if_use_vtrace` (never consumed): accept both
        self.if_use_v_trace = getattr(args, "if_use_v_trace", getattr(args, "if_use_vtrace", True))
        self.gae_algo = getattr(args, "gae_algo", "auto")
        # arithmetic of the minibatch kernel's large products (include/erl_hip.h, erl_ppo_set_arith): "auto" = the library default
        # (split bf16 operands on the bf16 matrix pipe, fp32-equivalent, where the net shape allows), "f32" = the fp32 MFMA.
        # Passed with every call of the minibatch entry points (ERL_PPO_MODE, ABI 17), so agents of one process do not share it.
        # `agent.last_state` after a fused rollout: a tensor of the agent's own, written by the rollout kernel itself (the reference's
        # behaviour, AgentPPO.py:125; no extra launch).  args.snapshot_last_state = False aliases the env's live state buffer instead.
        self.snapshot_last_state = bool(getattr(args, "snapshot_last_state", True))
        # the persistent rollout's epilogue also runs get_advantages + its statistics (three launches less per iteration);
        # args.fused_gae = False / ERL_FUSED_GAE=0 keeps them in update_net (A/B runs, parity tests)
        import os as _os
        self.fused_gae = bool(getattr(args, "fused_gae", _os.environ.get("ERL_FUSED_GAE", "1") != "0"))
        # left to itself the epilogue serves horizons up to 128 steps (its inputs wait in LDS); beyond, a 16-env workgroup walking its
        # steps back from memory costs more than the scan kernels do (200 x 4096: explore_env +33 us against +25 us in update_net:
        # profiles/r06_c2_fused_gae_ab.txt), so longer horizons take them unless args.fused_gae / ERL_FUSED_GAE says otherwise
        self._fused_gae_explicit = hasattr(args, "fused_gae") or "ERL_FUSED_GAE" in _os.environ
        self._last_state_token = None
        self.ppo_arith = str(getattr(args, "ppo_arith", "auto"))
        assert self.ppo_arith in ("auto", "f32", "split"), f"args.ppo_arith = {self.ppo_arith!r}"
        if self._wide and self.ppo_arith == "f32":
            # the (256, h2) minibatch kernel exists in split arithmetic only: an agent that asks for the fp32 MFMA (the validation
            # mode) gets the layered path's fp32 GEMMs instead of silently running the split kernel
            self._wide = False
            self._fused_update = self._fused
        # which actor objective the kernels differentiate: the reference's sign-dependent scale (AgentPPO.py:199, default) or,
        # with args.canonical_ppo = True, the textbook min(r A, clamp(r, 1 - clip, 1 + clip) A) of
        # helloworld/helloworld_PPO_single_file.py:337-339
        self._objective = _hip.PPO_OBJ_CANONICAL if getattr(args, "canonical_ppo", False) else _hip.PPO_OBJ_REFERENCE

        from .. import ops  # deferred: importing the package must work without the built extension
        if self._fused:
            h1, h2 = net_dims
            self._spec_a = ops.MlpSpec(state_dim, h1, h2, action_dim, True)
            self._spec_c = ops.MlpSpec(state_dim, h1, h2, 1, False)
        else:
            self._spec_a = ops.MlpSpecN([state_dim, *net_dims, action_dim], not self._discrete)
            self._spec_c = ops.MlpSpecN([state_dim, *net_dims, 1], False)
        self._Pa, self._Pc = self._spec_a.count, self._spec_c.count    # raises for unsupported shapes
        # gradient row: [actor | critic | 4 logged values]; the fused kernels pad it to whole 128-byte lines
        self._stride = ops.ppo_slab_stride(state_dim, *net_dims, action_dim) if self._fused_update else self._Pa + self._Pc + 4
        f32 = dict(dtype=th.float32, device=self.device)
        self._flat = th.zeros(self._Pa + self._Pc, **f32)
        self._exp_avg = th.zeros_like(self._flat)
        self._exp_avg_sq = th.zeros_like(self._flat)
        self._adam_step = 0

        act = self._actor_class(net_dims=net_dims, state_dim=state_dim, action_dim=action_dim).to(self.device)
        cri = CriticPPO(net_dims=net_dims, state_dim=state_dim, action_dim=action_dim).to(self.device)
        self._flat_a = FlatNet(act, self._spec_a, self._flat[:self._Pa])
        self._flat_c = FlatNet(cri, self._spec_c, self._flat[self._Pa:])
        self._act = act
        self.cri = cri
        self.act_optimizer = FlatAdam(self._exp_avg[:self._Pa], self._exp_avg_sq[:self._Pa], self.learning_rate)
        self.cri_optimizer = FlatAdam(self._exp_avg[self._Pa:], self._exp_avg_sq[self._Pa:], self.learning_rate)

        self._slabs = None
        self._grads = None
        self._stats = None
        self._logs = None
        self._env_action = None
        # GPU-resident envs that expose `fused_rollout` get the whole horizon in ONE launch (csrc/rollout_fused.hip), which
        # also leaves the critic's values of the visited states for update_net; ERL_FUSED_ROLLOUT=0 / args.fused_rollout=False
        # keep the per-step launches (A/B runs, parity tests)
        import os
        self.fused_rollout = bool(getattr(args, "fused_rollout", os.environ.get("ERL_FUSED_ROLLOUT", "1") != "0"))
        self._rollout_cache = None
        self._norm_version = 0
        self.kernel_path = self._describe_kernel_path()
        if not getattr(args, "quiet", False) and os.environ.get("ERL_QUIET", "0") == "0":
            print(f"| {type(self).__name__}: {self.kernel_path}", flush=True)

    def _describe_kernel_path(self) -> str:
        """which kernels this agent's shapes get, and why (the limits are compile-time: include/erl_hip.h ERL_MAX_*; the layered
        path costs 2.2-2.4x per minibatch, DESIGN.md section 4 "Other kernels"; profiles/HISTORY.md "Generic-shape path")"""
        S, A, dims = self.state_dim, self.action_dim, list(self.net_dims)
        if self._fused:
            from .. import ops
            arith = ops.ppo_arith_in_use(S, dims[0], dims[1], A) if self.ppo_arith != "f32" else "f32"
            one_wave = all(d in (64, 128) for d in dims) and S <= 64 and A <= 8
            form = ("one wave per SIMD, " + ("bf16 matrix pipe with fp32-equivalent split arithmetic" if arith == "split" else "fp32 MFMA")
                    if one_wave else "8-wave fp32-MFMA form (one-wave form needs h1, h2 in {64, 128}, S <= 64, A <= 8)")
            return f"fused path (net_dims {dims}: two hidden layers of 32..{_hip.MAX_HIDDEN} in steps of 32): minibatch kernel {form}; persistent rollout on device-resident envs"
        if self._wide:
            return (f"wide fused path (net_dims {dims} = (256, 64 | 128), S <= 64, A <= 8): fused minibatch kernel with W2 streamed through LDS, "
                    "one-launch rollout step and value pre-pass")
        why = []
        if self._discrete:
            why.append("categorical policy")
        if len(dims) != 2:
            why.append(f"{len(dims)} hidden layers (fused kernels: 2" + ("; (256, 128, 64 | 128) has a fused minibatch kernel of its own" if
                       len(dims) == 3 and dims[0] == 256 and dims[1] == 128 and dims[2] in (64, 128) and S <= 64 and A <= 8 else "") + ")")
        elif not all(32 <= d <= _hip.MAX_HIDDEN and d % 32 == 0 for d in dims):
            why.append(f"hidden widths {dims} outside 32..{_hip.MAX_HIDDEN} in steps of 32" +
                       (" (wide_fused off or ppo_arith = 'f32')" if dims[0] == 256 and dims[1] in (64, 128) else ""))
        if S > _hip.MAX_STATE_DIM:
            why.append(f"state_dim {S} > {_hip.MAX_STATE_DIM}")
        if A > _hip.MAX_ACTION_DIM:
            why.append(f"action_dim {A} > {_hip.MAX_ACTION_DIM}")
        return "layered path (one MFMA GEMM launch per dense layer, ~2.3x the fused minibatch cost): " + "; ".join(why or ["shape outside the fused kernels"])

    # ---- checkpoints: AgentBase.save_or_load_agent (AgentBase.py:280-297) + the flat Adam state ------
    def save_or_load_agent(self, cwd: str, if_save: bool):
        """The reference pickles `th.optim.Adam` objects and gets moments + step back on load.  Here the optimiser files hold
        `FlatAdam` views of the flat moment buffers; on load their tensors are fresh copies, so they are copied back into
        the buffers the kernels use, the step counter (Adam bias correction) is restored and the views are re-created."""
        if if_save:
            self.act_optimizer.step_count = self.cri_optimizer.step_count = self._adam_step
        super().save_or_load_agent(cwd, if_save)
        if if_save:
            return
        steps = []
        for name, lo, n in (("act_optimizer", 0, self._Pa), ("cri_optimizer", self._Pa, self._Pc)):
            opt = getattr(self, name, None)
            m1, m2 = self._exp_avg[lo:lo + n], self._exp_avg_sq[lo:lo + n]
            if isinstance(opt, FlatAdam) and opt.exp_avg.numel() == n:
                if opt.exp_avg.data_ptr() != m1.data_ptr():
                    m1.copy_(opt.exp_avg.to(self.device, th.float32))
                    m2.copy_(opt.exp_avg_sq.to(self.device, th.float32))
                steps.append(int(opt.step_count))
            fresh = FlatAdam(m1, m2, self.learning_rate)
            fresh.step_count = steps[-1] if steps else self._adam_step
            setattr(self, name, fresh)
        if steps:
            self._adam_step = max(steps)
            self.act_optimizer.step_count = self.cri_optimizer.step_count = self._adam_step
        if self.cri is not None:
            self.cri = self.cri.to(self.device)
        self._on_act_replaced()
        self._sync_modules()

    # ---- keeping the kernels' view of the weights in sync with whatever module is installed ----------
    def _on_act_replaced(self):
        if getattr(self, "_flat_a", None) is not None and self._act is not None and not self._flat_a.is_bound(self._act):
            self._act = self._act.to(self.device)
            self._flat_a.bind(self._act)

    def _sync_modules(self):
        if not self._flat_a.is_bound(self._act):
            self._flat_a.bind(self._act)
        if not self._flat_c.is_bound(self.cri):
            self._flat_c.bind(self.cri)

    def _require_gpu(self, what: str):
        if self.device.type != "cuda":
            raise _hip.HipExtensionError(f"AgentPPO.{what} runs on the HIP kernels only; no GPU is visible "
                                         f"(device={self.device}) and there is no CPU fallback")

    # ---- rollout: AgentPPO.py:87-133 -----------------------------------------------------------------
    @_hip.on_device
    def explore_action(self, state: TEN, noise: Optional[TEN] = None) -> Tuple[TEN, TEN]:
        """(pre-tanh action, logprob) for a batch of states, through the K1 kernel."""
        from .. import ops
        self._require_gpu("explore_action")
        self._sync_modules()
        state = state.contiguous()
        n = state.shape[0]
        action = th.empty((n, self.action_dim), dtype=th.float32, device=self.device)
        logprob = th.empty((n,), dtype=th.float32, device=self.device)
        step = ops.rollout_step if self._fused else ops.mlpn_rollout_step
        step(self._flat_a.flat, self._spec_a, self._act.state_avg.data, self._act.state_std.data, state,
             noise=noise, seed=self.rng_seed, counter=self.rng_counter, out_action=action, out_logprob=logprob)
        self.rng_counter += 1
        return action, logprob

    @_hip.on_device
    def _explore_vec_env(self, env, horizon_len: int, if_random: bool = False, noise: Optional[TEN] = None):
        """H batched steps -> (states, actions, logprobs, rewards, undones, unmasks), time-major.
        `noise` (H, N, A) injects the N(0,1) draws (tests); otherwise Philox keyed by (seed, step, env)."""
        from .. import ops
        self._require_gpu("explore_env")
        self._sync_modules()
        H, N, S, A = horizon_len, self.num_envs, self.state_dim, self.action_dim
        dev = self.device
        # every element of the six buffers is written by the rollout below (row t by step t): no zero-fill launches
        # (the interpreter's time before the first launch is GPU idle time -- the previous iteration ended in a host sync -- so
        # the (H, N) planes come out of two allocator calls instead of seven)
        states = th.empty((H, N, S), dtype=th.float32, device=dev)
        actions = th.empty((H, N, A), dtype=th.float32, device=dev)
        hn = H * N
        pitch = (hn + 255) // 256 * 256                    # every plane starts on a 256-byte boundary whatever H * N is
        # planes: logprobs | rewards | values [| raw advantages | reward sums: the fused rollout's epilogue]
        fused_gae = bool(getattr(self, "fused_gae", False)) and (getattr(self, "_fused_gae_explicit", True) or H <= 128)
        planes = th.empty((5 if fused_gae else 3, pitch), dtype=th.float32, device=dev)
        logprobs, rewards = planes[0, :hn].view(H, N), planes[1, :hn].view(H, N)
        flags = th.empty((2, pitch), dtype=th.bool, device=dev)
        terminals, truncates = flags[0, :hn].view(H, N), flags[1, :hn].view(H, N)
        if self._env_action is None or self._env_action.shape != (N, A):
            self._env_action = th.empty((N, A), dtype=th.float32, device=dev)
        env_action = self._env_action
        P, spec = self._flat_a.flat, self._spec_a
        avg, std = self._act.state_avg.data, self._act.state_std.data
        native = hasattr(env, "step_into")

        state = self.last_state
        assert state.shape == (N, S), f"last_state {tuple(state.shape)} != {(N, S)}"
        state = state.to(dev, th.float32).contiguous()
        if hasattr(env, "raw_stepper") and state.data_ptr() != env.state.data_ptr():
            tok = self._last_state_token
            # `state` is the very tensor the last fused rollout of THIS env wrote as its copy of the final state, nobody has written
            # to it, and the env has not moved since: the live buffer already holds it (no copy-back launch)
            same = (tok is not None and tok[0] is self.last_state and tok[1] == self.last_state._version and tok[2] is env
                    and tok[3] == getattr(env, "state_epoch", None) and tok[4] == (id(env.state), env.state._version))
            if not same:
                env.state.copy_(state)             # the env owns the live state buffer; keep it authoritative
                if hasattr(env, "state_epoch"):
                    env.state_epoch += 1
        self._last_state_token = None
        self._rollout_cache = None
        if (self.fused_rollout and self._fused and hasattr(env, "fused_rollout") and getattr(env, "num_envs", None) == N
                and getattr(env, "device", None) == dev
                and _hip.lib().erl_rollout_fused_supported(S, self.net_dims[0], self.net_dims[1], A)):
            # one persistent launch for all H steps: policy, env, buffer rows, reward scaling, flag inversion AND the critic's
            # values of every visited state + cri(last_state) (update_net's pre-pass, AgentPPO.py:141-143, :219-220)
            undones, unmasks = terminals, truncates          # the kernel writes the inverted flags straight into these planes
            values = planes[2, :hn].view(H, N)
            next_value = th.empty((N,), dtype=th.float32, device=dev)
            noise = None if noise is None else noise.contiguous()
            epilogue, adv_raw, ret, stats = None, None, None, None
            last_out = th.empty((N, S), dtype=th.float32, device=dev) if self.snapshot_last_state else None
            if fused_gae:
                adv_raw, ret = planes[3, :hn].view(H, N), planes[4, :hn].view(H, N)
                # [8 doubles: the folded sums | 3 per 16-env workgroup: the epilogue's partial sums]
                n_parts = int(_hip.lib().erl_rollout_gae_partials(N))
                sums = th.empty(8 + 3 * n_parts, dtype=th.float64, device=dev)
                stats, gae_parts = sums[:8], sums[8:]
            if last_out is not None or fused_gae:
                epilogue = (last_out, adv_raw, ret, None, gae_parts if fused_gae else None, float(self.gamma),
                            float(self.lambda_gae_adv), bool(self.if_use_v_trace))
            env.fused_rollout(self, H, noise, (states, actions, logprobs, rewards, undones, unmasks), values, next_value,
                              **({} if epilogue is None else {"epilogue": epilogue}))
            self.rng_counter += H
            # the agent's copy of the final state is written by the rollout kernel itself (no clone, and no copy back at the next
            # call while nobody touches either side: _last_state_token); args.snapshot_last_state = False hands out the env's
            # live buffer instead
            self.last_state = last_out if last_out is not None else env.state
            if last_out is not None:
                self._last_state_token = (last_out, last_out._version, env, getattr(env, "state_epoch", None), (id(env.state), env.state._version))
            self._rollout_cache = dict(states=states, values=values, next_value=next_value, last_state=self.last_state,
                                       key=self._value_cache_key(states, self.last_state))
            if fused_gae:
                self._rollout_cache.update(adv=adv_raw, ret=ret, stats=stats, parts=gae_parts, n_parts=n_parts, rewards=rewards,
                                           undones=undones, unmasks=unmasks, adv_key=self._adv_cache_key(rewards, undones, unmasks))
            return states, actions, logprobs, rewards, undones, unmasks
        rollout_step = ops.rollout_step if self._fused else ops.mlpn_rollout_step
        if hasattr(env, "raw_stepper") and self._fused:
            # GPU-resident env: both launches of a step go straight to the C ABI on raw pointers (no tensor views, no
            # per-call argument checks on the interpreter's launch path)
            fn = _hip.lib().erl_rollout_step_f32
            env_step = env.raw_stepper()
            sp = _hip.stream_ptr()
            pP, pavg, pstd, pst, pea = (_hip.ptr(x, th.float32) for x in (P, avg, std, env.state, env_action))
            p_s, p_a, p_l, p_r = states.data_ptr(), actions.data_ptr(), logprobs.data_ptr(), rewards.data_ptr()
            p_te, p_tr = terminals.data_ptr(), truncates.data_ptr()
            noise = None if noise is None else noise.contiguous()      # keep the (possibly fresh) contiguous copy alive for all H launches
            p_n = None if noise is None else _hip.ptr(noise, th.float32)
            h1, h2 = spec.h1, spec.h2
            seed = self.rng_seed & (2 ** 64 - 1)
            for t in range(H):
                rc = fn(pP, pavg, pstd, S, h1, h2, A, pst, N, None if p_n is None else p_n + 4 * t * N * A, seed,
                        self.rng_counter & (2 ** 64 - 1), p_s + 4 * t * N * S, p_a + 4 * t * N * A, p_l + 4 * t * N, pea, sp)
                if rc:
                    _hip.check(rc, "erl_rollout_step_f32")
                self.rng_counter += 1
                env_step(pea, p_r + 4 * t * N, p_te + t * N, p_tr + t * N, sp)
            state = env.state
            native = True
        else:
            for t in range(H):
                rollout_step(P, spec, avg, std, state, noise=None if noise is None else noise[t], seed=self.rng_seed,
                             counter=self.rng_counter, out_state=states[t], out_action=actions[t],
                             out_logprob=logprobs[t], out_env_action=env_action)
                self.rng_counter += 1
                if native:     # GPU-resident env writes its outputs straight into row t (no copies, no host sync)
                    state = env.step_into(env_action, rewards[t], terminals[t], truncates[t])
                else:          # any env honouring the reference protocol (device tensors, auto-reset)
                    state, reward, terminal, truncate, _ = env.step(env_action)
                    state = state.to(dev, th.float32).contiguous()
                    rewards[t] = reward
                    terminals[t] = terminal
                    truncates[t] = truncate
        self.last_state = state.clone() if native else state
        if self.reward_scale != 1.0:
            rewards *= self.reward_scale
        undones = th.logical_not(terminals)
        unmasks = th.logical_not(truncates)
        return states, actions, logprobs, rewards, undones, unmasks

    @_hip.on_device
    def _explore_one_env(self, env, horizon_len: int, if_random: bool = False):
        """single (numpy, non-vectorised) env: AgentPPO.py:34-85; the policy still runs on the GPU."""
        from .. import ops
        self._require_gpu("explore_env")
        self._sync_modules()
        H, S, A, dev = horizon_len, self.state_dim, self.action_dim, self.device
        states = th.zeros((H, 1, S), dtype=th.float32, device=dev)
        actions = th.zeros((H, 1, A), dtype=th.float32, device=dev)
        logprobs = th.zeros((H, 1), dtype=th.float32, device=dev)
        rewards = th.zeros((H, 1), dtype=th.float32, device=dev)
        terminals = th.zeros((H, 1), dtype=th.bool, device=dev)
        truncates = th.zeros((H, 1), dtype=th.bool, device=dev)
        env_action = th.empty((1, A), dtype=th.float32, device=dev)
        state = self.last_state.to(dev, th.float32).reshape(1, S).contiguous()
        for t in range(H):
            (ops.rollout_step if self._fused else ops.mlpn_rollout_step)(
                self._flat_a.flat, self._spec_a, self._act.state_avg.data, self._act.state_std.data, state,
                seed=self.rng_seed, counter=self.rng_counter, out_state=states[t], out_action=actions[t],
                out_logprob=logprobs[t], out_env_action=env_action)
            self.rng_counter += 1
            ary_state, reward, terminal, truncate, _ = env.step(env_action[0].cpu().numpy())
            if terminal or truncate:
                ary_state, _ = env.reset()
            state = th.as_tensor(ary_state, dtype=th.float32, device=dev).reshape(1, S)
            rewards[t, 0] = float(reward)
            terminals[t, 0] = bool(terminal)
            truncates[t, 0] = bool(truncate)
        self.last_state = state
        rewards *= self.reward_scale
        return states, actions, logprobs, rewards, th.logical_not(terminals), th.logical_not(truncates)

    # ---- GAE: AgentPPO.py:207-232 ---------------------------------------------------------------------
    @_hip.on_device
    def get_values(self, states: TEN) -> TEN:
        """cri(states).squeeze(-1) for states (..., S) (the value pre-pass of :141-143, in one launch)."""
        from .. import ops
        self._require_gpu("get_values")
        self._sync_modules()
        fwd = ops.value_forward if self._fused else ops.mlpn_value_forward
        return fwd(self._flat_c.flat, self._spec_c, self.cri.state_avg.data, self.cri.state_std.data, states.contiguous())

    @_hip.on_device
    def get_advantages(self, states: TEN, rewards: TEN, undones: TEN, unmasks: TEN, values: TEN) -> TEN:
        """Same signature and side effects as the reference: returns `advantages` (H, N) and applies the
        truncation fix-up to the caller's `rewards` / `undones` in place."""
        adv, _ = self._gae(rewards, undones, unmasks, values, with_ret=False)
        return adv

    get_reward_sum_gae = get_advantages   # older spelling used by the north star / stale tests

    def _value_cache_key(self, states: TEN, last_state: TEN):
        """what the values left by the fused rollout depend on: the very tensors, the critic's weights (Adam step, in-place
        torch edits of the flat block, the module object) and the normalisation vectors."""
        return (states.data_ptr(), states._version, tuple(states.shape), last_state.data_ptr(), last_state._version,
                self._adam_step, self._flat._version, id(self.cri), self._norm_version,
                sum(p._version for p in self.cri.parameters()))      # in-place edits / load_state_dict of the critic

    def _adv_cache_key(self, rewards: TEN, undones: TEN, unmasks: TEN):
        """what the advantages left by the rollout's epilogue depend on besides the values: the flag / reward tensors as they
        were written (an in-place edit by the caller bumps _version) and the scan's hyper-parameters"""
        return (rewards.data_ptr(), rewards._version, undones.data_ptr(), undones._version, unmasks.data_ptr(), unmasks._version,
                float(self.gamma), float(self.lambda_gae_adv), bool(self.if_use_v_trace))

    def _cached_advantages(self, rewards: TEN, undones: TEN, unmasks: TEN):
        """(raw advantages, reward_sums, stats block, partial sums, their count) computed by the fused rollout's epilogue for exactly these tensors, else None
        (call after _cached_values said yes)."""
        c = self._rollout_cache
        if c is None or "adv" not in c or rewards is not c["rewards"] or undones is not c["undones"] or unmasks is not c["unmasks"]:
            return None
        if c["adv_key"] != self._adv_cache_key(rewards, undones, unmasks):
            return None
        return c["adv"], c["ret"], c["stats"], c["parts"], c["n_parts"]

    def _cached_values(self, states: TEN):
        """(values, next_value) computed by the fused rollout for exactly this buffer and this critic, else None."""
        c = self._rollout_cache
        if c is None or states is not c["states"] or self.last_state is not c["last_state"]:
            return None
        if not self._flat_c.is_bound(self.cri) or c["key"] != self._value_cache_key(states, self.last_state):
            return None
        return c["values"], c["next_value"]

    def _gae(self, rewards, undones, unmasks, values, with_ret=True, stats=None, next_value=None):
        from .. import ops
        self._require_gpu("get_advantages")
        if next_value is None:
            next_value = self.get_values(self.last_state.to(self.device, th.float32))
        return ops.gae_scan(rewards, undones, unmasks, values, next_value, float(self.gamma), float(self.lambda_gae_adv),
                            use_v_trace=bool(self.if_use_v_trace), mutate=True, algo=self.gae_algo, with_ret=with_ret,
                            stats=stats)

    # ---- update: AgentPPO.py:135-205 ------------------------------------------------------------------
    @_hip.on_device
    def update_net(self, buffer, ids: Optional[TEN] = None, lazy: bool = False) -> Tuple[float, float, float]:
        """One PPO update on the rollout `buffer`; returns (obj_critic, obj_surrogate, obj_entropy) means.
        `ids` (update_times, batch_size) int64 injects the minibatch indices (tests); otherwise they are drawn
        with th.randint(H*N, ...) like the reference."""
        from .. import ops, parallel
        self._require_gpu("update_net")
        self._sync_modules()
        # the minibatch kernels' arithmetic travels with every call (ERL_PPO_MODE: objective | arith << 8, ABI 17), not through the
        # library's process-wide setting: two agents of one process keep their own `ppo_arith`
        mode = self._objective | ({"auto": 0, "f32": 1, "split": 2}[self.ppo_arith] << 8)
        states, actions, logprobs, rewards, undones, unmasks = buffer
        H, N = rewards.shape
        dev = self.device
        if self._stats is None:
            self._stats = th.zeros(8, dtype=th.float64, device=dev)

        dp = self.world_size > 1 or parallel.force_dp()
        # the job's exchange route (selected once, by a self-test: parallel.gradient_comm); None: torch.distributed
        comm = parallel.gradient_comm(self._stride) if dp else None
        c_loop = self._fused_update and (not dp or comm is not None)       # the whole minibatch loop in one C call (below)
        cached = self._cached_values(states)
        from_rollout = None
        if cached is not None:                                                        # left by the fused rollout (same critic)
            values, next_value = cached
            if c_loop:
                from_rollout = self._cached_advantages(rewards, undones, unmasks)     # ... and get_advantages + its sums
            self._rollout_cache = None                                                # the GAE below mutates rewards / undones
        else:
            values, next_value = self.get_values(states), None                        # (H, N)
        if from_rollout is not None:
            # raw advantages, reward sums and the five sums come from the rollout's epilogue; rewards / undones still are what
            # explore_env returned: get_advantages' in-place fix-up of truncated steps rides the last launch (erl_ppo_finish_f32)
            advantages, reward_sums, stats, adv_parts, n_parts = from_rollout
            self._stats = stats                # (the sums this update normalised with stay inspectable, as on the other path)
            if self.world_size > 1:            # the sums are all-reduced below: fold them now (one small launch)
                ops.adv_stats_fold(adv_parts, n_parts, H, N, stats)
                adv_parts = None
        else:
            stats, adv_parts, n_parts = self._stats, None, 0
            advantages, reward_sums = self._gae(rewards, undones, unmasks, values, stats=stats, next_value=next_value)
        if self.world_size > 1:                                                       # one normalisation for the whole job:
            if comm is not None:                                                      # the 5 sums ride the gradient's route
                comm.all_reduce_sum(stats)
            else:
                parallel.all_reduce_sum(stats)
        if not c_loop:     # (the C loop's minibatch kernels normalise at their row loads from `stats`: no launch for it)
            advantages = ops.adv_normalize(advantages, stats, out=advantages)
        assert logprobs.shape == advantages.shape == reward_sums.shape == (H, N)

        B = int(self.batch_size)
        update_times = int(H * self.repeat_times / B)
        assert update_times >= 1
        if ids is None:
            ids = th.randint(H * N, size=(update_times, B), device=dev)
        assert ids.shape == (update_times, B) and ids.dtype == th.int64
        n_slabs = self._n_slabs(B) if self._fused_update else 0
        if self._fused_update and (self._slabs is None or self._slabs.shape[0] != n_slabs):
            self._slabs = th.empty((n_slabs, self._stride), dtype=th.float32, device=dev)
        if self._grads is None or self._grads.shape[0] < update_times:
            self._grads = th.empty((update_times, self._stride), dtype=th.float32, device=dev)
        a, c = self._act, self.cri
        groups = [(0, self._Pa), (self._Pa, self._Pc)]
        inv_batch = 1.0 / B
        grad_scale = 1.0 / self.world_size
        if not self._fused_update:      # generic-shape networks: layered path, summed gradient written directly
            for k in range(update_times):
                g = self._grads[k]
                step = ops.mlpn_ppo_step_discrete if self._discrete else ops.mlpn_ppo_step
                extra = {} if self._discrete else {"objective": self._objective}
                step(self._flat_a.flat, self._flat_c.flat, a.state_avg.data, a.state_std.data, c.state_avg.data,
                     c.state_std.data, self._spec_a, states, actions, unmasks, logprobs, advantages, reward_sums,
                     ids[k], float(self.ratio_clip), self.lambda_entropy_value, inv_batch, g, **extra)
                if comm is not None:
                    comm.all_reduce_sum(g)
                elif dp:
                    parallel.all_reduce_sum(g)
                self._adam_step += 1
                ops.clip_adam(self._flat, g, self._exp_avg, self._exp_avg_sq, groups, self._adam_step, float(self.learning_rate),
                              float(self.clip_grad_norm), grad_scale=grad_scale)
            self.act_optimizer.step_count = self.cri_optimizer.step_count = self._adam_step
            logs = self._grads[:update_times, self._Pa + self._Pc:self._Pa + self._Pc + 3].mean(dim=0) * grad_scale
            obj_critic, obj_actor, obj_entropy = (float(x) for x in logs.cpu())
            _hip.check_async_faults()          # the stream is drained: a lost look-back predecessor (NaN advantages) raises here
            return obj_critic, obj_actor, obj_entropy
        h1, h2 = self.net_dims
        if not dp or comm is not None:  # the whole minibatch loop is enqueued by one C call (no interpreter on the launch
            # path); data-parallel ranks pass the library's communicator: the exchange is part of the slab-reduction launch
            # (peer-to-peer route) or an RCCL all-reduce on the same stream
            ops.ppo_update(self._flat, self._exp_avg, self._exp_avg_sq, a.state_avg.data, a.state_std.data, c.state_avg.data,
                           c.state_std.data, self.state_dim, h1, h2, self.action_dim, states, actions, unmasks, logprobs,
                           advantages, reward_sums, ids, float(self.ratio_clip), self.lambda_entropy_value, self._slabs,
                           self._grads, self._adam_step + 1, float(self.learning_rate), float(self.clip_grad_norm), comm=comm,
                           objective=mode, adv_stats=stats, adv_partials=adv_parts, n_partials=n_parts)
            self._adam_step += update_times
        else:                           # data parallel through torch.distributed (gloo tests, ERL_DP_COLLECTIVE=torch)
            # raw pointers + direct C-ABI calls: the interpreter spends ~3 us per launch instead of ~10 (ptr checks, views)
            L, sp = _hip.lib(), _hip.stream_ptr()
            pf = _hip.ptr(self._flat, th.float32)
            p_avg_a, p_std_a, p_avg_c, p_std_c = (_hip.ptr(x) for x in (a.state_avg.data, a.state_std.data, c.state_avg.data,
                                                                         c.state_std.data))
            p_s, p_ac, p_um = _hip.ptr(states, th.float32), _hip.ptr(actions, th.float32), _hip.flag_ptr(unmasks)
            p_lp, p_adv, p_rs = (_hip.ptr(x, th.float32) for x in (logprobs, advantages, reward_sums))
            p_ids, p_slabs, p_g = _hip.ptr(ids, th.int64), _hip.ptr(self._slabs, th.float32), _hip.ptr(self._grads, th.float32)
            p_m1, p_m2 = _hip.ptr(self._exp_avg, th.float32), _hip.ptr(self._exp_avg_sq, th.float32)
            import ctypes
            off = (ctypes.c_int64 * 2)(0, self._Pa)
            ln = (ctypes.c_int64 * 2)(self._Pa, self._Pc)
            S_, A_, clip, lam_e = self.state_dim, self.action_dim, float(self.ratio_clip), self.lambda_entropy_value
            lr, max_norm, stride = float(self.learning_rate), float(self.clip_grad_norm), self._stride
            for k in range(update_times):
                rc = L.erl_ppo_step_f32(pf, pf + 4 * self._Pa, p_avg_a, p_std_a, p_avg_c, p_std_c, S_, h1, h2, A_, p_s, p_ac, p_um,
                                        p_lp, p_adv, p_rs, H, N, p_ids + 8 * k * B, B, clip, lam_e, inv_batch, mode, p_slabs,
                                        n_slabs, sp)
                rc = rc or L.erl_grad_reduce_f32(p_slabs, n_slabs, stride, p_g + 4 * k * stride, sp)
                if rc:
                    _hip.check(rc, "erl_ppo_step_f32 / erl_grad_reduce_f32")
                parallel.all_reduce_sum(self._grads[k])
                self._adam_step += 1
                # the same two-launch tail as the C loop (partial norms, then clip + Adam): bit-identical weights on every route
                rc = L.erl_grad_sq_partials_f32(p_g + 4 * k * stride, stride, off, ln, 2, grad_scale, sp)
                rc = rc or L.erl_clip_adam_partials_f32(pf, p_g + 4 * k * stride, p_m1, p_m2, stride, off, ln, 2, self._adam_step, lr,
                                                        0.9, 0.999, 1e-8, max_norm, grad_scale, sp)
                if rc:
                    _hip.check(rc, "erl_grad_sq_partials_f32 / erl_clip_adam_partials_f32")
        self.act_optimizer.step_count = self.cri_optimizer.step_count = self._adam_step
        if self._logs is None:
            self._logs = th.empty(4, dtype=th.float32, device=dev)
        if from_rollout is not None:           # the logged means + get_advantages' side effect on rewards / undones (AgentPPO.py:211-214)
            _hip.check(_hip.lib().erl_ppo_finish_f32(_hip.ptr(self._grads, th.float32), self._stride, self._Pa + self._Pc, update_times,
                                                     grad_scale, _hip.ptr(self._logs, th.float32), _hip.ptr(rewards, th.float32),
                                                     _hip.flag_ptr(undones), _hip.flag_ptr(unmasks), _hip.ptr(values, th.float32), H * N,
                                                     _hip.stream_ptr()), "erl_ppo_finish_f32")
        else:
            _hip.check(_hip.lib().erl_ppo_logs_mean_f32(_hip.ptr(self._grads, th.float32), self._stride, self._Pa + self._Pc, update_times,
                                                        grad_scale, _hip.ptr(self._logs, th.float32), _hip.stream_ptr()), "erl_ppo_logs_mean_f32")
        if lazy:
            # the three logged means travel to a pinned host block behind the update's last kernel; PendingLogs.result() waits for that
            # copy only.  The caller enqueues the next rollout first (train_agent does): the GPU no longer idles behind this host sync
            # while the interpreter works its way to the next launch (~70 us of a 2.3 ms iteration at config 4).
            return PendingLogs(self)
        obj_critic, obj_actor, obj_entropy = self._logs[:3].tolist()                  # the only host sync of update_net
        _hip.check_async_faults()              # the stream is drained: a lost look-back predecessor (NaN advantages) raises here
        return obj_critic, obj_actor, obj_entropy

    def _n_slabs(self, batch_size: int) -> int:
        from .. import ops
        return ops.ppo_num_slabs(batch_size)

    # ---- running state normalisation: AgentPPO.py:234-249 (never called by the reference's loops) -----
    def update_avg_std_for_normalization(self, states: TEN):
        tau = self.state_value_tau
        if tau == 0:
            return
        self._norm_version += 1
        state_avg = states.mean(dim=0, keepdim=True)
        state_std = states.std(dim=0, keepdim=True)
        with th.no_grad():
            self.act.state_avg[:] = self.act.state_avg * (1 - tau) + state_avg * tau
            self.act.state_std[:] = (self.act.state_std * (1 - tau) + state_std * tau).clamp_min(1e-4)
            self.cri.state_avg[:] = self.act.state_avg
            self.cri.state_std[:] = self.act.state_std


class AgentA2C(AgentPPO):
    """A2C (elegantrl/agents/AgentPPO.py:252-303): AgentPPO's rollout, GAE and advantage normalisation, minibatches of whole
    TIME ROWS (`indices = randint(buffer_size)`, :289) and the un-clipped objective `mean(advantage * new_logprob)` (:301)
    without unmask and without an entropy term; `update_net` returns (obj_critic, obj_actor, 0).

    Like the reference's this agent is for ONE environment: with `num_envs > 1` the reference's `states[indices]` keeps the
    env axis, `cri(state).squeeze(1)` no longer squeezes and `criterion(value, reward_sum)` broadcasts (B, N, 1) against
    (B, N) -- the class only produces the intended objective for `num_envs == 1`, so that is what is supported here (a
    ValueError otherwise).  With one env the reference's `get_logprob_entropy(...).sum(1)` sums over the env axis (size 1)
    instead of the action axis, so `new_logprob` is per action dimension and the mean runs over (batch, action_dim): the
    objective is mean_B(adv * logp) / action_dim -- reproduced (ERL_PPO_OBJ_A2C, csrc/ppo_objective.h)."""
    supports_lazy_logs = False

    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int, gpu_id: int = 0, args: Config = None):
        super().__init__(net_dims, state_dim, action_dim, gpu_id, args)
        if self.num_envs != 1:
            raise ValueError("AgentA2C follows the reference's time-row minibatches, which are only well formed for num_envs == 1 "
                             "(elegantrl/agents/AgentPPO.py:289-303); use AgentPPO for vectorised envs")
        self._objective = _hip.PPO_OBJ_A2C

    def update_net(self, buffer, ids: Optional[TEN] = None) -> Tuple[float, float, float]:
        """`ids`, if given, are the time indices (update_times, batch_size) in [0, horizon_len)."""
        obj_critic, obj_actor, _ = super().update_net(buffer, ids=ids)      # H * N == H: ids are time rows, env 0
        return obj_critic, obj_actor, 0


class AgentDiscretePPO(AgentPPO):
    """PPO with a categorical policy (elegantrl/agents/AgentPPO.py:305-320, ActorDiscretePPO :393-422) on the layered
    path: logits from the layered path's MFMA GEMMs, inverse-CDF sampling / log-prob and the clipped-scale objective
    with its state-dependent entropy in hand-written kernels (erl_mlpn_rollout_step_discrete_f32,
    erl_mlpn_ppo_step_discrete_f32).  Rollout dtypes as the reference: actions (H, N) int32, env receives int64."""
    _discrete = True
    _actor_class = ActorDiscretePPO

    def __init__(self, net_dims: List[int], state_dim: int, action_dim: int, gpu_id: int = 0, args: Config = None):
        args = Config() if args is None else args
        super().__init__(net_dims, state_dim, action_dim, gpu_id, args)
        self.lambda_entropy_value = float(getattr(args, "lambda_entropy", 0.01))          # AgentPPO.py:318
        self.lambda_entropy = th.tensor(self.lambda_entropy_value, dtype=th.float32, device=self.device)

    @_hip.on_device
    def explore_action(self, state: TEN, uniform: Optional[TEN] = None) -> Tuple[TEN, TEN]:
        """(action int32 (n,), logprob (n,)); `uniform` (n,) injects the U[0,1) draws (tests)."""
        from .. import ops
        self._require_gpu("explore_action")
        self._sync_modules()
        state = state.contiguous()
        n = state.shape[0]
        action = th.empty((n,), dtype=th.int32, device=self.device)
        logprob = th.empty((n,), dtype=th.float32, device=self.device)
        ops.mlpn_rollout_step_discrete(self._flat_a.flat, self._spec_a, self._act.state_avg.data, self._act.state_std.data, state,
                                       uniform=uniform, seed=self.rng_seed, counter=self.rng_counter, out_action=action,
                                       out_logprob=logprob)
        self.rng_counter += 1
        return action, logprob

    @_hip.on_device
    def _explore_vec_env(self, env, horizon_len: int, if_random: bool = False, noise: Optional[TEN] = None):
        """H batched steps -> (states, actions int32 (H, N), logprobs, rewards, undones, unmasks); `noise` (H, N) injects
        the uniform draws."""
        from .. import ops
        self._require_gpu("explore_env")
        self._sync_modules()
        H, N, S, dev = horizon_len, self.num_envs, self.state_dim, self.device
        states = th.empty((H, N, S), dtype=th.float32, device=dev)
        actions = th.empty((H, N), dtype=th.int32, device=dev)
        logprobs = th.empty((H, N), dtype=th.float32, device=dev)
        rewards = th.empty((H, N), dtype=th.float32, device=dev)
        terminals = th.empty((H, N), dtype=th.bool, device=dev)
        truncates = th.empty((H, N), dtype=th.bool, device=dev)
        env_action = th.empty((N,), dtype=th.int64, device=dev)
        state = self.last_state
        assert state.shape == (N, S), f"last_state {tuple(state.shape)} != {(N, S)}"
        state = state.to(dev, th.float32).contiguous()
        for t in range(H):
            ops.mlpn_rollout_step_discrete(self._flat_a.flat, self._spec_a, self._act.state_avg.data, self._act.state_std.data, state,
                                           uniform=None if noise is None else noise[t].contiguous(), seed=self.rng_seed,
                                           counter=self.rng_counter, out_state=states[t], out_action=actions[t],
                                           out_logprob=logprobs[t], out_env_action=env_action)
            self.rng_counter += 1
            state, reward, terminal, truncate, _ = env.step(env_action)
            state = state.to(dev, th.float32).contiguous()
            rewards[t] = reward
            terminals[t] = terminal
            truncates[t] = truncate
        self.last_state = state
        if self.reward_scale != 1.0:
            rewards *= self.reward_scale
        return states, actions, logprobs, rewards, th.logical_not(terminals), th.logical_not(truncates)

    @_hip.on_device
    def _explore_one_env(self, env, horizon_len: int, if_random: bool = False):
        """single (numpy) env, AgentPPO.py:34-85 with if_discrete: actions (H, 1) int32, env.step gets a python int."""
        from .. import ops
        self._require_gpu("explore_env")
        self._sync_modules()
        H, S, dev = horizon_len, self.state_dim, self.device
        states = th.zeros((H, 1, S), dtype=th.float32, device=dev)
        actions = th.zeros((H, 1), dtype=th.int32, device=dev)
        logprobs = th.zeros((H, 1), dtype=th.float32, device=dev)
        rewards = th.zeros((H, 1), dtype=th.float32, device=dev)
        terminals = th.zeros((H, 1), dtype=th.bool, device=dev)
        truncates = th.zeros((H, 1), dtype=th.bool, device=dev)
        env_action = th.empty((1,), dtype=th.int64, device=dev)
        state = self.last_state.to(dev, th.float32).reshape(1, S).contiguous()
        for t in range(H):
            ops.mlpn_rollout_step_discrete(self._flat_a.flat, self._spec_a, self._act.state_avg.data, self._act.state_std.data, state,
                                           seed=self.rng_seed, counter=self.rng_counter, out_state=states[t], out_action=actions[t],
                                           out_logprob=logprobs[t], out_env_action=env_action)
            self.rng_counter += 1
            ary_state, reward, terminal, truncate, _ = env.step(int(env_action[0].item()))
            if terminal or truncate:
                ary_state, _ = env.reset()
            state = th.as_tensor(ary_state, dtype=th.float32, device=dev).reshape(1, S)
            rewards[t, 0] = float(reward)
            terminals[t, 0] = bool(terminal)
            truncates[t, 0] = bool(truncate)
        self.last_state = state
        rewards *= self.reward_scale
        return states, actions, logprobs, rewards, th.logical_not(terminals), th.logical_not(truncates)


class AgentDiscreteA2C(AgentDiscretePPO):
    """elegantrl/agents/AgentPPO.py:332-342.  The reference's class derives from AgentDiscretePPO and overrides nothing but the
    constructor (which rebuilds the same ActorDiscretePPO / CriticPPO / optimisers): `update_net` / `update_objectives` resolve to
    AgentPPO's -- AgentA2C is not in its MRO -- so it trains with the clipped-scale PPO objective and the categorical entropy
    term, exactly like AgentDiscretePPO.  Kept as its own class so that `Config(agent_class=AgentDiscreteA2C, ...)` scripts run."""
