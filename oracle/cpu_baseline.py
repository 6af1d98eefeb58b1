"""The `cpu_baseline` leg of bench.py: the reference's own CPU path timed on the host cores, stage by stage.
ORACLE / BASELINE ONLY (see oracle/__init__.py): nothing in the product imports this.

Two kinds:
  * "reference" -- /root/reference's own classes (elegantrl.agents.AgentPPO / AgentSAC, elegantrl.train.replay_buffer.ReplayBuffer)
    imported and run with gpu_id = -1.  Only where the reference is mounted (the authoring container; never the GPU box).
  * "port"      -- oracle/torch_port.py (PPO) / oracle/sac_torch.py + a torch ring (SAC): the same ATen op sequence restated
    (pinned to the reference's outputs by tests/test_oracle_golden.py, tests/test_sac.py).
Both run the SAME workload on the same synthetic env (TorchSynEnv behind the reference's env protocol) and report the time
spent in the reference's hot-path functions:
    explore_env      AgentPPO._explore_vec_env   elegantrl/agents/AgentPPO.py:87-129   (AgentBase._explore_vec_env :130-170 for SAC)
    get_advantages   AgentPPO.get_advantages     elegantrl/agents/AgentPPO.py:207-232  (inside update_net)
    update_net       AgentPPO.update_net         elegantrl/agents/AgentPPO.py:135-171  (AgentBase.update_net :172-189 for SAC)
    buffer_update    ReplayBuffer.update         elegantrl/train/replay_buffer.py:78-118
    sample           ReplayBuffer.sample         elegantrl/train/replay_buffer.py:120-134 (inside update_net)
"""
from __future__ import annotations

import os
import sys
import time
from typing import Dict

import torch as th

REF = os.environ.get("ERL_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF, "elegantrl", "agents", "AgentPPO.py"))


def _import_reference():
    """the reference's package, not this repository's `elegantrl` import alias (same name): REF goes first on sys.path and any
    already-imported alias is dropped.  Call only in a process that does not use the product (bench.py's CPU subprocess)."""
    for k in [k for k in sys.modules if k == "elegantrl" or k.startswith("elegantrl.")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True           # the reference tree is read-only
    import elegantrl  # noqa: F401
    assert os.path.realpath(elegantrl.__file__).startswith(os.path.realpath(REF)), elegantrl.__file__
    return elegantrl


class StageTimer:
    """wall seconds per named stage, accumulated over calls (host clocks: everything here is synchronous CPU work)"""

    def __init__(self):
        self.s: Dict[str, float] = {}
        self.n: Dict[str, int] = {}

    def wrap(self, name, fn):
        def inner(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                self.s[name] = self.s.get(name, 0.0) + time.perf_counter() - t0
                self.n[name] = self.n.get(name, 0) + 1
        return inner

    def reset(self):
        self.s.clear()
        self.n.clear()

    def report(self):
        return {k: round(v, 4) for k, v in self.s.items()}


def _port_env(env_kind, num_envs, state_dim, action_dim, max_step, seed=0):
    from oracle.torch_port import TorchPendulumEnv, TorchSynEnv
    if env_kind == "pendulum":
        return TorchPendulumEnv(num_envs, max_step, seed=seed)
    return TorchSynEnv(num_envs, state_dim, action_dim, max_step, seed=seed)


class RefProtocolEnv:
    """oracle.torch_port's CPU env twins behind the reference's vectorised-env protocol:
    reset() -> (state, info); step(action) -> (state, reward, terminal, truncate, info); auto-reset inside."""

    def __init__(self, num_envs, state_dim, action_dim, max_step, seed=0, env_kind="syn"):
        self.env = _port_env(env_kind, num_envs, state_dim, action_dim, max_step, seed)
        self.num_envs, self.state_dim, self.action_dim, self.max_step = num_envs, state_dim, action_dim, max_step
        self.if_discrete = False

    def reset(self):
        return self.env.reset(), {}

    def step(self, action):
        s, r, te, tr = self.env.step(action)
        return s, r, te, tr, {}


# ---------------------------------------------------------------------------------------------------------
# PPO (BASELINE configs[1] / [3] / [4])
# ---------------------------------------------------------------------------------------------------------
def ppo(kind: str, *, N, S, A, H, B, update_times, net_dims, iters, hyper=None, max_step=1000, env_kind="syn") -> dict:
    hyper = dict(hyper or {})
    th.manual_seed(0)
    timer = StageTimer()
    if kind == "reference":
        _import_reference()
        from elegantrl.agents import AgentPPO
        from elegantrl.train.config import Config
        args = Config(AgentPPO, None, {"env_name": "syn", "num_envs": N, "max_step": max_step, "state_dim": S, "action_dim": A,
                                       "if_discrete": False})
        args.net_dims = list(net_dims)
        args.horizon_len, args.batch_size = H, B
        args.repeat_times = update_times * B / H             # int(H * repeat_times / B) = update_times  (AgentPPO.py:159)
        for k, v in hyper.items():
            setattr(args, k, v)
        agent = AgentPPO(args.net_dims, S, A, gpu_id=-1, args=args)
        env = RefProtocolEnv(N, S, A, max_step, env_kind=env_kind)
        agent.last_state = env.reset()[0]
        agent.explore_env = timer.wrap("explore_env", agent.explore_env)
        agent.get_advantages = timer.wrap("get_advantages", agent.get_advantages)
        agent.update_net = timer.wrap("update_net", agent.update_net)

        def one():
            th.set_grad_enabled(False)                        # the caller toggles grad mode exactly as run.py does (:41,124-127)
            items = agent.explore_env(env, H)
            th.set_grad_enabled(True)
            agent.update_net(list(items))
            th.set_grad_enabled(False)
    else:
        from oracle.torch_port import TorchPortPPO
        env = _port_env(env_kind, N, S, A, max_step)
        port = TorchPortPPO(S, A, tuple(net_dims), lr=hyper.get("learning_rate", 6e-5), gamma=hyper.get("gamma", 0.99),
                            lambda_entropy=hyper.get("lambda_entropy", 0.001), reward_scale=hyper.get("reward_scale", 1.0))
        port.last_state = env.reset()
        port.explore = timer.wrap("explore_env", port.explore)
        port.advantages = timer.wrap("get_advantages", port.advantages)
        port.update = timer.wrap("update_net", port.update)

        def one():
            buf = port.explore(env, H)
            port.update(list(buf), B, update_times)

    one()                                                     # warm-up (allocator, OpenMP pool)
    timer.reset()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    dt = time.perf_counter() - t0
    return {"value": round(N * H * iters / dt, 1), "unit": "env-steps/s", "cores": th.get_num_threads(), "kind": kind,
            "sample": f"{iters} PPO iterations (+1 warm-up) of the same workload ({N} envs x {H} steps, {update_times} minibatches of {B}, "
                      f"net {list(net_dims)}) via " + ("/root/reference elegantrl.agents.AgentPPO (gpu_id=-1)" if kind == "reference"
                                                      else "oracle/torch_port.py"),
            "seconds": round(dt, 2),
            "stage_seconds": {**timer.report(), "note": "get_advantages is inside update_net; per-stage sums over the sample"}}


# ---------------------------------------------------------------------------------------------------------
# SAC + replay ring (BASELINE configs[2])
# ---------------------------------------------------------------------------------------------------------
class _PortRing:
    """torch restatement of ReplayBuffer.update / sample (elegantrl/train/replay_buffer.py:78-134), float flags, ids % L / ids // L"""

    def __init__(self, max_size, num_seqs, S, A):
        self.max_size, self.num_seqs = max_size, num_seqs
        self.p, self.cur_size, self.if_full = 0, 0, False
        self.states = th.empty((max_size, num_seqs, S))
        self.actions = th.empty((max_size, num_seqs, A))
        self.rewards = th.empty((max_size, num_seqs))
        self.undones = th.empty((max_size, num_seqs))
        self.unmasks = th.empty((max_size, num_seqs))

    def update(self, items):
        add = items[2].shape[0]
        bufs = (self.states, self.actions, self.rewards, self.undones, self.unmasks)
        p = self.p + add
        if p > self.max_size:
            self.if_full = True
            p0, p1 = self.p, self.max_size
            p2 = self.max_size - self.p
            p = p - self.max_size
            for b, it in zip(bufs, items):
                b[p0:p1], b[0:p] = it[:p2], it[-p:]
        else:
            for b, it in zip(bufs, items):
                b[self.p:p] = it
        self.p = p
        self.cur_size = self.max_size if self.if_full else self.p

    def sample(self, batch_size):
        L = self.cur_size - 1
        ids = th.randint(L * self.num_seqs, size=(batch_size,), requires_grad=False)
        ids0, ids1 = th.fmod(ids, L), th.div(ids, L, rounding_mode="floor")
        return (self.states[ids0, ids1], self.actions[ids0, ids1], self.rewards[ids0, ids1], self.undones[ids0, ids1],
                self.unmasks[ids0, ids1], self.states[ids0 + 1, ids1])


def sac(kind: str, *, N, S, A, H, B, updates, net_dims, max_size, iters, max_step=1000) -> dict:
    th.manual_seed(0)
    timer = StageTimer()
    g = th.Generator().manual_seed(1)
    fill = lambda rows: (th.randn((rows, N, S), generator=g), th.randn((rows, N, A), generator=g).tanh(),  # noqa: E731
                         th.randn((rows, N), generator=g), (th.rand((rows, N), generator=g) < 0.99),
                         (th.rand((rows, N), generator=g) < 0.995))
    if kind == "reference":
        _import_reference()
        from elegantrl.agents import AgentSAC
        from elegantrl.train.config import Config
        from elegantrl.train.replay_buffer import ReplayBuffer
        args = Config(AgentSAC, None, {"env_name": "syn", "num_envs": N, "max_step": max_step, "state_dim": S, "action_dim": A,
                                       "if_discrete": False})
        args.net_dims, args.horizon_len, args.batch_size = list(net_dims), H, B
        args.repeat_times = updates * B / max_size            # AgentBase.update_net: int(cur_size * repeat_times / B) = updates
        agent = AgentSAC(args.net_dims, S, A, gpu_id=-1, args=args)
        env = RefProtocolEnv(N, S, A, max_step)
        agent.last_state = env.reset()[0]
        buf = ReplayBuffer(max_size=max_size, state_dim=S, action_dim=A, gpu_id=-1, num_seqs=N)
        for _ in range(2):                                    # fill the ring completely (and wrap once)
            buf.update(fill(max_size // 2 + 7))
        assert buf.if_full and buf.cur_size == max_size
        agent.explore_env = timer.wrap("explore_env", agent.explore_env)
        buf.update = timer.wrap("buffer_update", buf.update)
        buf.sample = timer.wrap("sample", buf.sample)
        agent.update_net = timer.wrap("update_net", agent.update_net)

        def one():
            th.set_grad_enabled(False)
            buf.update(agent.explore_env(env, H))
            th.set_grad_enabled(True)
            agent.update_net(buf)
            th.set_grad_enabled(False)
    else:
        from oracle.sac_torch import SacStepper
        from oracle.torch_port import TorchSynEnv
        env = TorchSynEnv(N, S, A, max_step, seed=0)
        st = SacStepper(list(net_dims), S, A, 4, 1e-4, 0.99, 5e-3, 3.0)     # AgentSAC defaults: 4 critics (AgentSAC.py:17), config.py:36-49
        ring = _PortRing(max_size, N, S, A)
        for _ in range(2):
            ring.update(tuple(x.float() for x in fill(max_size // 2 + 7)))
        assert ring.if_full and ring.cur_size == max_size
        state = [env.reset()]

        @th.no_grad()
        def explore():                                        # AgentBase._explore_vec_env (AgentBase.py:130-170)
            states, actions = th.zeros((H, N, S)), th.zeros((H, N, A))
            rewards, terminals, truncates = th.zeros((H, N)), th.zeros((H, N), dtype=th.bool), th.zeros((H, N), dtype=th.bool)
            s = state[0]
            for t in range(H):
                a = st.act.get_action(s)
                states[t], actions[t] = s, a
                s, rewards[t], terminals[t], truncates[t] = env.step(a)
            state[0] = s
            return states, actions, rewards, (~terminals).float(), (~truncates).float()

        def update():                                         # AgentBase.update_net (AgentBase.py:172-189)
            with th.enable_grad():
                for _ in range(updates):
                    batch = sample(B)
                    st.step(batch, None, None)

        explore = timer.wrap("explore_env", explore)
        ring.update = timer.wrap("buffer_update", ring.update)
        sample = timer.wrap("sample", ring.sample)
        update = timer.wrap("update_net", update)

        def one():
            ring.update(explore())
            update()

    one()
    timer.reset()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    dt = time.perf_counter() - t0
    return {"value": round(updates * iters / dt, 1), "unit": "updates/s", "cores": th.get_num_threads(), "kind": kind,
            "sample": f"{iters} off-policy iterations (+1 warm-up) of the same workload ({N} envs x {H} steps into a full ring of "
                      f"{max_size * N} transitions, {updates} x [sample({B}) + SAC update], net {list(net_dims)}) via "
                      + ("/root/reference elegantrl AgentSAC + ReplayBuffer (gpu_id=-1)" if kind == "reference"
                         else "oracle/sac_torch.py + a torch ring"),
            "seconds": round(dt, 2), "env_steps_per_sec": round(N * H * iters / dt, 1),
            "stage_seconds": {**timer.report(), "note": "sample is inside update_net; per-stage sums over the sample"}}

