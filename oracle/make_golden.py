"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (imported from /root/reference).

Run in the authoring container only (the reference does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The fixtures pin the oracle (oracle/ppo_numpy.py, oracle/gae_scan.c, oracle/torch_port.py) to the
reference's real outputs on seeded inputs; tests/test_oracle_golden.py re-checks them everywhere.
Nothing from the reference is copied: only its *outputs* on our inputs are stored.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch as th

REF = os.environ.get("ERL_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class ScriptedVecEnv:
    """Linear toy env whose terminal/truncate flags come from a seeded script (so both occur)."""

    def __init__(self, num_envs, state_dim, action_dim, seed):
        g = th.Generator().manual_seed(seed)
        self.num_envs, self.state_dim, self.action_dim = num_envs, state_dim, action_dim
        self.ws = 0.9 * th.eye(state_dim) + 0.05 * th.randn(state_dim, state_dim, generator=g)
        self.wa = 0.1 * th.randn(action_dim, state_dim, generator=g)
        self.g = g
        self.state = th.randn(num_envs, state_dim, generator=g)

    def reset(self):
        return self.state.clone(), {}

    def step(self, action):
        s = self.state @ self.ws + action @ self.wa
        reward = -(s * s).mean(1) - 0.01 * (action * action).mean(1)
        u = th.rand(self.num_envs, generator=self.g)
        terminal = u < 0.10
        truncate = (u >= 0.10) & (u < 0.22)
        done = terminal | truncate
        fresh = th.randn(self.num_envs, self.state_dim, generator=self.g)
        s = th.where(done[:, None], fresh, s)
        self.state = s
        return s.clone(), reward, terminal, truncate, {}


def np32(t):
    return t.detach().cpu().numpy().copy()


def net_arrays(prefix, net):
    out = {}
    for k, v in net.state_dict().items():
        out[f"{prefix}.{k}"] = np32(v)
    return out


def make_ppo(tag, *, N, S, A, H, net_dims, batch_size, repeat_times, use_v_trace, seed):
    sys.path.insert(0, REF)
    from elegantrl.agents import AgentPPO
    from elegantrl.train.config import Config

    th.manual_seed(seed)
    args = Config(AgentPPO, None, {"env_name": "scripted", "num_envs": N, "max_step": 100,
                                   "state_dim": S, "action_dim": A, "if_discrete": False})
    args.net_dims = list(net_dims)
    args.horizon_len, args.batch_size, args.repeat_times = H, batch_size, repeat_times
    args.learning_rate = 1e-3
    args.gamma = 0.99
    args.reward_scale = 0.5
    args.if_use_v_trace = use_v_trace
    agent = AgentPPO(args.net_dims, S, A, gpu_id=-1, args=args)
    with th.no_grad():  # non-trivial normalisation buffers and action std
        for net in (agent.act, agent.cri):
            net.state_avg[:] = 0.1 * th.randn(S)
            net.state_std[:] = 1.0 + 0.2 * th.rand(S)
        agent.act.action_std_log[:] = -0.3 + 0.1 * th.randn(1, A)

    env = ScriptedVecEnv(N, S, A, seed + 1)
    agent.last_state = env.reset()[0]
    first_state = np32(agent.last_state)

    th.set_grad_enabled(False)
    g = {}
    g.update(net_arrays("act0", agent.act))
    g.update(net_arrays("cri0", agent.cri))

    # --- rollout: record the policy mean so eps can be recovered ---
    means = []
    orig_get_action = agent.act.get_action

    def recording_get_action(state):
        means.append(agent.act.net(agent.act.state_norm(state)).clone())
        return orig_get_action(state)

    agent.act.get_action = recording_get_action
    items = agent.explore_env(env, H)
    agent.act.get_action = orig_get_action
    states, actions, logprobs, rewards, undones, unmasks = items
    mean = th.stack(means)
    std = agent.act.action_std_log.exp()
    eps = ((actions.double() - mean.double()) / std.double())
    g.update(first_state=first_state, states=np32(states), actions=np32(actions), logprobs=np32(logprobs),
             rewards=np32(rewards), undones=np32(undones), unmasks=np32(unmasks),
             eps=eps.numpy().copy(), action_mean=np32(mean), last_state=np32(agent.last_state))

    # --- value pre-pass + GAE on clones (get_advantages mutates rewards/undones) ---
    values = agent.cri(states).squeeze(-1)
    r2, u2 = rewards.clone(), undones.clone()
    adv = agent.get_advantages(states, r2, u2, unmasks, values)
    next_value = agent.cri(agent.last_state).squeeze(-1)
    adv_norm = (adv - adv.mean()) / (adv[::4, ::4].std() + 1e-5)
    g.update(values=np32(values), next_value=np32(next_value), advantages=np32(adv),
             rewards_after=np32(r2), undones_after=np32(u2), reward_sums=np32(adv + values),
             advantages_norm=np32(adv_norm))

    # --- update_net with recorded minibatch ids ---
    ids_log = []
    orig_randint = th.randint

    def recording_randint(*a, **k):
        out = orig_randint(*a, **k)
        ids_log.append(out.clone())
        return out

    th.randint = recording_randint
    th.set_grad_enabled(True)
    objs = agent.update_net([t.clone() for t in items])
    th.set_grad_enabled(False)
    th.randint = orig_randint
    g.update(net_arrays("act1", agent.act))
    g.update(net_arrays("cri1", agent.cri))
    g.update(ids=np.stack([np32(i) for i in ids_log]).astype(np.int64),
             objs=np.array([float(o) for o in objs], dtype=np.float64),
             hyper=np.array([args.gamma, agent.lambda_gae_adv, agent.ratio_clip, float(agent.lambda_entropy),
                             args.learning_rate, args.clip_grad_norm, args.reward_scale], dtype=np.float64),
             dims=np.array([N, S, A, H, batch_size, len(ids_log), int(use_v_trace), *net_dims], dtype=np.int64))
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, f"ppo_{tag}.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, {k: v.shape for k, v in g.items() if k in ("states", "ids", "advantages")})


class ScriptedDiscreteVecEnv(ScriptedVecEnv):
    """the same toy dynamics driven by a discrete action (one row of wa per action index, int64 as the reference's
    convert_action_for_env produces)."""

    def step(self, action):
        assert action.dtype == th.int64 and action.shape == (self.num_envs,)
        s = self.state @ self.ws + self.wa[action]
        reward = -(s * s).mean(1)
        u = th.rand(self.num_envs, generator=self.g)
        terminal = u < 0.10
        truncate = (u >= 0.10) & (u < 0.22)
        fresh = th.randn(self.num_envs, self.state_dim, generator=self.g)
        s = th.where((terminal | truncate)[:, None], fresh, s)
        self.state = s
        return s.clone(), reward, terminal, truncate, {}


def make_ppo_discrete(tag, *, N, S, A, H, net_dims, batch_size, repeat_times, seed):
    """reference AgentDiscretePPO: rollout (actions int32, log-probs, logits recorded) and one update_net with recorded ids."""
    sys.path.insert(0, REF)
    from elegantrl.agents import AgentDiscretePPO
    from elegantrl.train.config import Config

    th.manual_seed(seed)
    args = Config(AgentDiscretePPO, None, {"env_name": "scripted", "num_envs": N, "max_step": 100, "state_dim": S, "action_dim": A,
                                           "if_discrete": True})
    args.net_dims = list(net_dims)
    args.horizon_len, args.batch_size, args.repeat_times = H, batch_size, repeat_times
    args.learning_rate, args.gamma, args.reward_scale = 1e-3, 0.99, 0.5
    agent = AgentDiscretePPO(args.net_dims, S, A, gpu_id=-1, args=args)
    with th.no_grad():
        for net in (agent.act, agent.cri):
            net.state_avg[:] = 0.1 * th.randn(S)
            net.state_std[:] = 1.0 + 0.2 * th.rand(S)
        agent.act.net[-1].weight *= 8.0           # spread the logits so the policy is far from uniform
    env = ScriptedDiscreteVecEnv(N, S, A, seed + 1)
    agent.last_state = env.reset()[0]
    th.set_grad_enabled(False)
    g = {}
    g.update(net_arrays("act0", agent.act))
    g.update(net_arrays("cri0", agent.cri))
    g.update(first_state=np32(agent.last_state))
    logits = []
    orig_get_action = agent.act.get_action

    def recording_get_action(state):
        logits.append(agent.act.net(agent.act.state_norm(state)).clone())
        return orig_get_action(state)

    agent.act.get_action = recording_get_action
    items = agent.explore_env(env, H)
    agent.act.get_action = orig_get_action
    states, actions, logprobs, rewards, undones, unmasks = items
    assert actions.dtype == th.int32 and actions.shape == (H, N)
    g.update(states=np32(states), actions=np32(actions), logprobs=np32(logprobs), rewards=np32(rewards), undones=np32(undones),
             unmasks=np32(unmasks), logits=np32(th.stack(logits)), last_state=np32(agent.last_state))
    values = agent.cri(states).squeeze(-1)
    r2, u2 = rewards.clone(), undones.clone()
    adv = agent.get_advantages(states, r2, u2, unmasks, values)
    g.update(values=np32(values), advantages=np32(adv), reward_sums=np32(adv + values),
             advantages_norm=np32((adv - adv.mean()) / (adv[::4, ::4].std() + 1e-5)))
    ids_log = []
    orig_randint = th.randint

    def recording_randint(*a, **k):
        out = orig_randint(*a, **k)
        ids_log.append(out.clone())
        return out

    th.randint = recording_randint
    th.set_grad_enabled(True)
    objs = agent.update_net([t.clone() for t in items])
    th.set_grad_enabled(False)
    th.randint = orig_randint
    g.update(net_arrays("act1", agent.act))
    g.update(net_arrays("cri1", agent.cri))
    g.update(ids=np.stack([np32(i) for i in ids_log]).astype(np.int64), objs=np.array([float(o) for o in objs], dtype=np.float64),
             hyper=np.array([args.gamma, agent.lambda_gae_adv, agent.ratio_clip, float(agent.lambda_entropy), args.learning_rate,
                             args.clip_grad_norm, args.reward_scale], dtype=np.float64),
             dims=np.array([N, S, A, H, batch_size, len(ids_log), 1, *net_dims], dtype=np.int64))
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, f"ppo_discrete_{tag}.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, {k: v.shape for k, v in g.items() if k in ("states", "ids", "actions")}, "objs", objs)


def make_replay():
    sys.path.insert(0, REF)
    from elegantrl.train.replay_buffer import ReplayBuffer

    th.manual_seed(7)
    max_size, S, A, num_seqs = 20, 3, 2, 2
    buf = ReplayBuffer(max_size=max_size, state_dim=S, action_dim=A, gpu_id=-1, num_seqs=num_seqs)
    buf.states.zero_(); buf.actions.zero_(); buf.rewards.zero_(); buf.undones.zero_(); buf.unmasks.zero_()
    adds = [7, 7, 6, 5, 9, 20, 3]   # 7+7+6 lands exactly on max_size (edge A12); 20 == max_size
    g = {"adds": np.array(adds), "dims": np.array([max_size, S, A, num_seqs])}
    orig_randint = th.randint
    for k, add in enumerate(adds):
        items = (th.randn(add, num_seqs, S), th.randn(add, num_seqs, A), th.randn(add, num_seqs),
                 th.rand(add, num_seqs) > 0.2, th.rand(add, num_seqs) > 0.1)
        buf.update(items)
        for name, t in zip(("states", "actions", "rewards", "undones", "unmasks"), items):
            g[f"in{k}_{name}"] = np32(t)
        g[f"cursor{k}"] = np.array([buf.p, buf.cur_size, int(buf.if_full), buf.add_size])
        for name in ("states", "actions", "rewards", "undones", "unmasks"):
            g[f"buf{k}_{name}"] = np32(getattr(buf, name))
        log = []

        def rec(*a, **kw):
            out = orig_randint(*a, **kw)
            log.append(out.clone())
            return out

        th.randint = rec
        out = buf.sample(16)
        th.randint = orig_randint
        g[f"ids{k}"] = np32(log[0]).astype(np.int64)
        g[f"ids0_{k}"] = np32(buf.ids0).astype(np.int64)
        g[f"ids1_{k}"] = np32(buf.ids1).astype(np.int64)
        for name, t in zip(("state", "action", "reward", "undone", "unmask", "next_state"), out):
            g[f"out{k}_{name}"] = np32(t)
    path = os.path.join(OUT, "replay_ring.npz")
    np.savez_compressed(path, **g)
    print("wrote", path)


def make_replay_discrete():
    """reference ReplayBuffer(if_discrete=True): uint8 action ring fed with (add, num_seqs) int32 actions (AgentBase.py:146),
    same cursor schedule as make_replay (wrap-around, landing exactly on max_size)."""
    sys.path.insert(0, REF)
    from elegantrl.train.replay_buffer import ReplayBuffer

    th.manual_seed(8)
    max_size, S, n_actions, num_seqs = 20, 3, 6, 2
    buf = ReplayBuffer(max_size=max_size, state_dim=S, action_dim=1, gpu_id=-1, num_seqs=num_seqs, if_discrete=True)
    assert buf.actions.dtype == th.uint8 and buf.actions.shape == (max_size, num_seqs)
    buf.states.zero_(); buf.actions.zero_(); buf.rewards.zero_(); buf.undones.zero_(); buf.unmasks.zero_()
    adds = [7, 7, 6, 5, 9, 20, 3]
    g = {"adds": np.array(adds), "dims": np.array([max_size, S, n_actions, num_seqs])}
    orig_randint = th.randint
    for k, add in enumerate(adds):
        items = (th.randn(add, num_seqs, S), orig_randint(n_actions, (add, num_seqs), dtype=th.int32), th.randn(add, num_seqs),
                 th.rand(add, num_seqs) > 0.2, th.rand(add, num_seqs) > 0.1)
        buf.update(items)
        for name, t in zip(("states", "actions", "rewards", "undones", "unmasks"), items):
            g[f"in{k}_{name}"] = np32(t)
        g[f"cursor{k}"] = np.array([buf.p, buf.cur_size, int(buf.if_full), buf.add_size])
        for name in ("states", "actions", "rewards", "undones", "unmasks"):
            g[f"buf{k}_{name}"] = np32(getattr(buf, name))
        log = []

        def rec(*a, **kw):
            out = orig_randint(*a, **kw)
            log.append(out.clone())
            return out

        th.randint = rec
        out = buf.sample(16)
        th.randint = orig_randint
        g[f"ids{k}"] = np32(log[0]).astype(np.int64)
        g[f"ids0_{k}"] = np32(buf.ids0).astype(np.int64)
        g[f"ids1_{k}"] = np32(buf.ids1).astype(np.int64)
        for name, t in zip(("state", "action", "reward", "undone", "unmask", "next_state"), out):
            g[f"out{k}_{name}"] = np32(t)
        assert out[1].dtype == th.uint8 and out[1].shape == (16,)
    path = os.path.join(OUT, "replay_ring_discrete.npz")
    np.savez_compressed(path, **g)
    print("wrote", path)


def make_sac(tag, *, N, S, A, rows, net_dims, batch_size, n_updates, seed, lambda_fit=0.0):
    """Run the reference's AgentSAC.update_objectives on a seeded ring; record every random draw it makes
    (minibatch ids via th.randint, the two rsample() noise tensors per step via th.distributions.Normal.rsample) so the
    same step can be replayed elsewhere with injected draws.  Also records the off-policy rollout contract
    (AgentBase._explore_vec_env): stored action == action sent to the env == tanh(mean + std * eps)."""
    sys.path.insert(0, REF)
    from elegantrl.agents import AgentSAC
    from elegantrl.train.config import Config
    from elegantrl.train.replay_buffer import ReplayBuffer

    th.manual_seed(seed)
    args = Config(AgentSAC, None, {"env_name": "scripted", "num_envs": N, "max_step": 100,
                                   "state_dim": S, "action_dim": A, "if_discrete": False})
    args.net_dims = list(net_dims)
    args.batch_size, args.learning_rate, args.gamma, args.reward_scale = batch_size, 1e-3, 0.98, 0.5
    args.soft_update_tau = 5e-3
    agent = AgentSAC(args.net_dims, S, A, gpu_id=-1, args=args)
    g = {}
    g.update(net_arrays("act0", agent.act))
    g.update(net_arrays("cri0", agent.cri))
    g["alpha_log0"] = np32(agent.alpha_log)

    # --- record rsample draws ---
    Normal = th.distributions.normal.Normal
    orig_rsample = Normal.rsample
    eps_log = []

    def rec_rsample(self, sample_shape=th.Size()):
        eps = th.randn(self.loc.shape)
        eps_log.append(eps.clone())
        return self.loc + eps * self.scale

    Normal.rsample = rec_rsample

    # --- off-policy rollout through the reference's loop ---
    env = ScriptedVecEnv(N, S, A, seed + 1)
    agent.last_state = env.reset()[0]
    g["first_state"] = np32(agent.last_state)
    th.set_grad_enabled(False)
    items = agent.explore_env(env, rows)
    states, actions, rewards, undones, unmasks = items
    g.update(ro_states=np32(states), ro_actions=np32(actions), ro_rewards=np32(rewards), ro_undones=np32(undones),
             ro_unmasks=np32(unmasks), ro_eps=np.stack([np32(e) for e in eps_log]), ro_last_state=np32(agent.last_state))
    eps_log.clear()

    buf = ReplayBuffer(max_size=rows + 5, state_dim=S, action_dim=A, gpu_id=-1, num_seqs=N)
    buf.update(items)
    if lambda_fit:
        # the critic's `lambda_fit_cum_r` term (AgentSAC.py:66-68) reads buffer.cum_rewards[buffer.ids0, buffer.ids1].  The
        # reference fills that array through AgentBase.get_cumulative_rewards, which for AgentSAC calls `self.act_target`
        # = None (AgentBase.py:55, AgentSAC.py:24) and raises; update_objectives itself runs on any contents, so the array is
        # seeded here and recorded
        agent.lambda_fit_cum_r = float(lambda_fit)
        buf.cum_rewards[:] = th.randn(buf.cum_rewards.shape) * 2.0 + 1.0
        g["cum_rewards"] = np32(buf.cum_rewards)
        g["lambda_fit_cum_r"] = np.array([lambda_fit], dtype=np.float64)

    # --- n_updates SAC steps with recorded ids / noise ---
    ids_log = []
    orig_randint = th.randint

    def rec_randint(*a, **k):
        out = orig_randint(*a, **k)
        ids_log.append(out.clone())
        return out

    th.randint = rec_randint
    th.set_grad_enabled(True)
    objs = []
    for t in range(n_updates):
        objs.append(agent.update_objectives(buf, t))
        g.update(net_arrays(f"act{t + 1}", agent.act))
        g.update(net_arrays(f"cri{t + 1}", agent.cri))
        g.update(net_arrays(f"crit{t + 1}", agent.cri_target))
        g[f"alpha_log{t + 1}"] = np32(agent.alpha_log)
    th.set_grad_enabled(False)
    th.randint = orig_randint
    Normal.rsample = orig_rsample
    assert len(eps_log) == 2 * n_updates and len(ids_log) == n_updates
    g.update(ids=np.stack([np32(i) for i in ids_log]).astype(np.int64), eps_next=np.stack([np32(e) for e in eps_log[0::2]]),
             eps_cur=np.stack([np32(e) for e in eps_log[1::2]]), objs=np.array(objs, dtype=np.float64),
             hyper=np.array([args.gamma, args.learning_rate, args.clip_grad_norm, args.reward_scale, args.soft_update_tau,
                             agent.target_entropy], dtype=np.float64),
             dims=np.array([N, S, A, rows, batch_size, n_updates, agent.num_ensembles, *net_dims], dtype=np.int64))
    path = os.path.join(OUT, f"sac_{tag}.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, "objs", objs)


def make_sac_mod(tag, *, N, S, A, rows, net_dims, batch_size, n_updates, seed):
    """The reference's AgentModSAC (elegantrl/agents/AgentSAC.py:89-165) with its ActorFixSAC (:201-243): run update_objectives on a
    seeded ring and record every random draw (minibatch ids via th.randint; the rollout's and the two per-step noise tensors via
    th.randn_like, which is what ActorFixSAC draws from) so that the steps can be replayed with injected draws.  The two-time-scale
    rule (:148-158) skips the actor on some steps (obj_actor = nan there): the fixture covers updated and skipped steps."""
    sys.path.insert(0, REF)
    from elegantrl.agents.AgentSAC import AgentModSAC
    from elegantrl.train.config import Config
    from elegantrl.train.replay_buffer import ReplayBuffer

    th.manual_seed(seed)
    args = Config(AgentModSAC, None, {"env_name": "scripted", "num_envs": N, "max_step": 100,
                                      "state_dim": S, "action_dim": A, "if_discrete": False})
    assert args.if_off_policy
    args.net_dims = list(net_dims)
    args.batch_size, args.learning_rate, args.gamma, args.reward_scale = batch_size, 1e-3, 0.98, 0.5
    args.soft_update_tau = 5e-3
    agent = AgentModSAC(args.net_dims, S, A, gpu_id=-1, args=args)
    g = {}
    g.update(net_arrays("act0", agent.act))
    g.update(net_arrays("cri0", agent.cri))
    g["alpha_log0"] = np32(agent.alpha_log)

    orig_randn_like = th.randn_like
    eps_log = []

    def rec_randn_like(t, **k):
        eps = orig_randn_like(t, **k)
        eps_log.append(eps.detach().clone())
        return eps

    th.randn_like = rec_randn_like
    env = ScriptedVecEnv(N, S, A, seed + 1)
    agent.last_state = env.reset()[0]
    g["first_state"] = np32(agent.last_state)
    th.set_grad_enabled(False)
    items = agent.explore_env(env, rows)
    states, actions, rewards, undones, unmasks = items
    assert len(eps_log) == rows
    g.update(ro_states=np32(states), ro_actions=np32(actions), ro_rewards=np32(rewards), ro_undones=np32(undones),
             ro_unmasks=np32(unmasks), ro_eps=np.stack([np32(e) for e in eps_log]), ro_last_state=np32(agent.last_state))
    eps_log.clear()
    buf = ReplayBuffer(max_size=rows + 5, state_dim=S, action_dim=A, gpu_id=-1, num_seqs=N)
    buf.update(items)

    ids_log = []
    orig_randint = th.randint

    def rec_randint(*a, **k):
        out = orig_randint(*a, **k)
        ids_log.append(out.clone())
        return out

    th.randint = rec_randint
    th.set_grad_enabled(True)
    objs, upd = [], []
    for t in range(n_updates):
        a_before = agent.update_a if t else 0
        objs.append(agent.update_objectives(buf, t))
        upd.append(int(agent.update_a != a_before))
        g.update(net_arrays(f"act{t + 1}", agent.act))
        g.update(net_arrays(f"actt{t + 1}", agent.act_target))
        g.update(net_arrays(f"cri{t + 1}", agent.cri))
        g.update(net_arrays(f"crit{t + 1}", agent.cri_target))
        g[f"alpha_log{t + 1}"] = np32(agent.alpha_log)
    th.set_grad_enabled(False)
    th.randint = orig_randint
    th.randn_like = orig_randn_like
    assert len(eps_log) == 2 * n_updates and len(ids_log) == n_updates
    assert 0 < sum(upd) < n_updates, upd                 # both kinds of step are in the fixture
    assert all(np.isnan(o[1]) != bool(u) for o, u in zip(objs, upd))
    g.update(ids=np.stack([np32(i) for i in ids_log]).astype(np.int64), eps_next=np.stack([np32(e) for e in eps_log[0::2]]),
             eps_cur=np.stack([np32(e) for e in eps_log[1::2]]), objs=np.array(objs, dtype=np.float64),
             actor_updated=np.array(upd, dtype=np.int64),
             hyper=np.array([args.gamma, args.learning_rate, args.clip_grad_norm, args.reward_scale, args.soft_update_tau,
                             agent.target_entropy, agent.critic_tau, agent.critic_value], dtype=np.float64),
             dims=np.array([N, S, A, rows, batch_size, n_updates, agent.num_ensembles, *net_dims], dtype=np.int64))
    path = os.path.join(OUT, f"sac_mod_{tag}.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, "objs", objs, "actor updated", upd)


def make_state_norm():
    """AgentPPO.update_avg_std_for_normalization (elegantrl/agents/AgentPPO.py:234-249) on seeded states, twice.  The reference's method
    goes on to `self.act_target.state_avg[:] = ...` (:246), and AgentPPO has no act_target (AgentBase.py:55: None): it raises AFTER the
    actor's and the critic's vectors are written.  What it leaves in them is recorded (and that it raised)."""
    sys.path.insert(0, REF)
    from elegantrl.agents import AgentPPO
    from elegantrl.train.config import Config
    th.manual_seed(51)
    S, A = 6, 2
    args = Config(AgentPPO, None, {"env_name": "scripted", "num_envs": 8, "max_step": 100, "state_dim": S, "action_dim": A,
                                   "if_discrete": False})
    args.net_dims = [64, 32]
    args.state_value_tau = 0.1
    agent = AgentPPO(args.net_dims, S, A, gpu_id=-1, args=args)
    g = {"tau": np.array([args.state_value_tau]), "dims": np.array([S, A], dtype=np.int64)}
    th.set_grad_enabled(False)
    raised = []
    for k in range(2):
        states = th.randn(50, S) * (1.0 + k) + 0.5 * k
        states[:, 3] = 2.0                                    # a constant feature: std 0 -> the clamp_min(1e-4) branch after enough calls
        g[f"states{k}"] = np32(states)
        try:
            agent.update_avg_std_for_normalization(states)
            raised.append(0)
        except AttributeError:
            raised.append(1)
        g[f"act_avg{k}"], g[f"act_std{k}"] = np32(agent.act.state_avg), np32(agent.act.state_std)
        g[f"cri_avg{k}"], g[f"cri_std{k}"] = np32(agent.cri.state_avg), np32(agent.cri.state_std)
    g["raised"] = np.array(raised, dtype=np.int64)
    path = os.path.join(OUT, "state_norm.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, "reference raised AttributeError:", raised)


def make_a2c(tag, *, S, A, H, net_dims, batch_size, repeat_times, seed):
    """reference AgentA2C (elegantrl/agents/AgentPPO.py:252-303) on a one-env buffer (the only shape its time-row minibatches
    are well formed for): update_net with recorded time indices; weights before / after and the returned objectives."""
    sys.path.insert(0, REF)
    from elegantrl.agents.AgentPPO import AgentA2C
    from elegantrl.train.config import Config

    th.manual_seed(seed)
    args = Config(AgentA2C, None, {"env_name": "scripted", "num_envs": 1, "max_step": 100, "state_dim": S, "action_dim": A,
                                   "if_discrete": False})
    args.net_dims = list(net_dims)
    args.horizon_len, args.batch_size, args.repeat_times = H, batch_size, repeat_times
    args.learning_rate, args.gamma, args.reward_scale = 1e-3, 0.98, 1.0
    agent = AgentA2C(args.net_dims, S, A, gpu_id=-1, args=args)
    with th.no_grad():
        for net in (agent.act, agent.cri):
            net.state_avg[:] = 0.1 * th.randn(S)
            net.state_std[:] = 1.0 + 0.2 * th.rand(S)
        agent.act.action_std_log[:] = -0.3 + 0.1 * th.randn(1, A)
    g = {}
    g.update(net_arrays("act0", agent.act))
    g.update(net_arrays("cri0", agent.cri))
    states = th.randn(H, 1, S)
    actions = th.randn(H, 1, A)
    logprobs = -1.0 * A + 0.3 * th.randn(H, 1)
    rewards = th.randn(H, 1)
    undones = th.rand(H, 1) > 0.1
    unmasks = th.rand(H, 1) > 0.12
    undones = undones & unmasks | (~unmasks & False)       # a truncated step is an episode end as well
    agent.last_state = th.randn(1, S)
    g.update(states=np32(states), actions=np32(actions), logprobs=np32(logprobs), rewards=np32(rewards), undones=np32(undones),
             unmasks=np32(unmasks), last_state=np32(agent.last_state))
    th.set_grad_enabled(False)
    values = agent.cri(states).squeeze(-1)
    r2, u2 = rewards.clone(), undones.clone()
    adv = agent.get_advantages(states, r2, u2, unmasks, values)
    g.update(values=np32(values), advantages=np32(adv), reward_sums=np32(adv + values),
             advantages_norm=np32((adv - adv.mean()) / (adv[::4, ::4].std() + 1e-5)))
    ids_log = []
    orig_randint = th.randint

    def recording_randint(*a, **k):
        out = orig_randint(*a, **k)
        ids_log.append(out.clone())
        return out

    th.randint = recording_randint
    objs = agent.update_net((states.clone(), actions.clone(), logprobs.clone(), rewards.clone(), undones.clone(), unmasks.clone()))
    th.set_grad_enabled(False)
    th.randint = orig_randint
    g.update(net_arrays("act1", agent.act))
    g.update(net_arrays("cri1", agent.cri))
    g.update(ids=np.stack([np32(i) for i in ids_log]).astype(np.int64), objs=np.array([float(o) for o in objs], dtype=np.float64),
             hyper=np.array([args.gamma, agent.lambda_gae_adv, args.learning_rate, args.clip_grad_norm], dtype=np.float64),
             dims=np.array([1, S, A, H, batch_size, len(ids_log), *net_dims], dtype=np.int64))
    path = os.path.join(OUT, f"a2c_{tag}.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, {k: v.shape for k, v in g.items() if k in ("states", "ids", "advantages")}, "objs", objs)


def make_cum_rewards():
    """reference AgentBase.get_cumulative_rewards (elegantrl/agents/AgentBase.py:226-237) through AgentTD3 (which owns
    act_target / cri_target), called the way AgentBase.update_net does: via ReplayBuffer.update_cum_rewards (both the
    contiguous and the `p < add_size` branch, replay_buffer.py:213-223)."""
    sys.path.insert(0, REF)
    from elegantrl.agents.AgentTD3 import AgentTD3
    from elegantrl.train.config import Config
    from elegantrl.train.replay_buffer import ReplayBuffer

    th.manual_seed(41)
    N, S, A, max_size = 6, 5, 2, 24
    args = Config(AgentTD3, None, {"env_name": "x", "num_envs": N, "max_step": 50, "state_dim": S, "action_dim": A,
                                   "if_discrete": False})
    args.gamma = 0.985
    agent = AgentTD3((32, 32), S, A, gpu_id=-1, args=args)
    # As written the reference function cannot run with its own critics: they return (N, 1), and `rewards[t] + masks[t] *
    # next_value` then broadcasts to (N, N) and the assignment raises (executed here: RuntimeError expand [6, 6] -> [6]).
    # The fixture is generated with the critic's output squeezed to (N,) -- the evident intent -- and nothing else changed.
    twin = agent.cri_target
    agent.cri_target = lambda s, a: twin(s, a).squeeze(-1)
    buf = ReplayBuffer(max_size=max_size, state_dim=S, action_dim=A, gpu_id=-1, num_seqs=N)
    for t in (buf.states, buf.actions, buf.rewards, buf.undones, buf.unmasks, buf.cum_rewards):
        t.zero_()
    g = {"dims": np.array([N, S, A, max_size]), "gamma": np.array([args.gamma])}
    adds = [10, 9, 11, 7]          # 10 + 9 = 19; +11 wraps to p = 6 (< add_size: the p1 = max_size branch); +7 -> p = 13
    g["adds"] = np.array(adds)
    with th.no_grad():
        for k, add in enumerate(adds):
            items = (th.randn(add, N, S), th.randn(add, N, A).tanh(), th.randn(add, N), th.rand(add, N) > 0.15,
                     th.rand(add, N) > 0.1)
            buf.update(items)
            agent.last_state = th.randn(N, S)
            nv = agent.cri_target(agent.last_state, agent.act_target(agent.last_state))
            buf.update_cum_rewards(get_cumulative_rewards=agent.get_cumulative_rewards)
            g[f"rewards{k}"], g[f"undones{k}"] = np32(buf.rewards), np32(buf.undones)
            g[f"next_value{k}"] = np32(nv).reshape(-1)
            g[f"cursor{k}"] = np.array([buf.p, buf.cur_size, int(buf.if_full), buf.add_size])
            g[f"cum_rewards{k}"] = np32(buf.cum_rewards)
            # the function on its own, on the slice update_cum_rewards hands it
            p1 = buf.p if buf.p >= buf.add_size else buf.max_size
            p0 = p1 - buf.add_size
            g[f"slice{k}"] = np.array([p0, p1])
            g[f"direct{k}"] = np32(agent.get_cumulative_rewards(rewards=buf.rewards[p0:p1], undones=buf.undones[p0:p1]))
    path = os.path.join(OUT, "cum_rewards.npz")
    np.savez_compressed(path, **g)
    print("wrote", path)


def make_per_update():
    """Row f2, the half of the reference's prioritised replay that DOES run: SumTree.update_ids (elegantrl/train/replay_buffer.py:249-258),
    driven the way the reference drives it -- ReplayBuffer(if_use_per=True).update appends rows at priority 10 (:107-115), then
    td_error_update_for_per's formula prob = td_error.clamp(1e-8, 10).pow(per_alpha) (:168) goes into update_ids for every sequence's
    tree.  Only power-of-two buffer lengths: there the reference's 0-based heap (root 0, leaves at buf_len - 1 + row) is the heap of
    csrc/per.hip shifted by one node (root 1, leaves at L + row).  The reference's propagation loop runs `depth - 2` times (:254), i.e. it
    recomputes the tree levels 1 .. depth - 2 below the root and never the root itself; the fixture stores its whole tensor after every
    step and the tests compare exactly the levels it touched (`levels_updated`).  important_sampling asserts (tests/test_per.py) and is
    not part of the fixture."""
    sys.path.insert(0, REF)
    from elegantrl.train.config import Config
    from elegantrl.train.replay_buffer import ReplayBuffer

    th.manual_seed(51)
    g = {}
    # (no append wraps: the reference's PER branch raises on a wrapping append -- th.arange(self.p, p) with p already reduced, :109 --
    # recorded by tests/test_per.py::test_reference_per_append_raises_on_wrap)
    cases = [(8, 2, [3, 4]), (1024, 3, [400, 500, 124]), (4096, 1, [4000, 96])]
    g["cases"] = np.array([(m, q, len(a)) for m, q, a in cases], dtype=np.int64)
    for ci, (max_size, Q, adds) in enumerate(cases):
        args = Config()
        args.per_alpha, args.per_beta = 0.6, 0.4
        buf = ReplayBuffer(max_size=max_size, state_dim=3, action_dim=2, gpu_id=-1, num_seqs=Q, if_use_per=True, args=args)
        depth = buf.sum_trees[0].depth
        g[f"c{ci}_depth"] = np.array([depth, depth - 2])          # levels_updated = depth - 2 (counted up from the leaves' parents)
        for k, add in enumerate(adds):
            p0 = buf.p
            items = (th.randn(add, Q, 3), th.randn(add, Q, 2), th.randn(add, Q), th.rand(add, Q) > 0.1, th.rand(add, Q) > 0.1)
            buf.update(items)
            g[f"c{ci}_s{k}_append"] = np.array([p0, add, buf.p, buf.cur_size, int(buf.if_full)], dtype=np.int64)
            g[f"c{ci}_s{k}_tree_after_append"] = np.stack([np32(t.tree) for t in buf.sum_trees])
            # a td-error update on DISTINCT rows of every sequence (repeats are resolved by a rule of ours, oracle/per_numpy.py D7)
            n = min(buf.cur_size, 64 if max_size > 8 else 3)
            ids0 = th.stack([th.randperm(buf.cur_size)[:n] for _ in range(Q)])            # (Q, n) rows
            td = th.rand(Q, n) * 12.0
            td[:, 0] = 0.0                                                                  # the clamp's lower edge
            prob = td.clamp(1e-8, 10).pow(buf.per_alpha)                                     # replay_buffer.py:168
            for q in range(Q):
                buf.sum_trees[q].update_ids(ids0[q], prob[q])
            g[f"c{ci}_s{k}_ids0"], g[f"c{ci}_s{k}_td"], g[f"c{ci}_s{k}_prob"] = ids0.numpy(), np32(td), np32(prob)
            g[f"c{ci}_s{k}_tree_after_td"] = np.stack([np32(t.tree) for t in buf.sum_trees])
    path = os.path.join(OUT, "per_update.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, {k: v.shape for k, v in g.items() if "tree_after_td" in k})


def make_evaluator():
    """the reference's Evaluator (elegantrl/train/evaluator.py:12-155) driven with a toy single env and a toy vectorised env
    (tests/helpers.py) through a fixed schedule of evaluate_and_save calls: the files it leaves in cwd and recorder.npy."""
    import tempfile
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from elegantrl.train.config import Config
    from elegantrl.train.evaluator import Evaluator
    from tests.helpers import EVAL_SCHEDULE, ToyActor, ToySingleEnv, ToyVecEnv
    g = {}
    for tag, env, over_write in (("single", ToySingleEnv(), False), ("vec", ToyVecEnv(6), False), ("vec_overwrite", ToyVecEnv(6), True)):
        args = Config()
        args.gpu_id, args.eval_times, args.eval_per_step, args.eval_record_step = 0, 4 if tag == "single" else 12, 150, 0
        args.save_gap, args.if_keep_save, args.if_over_write = 2, True, over_write
        actor = ToyActor.build(env.state_dim, env.action_dim)
        with tempfile.TemporaryDirectory() as cwd, th.no_grad():
            ev = Evaluator(cwd=cwd, env=env, args=args)
            for steps, exp_r, log in EVAL_SCHEDULE:
                ev.evaluate_and_save(actor, steps, exp_r, log)
            ev.save_or_load_recoder(if_save=True)
            g[f"{tag}_files"] = np.array(sorted(os.listdir(cwd)))
            g[f"{tag}_recorder"] = np.load(f"{cwd}/recorder.npy")
        print(tag, list(g[f"{tag}_files"]), g[f"{tag}_recorder"].shape)
    path = os.path.join(OUT, "evaluator_format.npz")
    np.savez_compressed(path, **g)
    print("wrote", path)


if __name__ == "__main__":
    assert os.path.isdir(REF), f"reference not mounted at {REF}"
    only = sys.argv[1:]
    if only:                       # regenerate selected fixtures only: python oracle/make_golden.py cum_rewards ...
        for name in only:
            if name == "sac_fit_cum_r":
                make_sac("fit_cum_r", N=4, S=11, A=3, rows=40, net_dims=(64, 32), batch_size=64, n_updates=3, seed=22, lambda_fit=0.3)
            elif name == "ppo_c4shape":
                make_ppo("c4shape", N=64, S=64, A=8, H=16, net_dims=(128, 128), batch_size=256, repeat_times=32.0, use_v_trace=True, seed=14)
            elif name == "sac_mod":
                make_sac_mod("small", N=4, S=11, A=3, rows=40, net_dims=(64, 32), batch_size=64, n_updates=4, seed=23)
            elif name == "a2c":
                make_a2c("small", S=6, A=3, H=24, net_dims=(64, 32), batch_size=16, repeat_times=2.0, seed=41)
                make_a2c("mid", S=64, A=8, H=48, net_dims=(128, 128), batch_size=32, repeat_times=2.0, seed=42)
            else:
                globals()[f"make_{name}"]()
        sys.exit(0)
    make_ppo("small_vtrace", N=8, S=6, A=2, H=12, net_dims=(64, 32), batch_size=16, repeat_times=4.0,
             use_v_trace=True, seed=11)
    make_ppo("small_alt", N=8, S=6, A=2, H=12, net_dims=(64, 32), batch_size=16, repeat_times=4.0,
             use_v_trace=False, seed=12)
    make_ppo("mid_vtrace", N=40, S=17, A=5, H=20, net_dims=(128, 128), batch_size=64, repeat_times=6.4,
             use_v_trace=True, seed=13)
    # the benchmark's own kernel instance (S = 64, A = 8, net [128,128], 16-byte-aligned rows): two minibatches of 256 out of 16 x 64
    make_ppo("c4shape", N=64, S=64, A=8, H=16, net_dims=(128, 128), batch_size=256, repeat_times=32.0, use_v_trace=True, seed=14)
    make_replay()
    make_replay_discrete()
    make_sac("small", N=4, S=11, A=3, rows=40, net_dims=(64, 32), batch_size=64, n_updates=3, seed=21)
    make_sac("fit_cum_r", N=4, S=11, A=3, rows=40, net_dims=(64, 32), batch_size=64, n_updates=3, seed=22, lambda_fit=0.3)
    make_ppo_discrete("small", N=8, S=6, A=4, H=12, net_dims=(64, 32), batch_size=16, repeat_times=4.0, seed=31)
    make_cum_rewards()
    make_a2c("small", S=6, A=3, H=24, net_dims=(64, 32), batch_size=16, repeat_times=2.0, seed=41)
    make_a2c("mid", S=64, A=8, H=48, net_dims=(128, 128), batch_size=32, repeat_times=2.0, seed=42)
    make_evaluator()
    make_sac_mod("small", N=4, S=11, A=3, rows=40, net_dims=(64, 32), batch_size=64, n_updates=4, seed=23)
    make_state_norm()
    make_per_update()
