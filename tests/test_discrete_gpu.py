"""Discrete-action PPO (AgentDiscretePPO / ActorDiscretePPO, elegantrl/agents/AgentPPO.py:305-320,
:393-422) on the layered path: categorical rollout step and the PPO objective with its state-dependent entropy against
the numpy oracle (fp64), the agent's update_net against the reference's own recorded run (tests/golden/
ppo_discrete_small.npz), and end-to-end learning on a CartPole vector env."""
import os

import numpy as np
import pytest
import torch as th

from oracle import ppo_numpy as O
from tests.helpers import hyper, load
from tests.test_kernels_gpu import cu, flat_params
from tests.test_mlpn_gpu import random_net_n

pytestmark = pytest.mark.gpu
DEV = th.device("cuda:0")
SHAPES = [(6, (64, 32), 4), (24, (256, 128), 6), (17, (256, 128, 64), 2), (4, (32,), 64), (9, (48, 16), 1)]


@pytest.fixture(scope="module")
def ops():
    from elegantrl_amd import ops as _ops
    return _ops


def spread(actor, k=6.0):
    actor.weights[-1] *= np.float32(k)        # logits far from uniform so that every branch of the draw is exercised
    return actor


@pytest.mark.parametrize("S,hidden,A", SHAPES)
def test_rollout_step_discrete_injected_uniforms(ops, S, hidden, A):
    rng = np.random.default_rng(S + A)
    N = 1500
    actor = spread(random_net_n(rng, [S, *hidden, A], False))
    x = rng.standard_normal((N, S), dtype=np.float32)
    u = rng.random(N, dtype=np.float32)
    spec = ops.MlpSpecN([S, *hidden, A], False)
    o_s, o_a, o_l = th.zeros((N, S), device=DEV), th.full((N,), -1, dtype=th.int32, device=DEV), th.zeros(N, device=DEV)
    o_e = th.full((N,), -1, dtype=th.int64, device=DEV)
    ops.mlpn_rollout_step_discrete(cu(flat_params(actor), DEV), spec, cu(actor.state_avg, DEV), cu(actor.state_std, DEV), cu(x, DEV),
                                   uniform=cu(u, DEV), out_state=o_s, out_action=o_a, out_logprob=o_l, out_env_action=o_e)
    a64 = actor.astype(np.float64)
    p = O.softmax(O.actor_mean(x.astype(np.float64), a64))
    c = np.cumsum(p, axis=1)
    a_ref, _ = O.categorical_sample(x.astype(np.float64), a64, u.astype(np.float64))
    got = o_a.cpu().numpy()
    # a draw within fp32 noise of a cumulative boundary may legitimately fall on either side
    near = (np.abs(c - u[:, None].astype(np.float64)) < 2e-6).any(axis=1)
    np.testing.assert_array_equal(got[~near], a_ref[~near])
    assert (np.abs(got[near] - a_ref[near]) <= 1).all() and near.mean() < 0.01
    np.testing.assert_array_equal(o_e.cpu().numpy(), got.astype(np.int64))
    np.testing.assert_array_equal(o_s.cpu().numpy(), x)
    lp_ref = O.categorical_logits(p)[np.arange(N), got]
    np.testing.assert_allclose(o_l.cpu().numpy(), lp_ref, rtol=1e-4, atol=1e-4)


def test_rollout_step_discrete_philox_frequencies_and_determinism(ops):
    rng = np.random.default_rng(1)
    S, hidden, A, N = 5, (32, 32), 5, 200_000
    actor = random_net_n(rng, [S, *hidden, A], False)
    x = np.tile(rng.standard_normal((1, S)).astype(np.float32), (N, 1))          # one state: every env draws from the same p
    spec = ops.MlpSpecN([S, *hidden, A], False)
    P, avg, sd, X = cu(flat_params(actor), DEV), cu(actor.state_avg, DEV), cu(actor.state_std, DEV), cu(x, DEV)

    def run(counter):
        o = th.empty((N,), dtype=th.int32, device=DEV)
        ops.mlpn_rollout_step_discrete(P, spec, avg, sd, X, seed=77, counter=counter, out_action=o)
        return o.cpu().numpy()

    a0, a0b, a1 = run(3), run(3), run(4)
    np.testing.assert_array_equal(a0, a0b)
    assert (a0 != a1).mean() > 0.3
    p = O.softmax(O.actor_mean(x[:1].astype(np.float64), actor.astype(np.float64)))[0]
    freq = np.bincount(a0, minlength=A) / N
    assert (np.abs(freq - p) <= 4 * np.sqrt(p * (1 - p) / N) + 1e-4).all(), (freq, p)      # 4 sigma of the binomial


def discrete_case(rng, H, N, S, A, B):
    states = rng.standard_normal((H, N, S), dtype=np.float32)
    actions = rng.integers(0, A, (H, N)).astype(np.int32)
    um = rng.random((H, N)) > 0.1
    lp = (-np.log(A) + 0.3 * rng.standard_normal((H, N))).astype(np.float32)
    adv = rng.standard_normal((H, N)).astype(np.float32)
    rs = rng.standard_normal((H, N)).astype(np.float32)
    ids = rng.integers(0, H * N, B).astype(np.int64)
    return states, actions, um, lp, adv, rs, ids


@pytest.mark.parametrize("S,hidden,A", SHAPES)
@pytest.mark.parametrize("B", [64, 1000, 1300])
def test_ppo_step_discrete_gradients(ops, S, hidden, A, B):
    rng = np.random.default_rng(S + B + A)
    H, N = 9, 50
    states, actions, um, lp, adv, rs, ids = discrete_case(rng, H, N, S, A, B)
    actor, critic = spread(random_net_n(rng, [S, *hidden, A], False), 3.0), random_net_n(rng, [S, *hidden, 1], False)
    spec = ops.MlpSpecN([S, *hidden, A], False)
    Pa, Pc = spec.count, ops.MlpSpecN([S, *hidden, 1], False).count
    flat = th.full((Pa + Pc + 4,), float("nan"), device=DEV)
    lam = 0.01
    ops.mlpn_ppo_step_discrete(cu(flat_params(actor), DEV), cu(flat_params(critic), DEV), cu(actor.state_avg, DEV),
                               cu(actor.state_std, DEV), cu(critic.state_avg, DEV), cu(critic.state_std, DEV), spec, cu(states, DEV),
                               cu(actions, DEV), cu(um, DEV), cu(lp, DEV), cu(adv, DEV), cu(rs, DEV), cu(ids, DEV), 0.25, lam, 1.0 / B, flat)
    got = flat.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    dt = np.float64
    i0, i1 = O.split_ids(ids, H)
    s = states[i0, i1].astype(dt)
    oc, gw, gb = O.critic_objective(s, rs[i0, i1].astype(dt), um[i0, i1], critic.astype(dt))
    gc = np.concatenate([x.reshape(-1) for pair in zip(gw, gb) for x in pair])
    os_, oe, gw, gb = O.actor_objective_discrete(s, actions[i0, i1], lp[i0, i1].astype(dt), adv[i0, i1].astype(dt), um[i0, i1],
                                                 actor.astype(dt), 0.25, lam)
    ga = np.concatenate([x.reshape(-1) for pair in zip(gw, gb) for x in pair])
    for name, g, ref in (("actor", got[:Pa], ga), ("critic", got[Pa:Pa + Pc], gc)):
        scale = np.abs(ref).max()
        assert np.abs(g - ref).max() <= 1e-4 * scale + 1e-7, f"{name} grad err {np.abs(g - ref).max():.3e} (scale {scale:.3e})"
    np.testing.assert_allclose(got[Pa + Pc:Pa + Pc + 3], [oc, os_, oe], rtol=1e-4, atol=1e-6)


def make_agent(g, agent_class=None):
    from elegantrl_amd.agents import AgentDiscretePPO
    from elegantrl_amd.train import Config
    AgentDiscretePPO = agent_class or AgentDiscretePPO
    hp = hyper(g)
    N, S, A, H, B, n_upd, _, *net = [int(x) for x in g["dims"]]
    args = Config(AgentDiscretePPO, None, {"env_name": "golden", "num_envs": N, "max_step": 100, "state_dim": S, "action_dim": A,
                                           "if_discrete": True})
    args.net_dims = net
    args.horizon_len, args.batch_size, args.repeat_times = H, B, n_upd * B / H
    args.learning_rate, args.gamma, args.reward_scale, args.clip_grad_norm = hp["lr"], hp["gamma"], hp["reward_scale"], hp["max_norm"]
    args.lambda_gae_adv, args.ratio_clip, args.lambda_entropy = hp["lam"], hp["ratio_clip"], hp["lambda_entropy"]
    agent = AgentDiscretePPO(args.net_dims, S, A, gpu_id=0, args=args)
    with th.no_grad():
        for net, prefix in ((agent.act, "act0"), (agent.cri, "cri0")):
            net.load_state_dict({k[len(prefix) + 1:]: th.from_numpy(v) for k, v in g.items() if k.startswith(prefix + ".")})
    return agent, args


def test_agent_rollout_rows_match_reference_logprobs():
    """feeding the reference's recorded states: the kernel's logits-path log-prob of the reference's own actions"""
    g = load("ppo_discrete_small.npz")
    agent, _ = make_agent(g)
    assert not agent._fused and agent._discrete and agent.lambda_entropy_value == pytest.approx(0.01)
    H, N = g["actions"].shape
    # choose u inside the probability interval of the recorded action => the kernel must reproduce action and log-prob
    p = O.softmax(g["logits"].reshape(H * N, -1).astype(np.float64))
    c = np.cumsum(p, axis=1)
    a = g["actions"].reshape(-1)
    lo = np.where(a > 0, c[np.arange(H * N), np.maximum(a - 1, 0)], 0.0)
    u = (0.5 * (lo + c[np.arange(H * N), a])).astype(np.float32).reshape(H, N)
    for t in range(H):
        act, lp = agent.explore_action(th.from_numpy(g["states"][t]).to(DEV), uniform=th.from_numpy(u[t]).to(DEV))
        np.testing.assert_array_equal(act.cpu().numpy(), g["actions"][t])
        np.testing.assert_allclose(lp.cpu().numpy(), g["logprobs"][t], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("cls", ["AgentDiscretePPO", "AgentDiscreteA2C"])
def test_agent_update_net_matches_reference_weights_and_objectives(cls):
    """AgentDiscreteA2C: the reference's class inherits AgentPPO.update_net through AgentDiscretePPO (AgentPPO.py:332-342 overrides
    only the constructor), so the same recorded run pins it."""
    import elegantrl_amd.agents as agents
    g = load("ppo_discrete_small.npz")
    agent, _ = make_agent(g, getattr(agents, cls))
    agent.last_state = th.from_numpy(g["last_state"]).to(DEV)
    buf = [th.from_numpy(g[k]).to(DEV) for k in ("states", "actions", "logprobs", "rewards", "undones", "unmasks")]
    objs = agent.update_net(buf, ids=th.from_numpy(g["ids"]).to(DEV))
    np.testing.assert_allclose(np.array(objs), g["objs"], rtol=5e-4, atol=5e-6)
    for net, prefix in ((agent.act, "act1"), (agent.cri, "cri1")):
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), g[f"{prefix}.{k}"], rtol=0, atol=3e-5, err_msg=f"{prefix}.{k}")
    assert np.abs(agent.act.net[0].weight.detach().cpu().numpy() - g["act0.net.0.weight"]).max() > 1e-4


@pytest.mark.timeout(600)
def test_train_agent_discrete_ppo_cartpole_learns(tmp_path):
    """the whole loop (train_agent, Evaluator, checkpoints) with the categorical policy: CartPole-v1 dynamics on 512 device
    envs; a random policy survives ~22 steps, the trained one must exceed 150 at some evaluation."""
    from elegantrl_amd import train_agent
    from elegantrl_amd.agents import AgentDiscretePPO
    from elegantrl_amd.envs import CartPoleVecEnv
    from elegantrl_amd.train import Config
    args = Config(AgentDiscretePPO, CartPoleVecEnv, {"env_name": "CartPole-v1", "num_envs": 512, "max_step": 500, "state_dim": 4,
                                                     "action_dim": 2, "if_discrete": True})
    args.net_dims = [64, 32]
    args.horizon_len, args.batch_size, args.repeat_times = 64, 4096, 4096 * 8 / 64
    args.gamma, args.learning_rate, args.lambda_entropy = 0.98, 2e-3, 0.01
    args.break_step, args.eval_per_step, args.eval_times = 64 * 40, 64 * 8, 8
    args.cwd, args.gpu_id, args.random_seed = str(tmp_path / "run"), 0, 0
    args.gae_algo = "exact"    # fixed association: the run is reproducible bit for bit (the look-back scan is not)
    train_agent(args, if_single_process=True)
    rec = np.load(os.path.join(args.cwd, "recorder.npy"))
    assert np.isfinite(rec[:, :4]).all()
    assert rec[:, 1].max() > 150.0, f"discrete PPO did not learn CartPole: evaluated returns {np.round(rec[:, 1], 1).tolist()}"
