"""Pin the numpy oracle to the reference's real outputs (tests/golden, made by oracle/make_golden.py)."""
import numpy as np
import pytest

from oracle import ppo_numpy as O
from tests.helpers import PPO_GOLDENS, dims, hyper, load, mlp_from


@pytest.mark.parametrize("name", PPO_GOLDENS)
def test_rollout_action_logprob(name):
    g = load(name)
    actor = mlp_from(g, "act0")
    H = dims(g)["H"]
    for t in range(H):
        a, lp = O.actor_sample(g["states"][t], actor, g["eps"][t].astype(np.float32))
        np.testing.assert_allclose(a, g["actions"][t], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(lp, g["logprobs"][t], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", PPO_GOLDENS)
def test_values(name):
    g = load(name)
    critic = mlp_from(g, "cri0")
    v = O.critic_value(g["states"], critic)
    np.testing.assert_allclose(v, g["values"], rtol=1e-5, atol=2e-6)
    nv = O.critic_value(g["last_state"], critic)
    np.testing.assert_allclose(nv, g["next_value"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("name", PPO_GOLDENS)
def test_gae_matches_reference(name):
    g = load(name)
    hp, d = hyper(g), dims(g)
    adv, r2, u2 = O.gae_scan(g["rewards"], g["undones"], g["unmasks"], g["values"], g["next_value"],
                             hp["gamma"], hp["lam"], use_v_trace=d["vtrace"])
    assert (~g["unmasks"]).any(), "fixture must exercise the truncation fix-up"
    assert (~g["undones"]).any()
    np.testing.assert_array_equal(u2, g["undones_after"])
    np.testing.assert_allclose(r2, g["rewards_after"], rtol=0, atol=2e-6)
    # the reference re-runs the critic on the truncated rows; everything else is the same op order
    np.testing.assert_allclose(adv, g["advantages"], rtol=1e-5, atol=1e-5)          # |x - ref| <= 1e-5 max(1, |ref|) (np: atol + rtol |ref|)
    np.testing.assert_allclose(O.reward_sums(adv, g["values"]), g["reward_sums"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(O.adv_normalize(g["advantages"]), g["advantages_norm"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", PPO_GOLDENS)
def test_gae_fp64_bounds_fp32(name):
    g = load(name)
    hp, d = hyper(g), dims(g)
    adv64, _, _ = O.gae_scan(g["rewards"].astype(np.float64), g["undones"], g["unmasks"],
                             g["values"].astype(np.float64), g["next_value"].astype(np.float64),
                             hp["gamma"], hp["lam"], use_v_trace=d["vtrace"])
    np.testing.assert_allclose(adv64, g["advantages"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", PPO_GOLDENS)
def test_update_net_weights_and_objectives(name):
    """Full minibatch loop (gather -> objectives -> manual backward -> clip -> Adam) vs the reference's
    autograd + torch.optim.Adam, on the reference's own recorded randint ids."""
    g = load(name)
    hp, d = hyper(g), dims(g)
    actor, critic = mlp_from(g, "act0"), mlp_from(g, "cri0")
    buf = (g["states"], g["actions"], g["unmasks"], g["logprobs"], g["advantages_norm"], g["reward_sums"])
    sa, sc = O.AdamState(), O.AdamState()
    objs = []
    for ids in g["ids"]:
        objs.append(O.ppo_minibatch_step(buf, ids, actor, critic, sa, sc, lr=hp["lr"], max_norm=hp["max_norm"],
                                         ratio_clip=hp["ratio_clip"], lambda_entropy=hp["lambda_entropy"]))
    objs = np.array(objs, dtype=np.float64).mean(axis=0)
    np.testing.assert_allclose(objs, g["objs"], rtol=2e-4, atol=2e-6)
    ref_a, ref_c = mlp_from(g, "act1"), mlp_from(g, "cri1")
    for mine, ref in ((actor, ref_a), (critic, ref_c)):
        for p, q in zip(mine.trainable(), ref.trainable()):
            # measured <= 1.3e-6 on the small fixtures, 9e-6 on ONE of the 50k weights of ppo_c4shape (a gradient of ~1e-9: Adam's first steps
            # are lr * g / (|g| + eps), where fp32 autograd and the fp64 restatement disagree)
            np.testing.assert_allclose(p, q, rtol=0, atol=2e-5 if name == "ppo_c4shape.npz" else 5e-6)
    moved = sum(float(np.abs(p - q).sum()) for p, q in zip(mlp_from(g, "act0").trainable(), ref_a.trainable()))
    assert moved > 0


@pytest.mark.parametrize("name", ["replay_ring.npz", "replay_ring_discrete.npz"])
def test_replay_ring_cursors_indices_and_rows(name):
    g = load(name)
    max_size, S, A, num_seqs = [int(x) for x in g["dims"]]
    ring = O.Ring(max_size, S, A, num_seqs, if_discrete="discrete" in name)
    for k, add in enumerate(g["adds"]):
        items = tuple(g[f"in{k}_{n}"] for n in ("states", "actions", "rewards", "undones", "unmasks"))
        ring.update(items)
        assert [ring.p, ring.cur_size, int(ring.if_full), ring.add_size] == list(g[f"cursor{k}"])
        for n in ("states", "actions", "rewards", "undones", "unmasks"):
            np.testing.assert_array_equal(getattr(ring, n), g[f"buf{k}_{n}"])
        out, (i0, i1) = ring.sample(g[f"ids{k}"])
        np.testing.assert_array_equal(i0, g[f"ids0_{k}"])
        np.testing.assert_array_equal(i1, g[f"ids1_{k}"])
        for arr, n in zip(out, ("state", "action", "reward", "undone", "unmask", "next_state")):
            np.testing.assert_array_equal(arr, g[f"out{k}_{n}"])
    assert list(g["cursor2"][:3]) == [max_size, max_size, 0]  # lands exactly on max_size: not yet "full"


def test_cum_rewards_matches_reference_bitwise():
    """AgentBase.get_cumulative_rewards run through ReplayBuffer.update_cum_rewards by the reference (4 appends, one wrapping
    with p < add_size): the fp32 numpy restatement reproduces the recorded returns bit for bit, and the slice rule too."""
    g = load("cum_rewards.npz")
    N, S, A, max_size = [int(x) for x in g["dims"]]
    gamma = float(g["gamma"][0])
    cum = np.zeros((max_size, N), np.float32)
    for k in range(len(g["adds"])):
        p, cur, full, add = [int(x) for x in g[f"cursor{k}"]]
        p0, p1 = O.cum_rewards_slice(p, add, max_size)
        assert [p0, p1] == list(g[f"slice{k}"])
        out = O.cum_rewards(g[f"rewards{k}"][p0:p1], g[f"undones{k}"][p0:p1], g[f"next_value{k}"], gamma)
        np.testing.assert_array_equal(out, g[f"direct{k}"])
        cum[p0:p1] = out
        np.testing.assert_array_equal(cum, g[f"cum_rewards{k}"])
    assert list(g["slice2"]) == [max_size - 11, max_size]      # the p < add_size branch was exercised


@pytest.mark.parametrize("name", PPO_GOLDENS)
def test_c_oracle_gae_bitwise_equals_numpy_and_matches_reference(name):
    from oracle import c_oracle
    g = load(name)
    hp, d = hyper(g), dims(g)
    adv_c, ret_c, r_c, u_c = c_oracle.gae(g["rewards"], g["undones"], g["unmasks"], g["values"], g["next_value"],
                                          hp["gamma"], hp["lam"], use_v_trace=d["vtrace"])
    adv_n, r_n, u_n = O.gae_scan(g["rewards"], g["undones"], g["unmasks"], g["values"], g["next_value"],
                                 hp["gamma"], hp["lam"], use_v_trace=d["vtrace"])
    np.testing.assert_array_equal(adv_c, adv_n)           # same op order, no contraction: bit-identical
    np.testing.assert_array_equal(ret_c, adv_n + g["values"])
    np.testing.assert_array_equal(r_c, r_n)
    np.testing.assert_array_equal(u_c, u_n)
    np.testing.assert_allclose(adv_c, g["advantages"], rtol=1e-5, atol=1e-5)


def test_c_oracle_cols_variant_equals_plain():
    from oracle import c_oracle
    rng = np.random.default_rng(0)
    H, N = 37, 133
    r = rng.standard_normal((H, N), dtype=np.float32)
    v = rng.standard_normal((H, N), dtype=np.float32)
    u = rng.random((H, N)) < 0.95
    m = rng.random((H, N)) < 0.97
    nv = rng.standard_normal(N, dtype=np.float32)
    adv, ret, r2, u2 = c_oracle.gae(r, u, m, v, nv, 0.99, 0.95)
    rr, uu, mm = r.copy(), u.astype(np.uint8), m.astype(np.uint8)
    a2, t2 = np.empty_like(r), np.empty_like(r)
    c_oracle.gae_cols_inplace(rr, uu, mm, v, nv, a2, t2, 0, 70, 0.99, 0.95)
    c_oracle.gae_cols_inplace(rr, uu, mm, v, nv, a2, t2, 70, N, 0.99, 0.95)
    np.testing.assert_array_equal(a2, adv)
    np.testing.assert_array_equal(t2, ret)
    np.testing.assert_array_equal(rr, r2)


def _port_from_golden(g):
    import torch as th
    from oracle.torch_port import TorchPortPPO
    hp, d = hyper(g), dims(g)
    port = TorchPortPPO(d["S"], d["A"], (d["h1"], d["h2"]), lr=hp["lr"], gamma=hp["gamma"], lam=hp["lam"],
                        ratio_clip=hp["ratio_clip"], lambda_entropy=hp["lambda_entropy"], max_norm=hp["max_norm"],
                        reward_scale=hp["reward_scale"], use_v_trace=d["vtrace"])
    for net, prefix in ((port.actor, "act0"), (port.critic, "cri0")):
        net.load_state_dict({k[len(prefix) + 1:]: th.from_numpy(v) for k, v in g.items() if k.startswith(prefix + ".")})
    port.last_state = th.from_numpy(g["last_state"])
    return port


@pytest.mark.parametrize("name", PPO_GOLDENS)
def test_torch_port_update_matches_reference(name):
    """the torch CPU port (bench.py's cpu_baseline) reproduces the reference's update_net on recorded ids."""
    import torch as th
    g = load(name)
    d = dims(g)
    port = _port_from_golden(g)
    buf = [th.from_numpy(g[k]).clone() for k in ("states", "actions", "logprobs", "rewards", "undones", "unmasks")]
    objs = port.update(buf, d["B"], d["n_upd"], ids=th.from_numpy(g["ids"]))
    np.testing.assert_allclose(np.array(objs), g["objs"], rtol=1e-5, atol=1e-7)
    for net, prefix in ((port.actor, "act1"), (port.critic, "cri1")):
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(v.numpy(), g[f"{prefix}.{k}"], rtol=0, atol=1e-6, err_msg=k)
    np.testing.assert_array_equal(buf[4].numpy(), g["undones_after"])      # in-place side effect kept


def test_torch_port_rollout_shapes_and_env_rules():
    import torch as th
    from oracle.torch_port import TorchPortPPO, TorchSynEnv
    th.manual_seed(0)
    env = TorchSynEnv(num_envs=64, state_dim=8, action_dim=2, max_step=5, seed=1)
    port = TorchPortPPO(8, 2, (32, 32))
    port.last_state = env.reset()
    s, a, lp, r, ud, um = port.explore(env, 12)
    assert s.shape == (12, 64, 8) and a.shape == (12, 64, 2) and lp.shape == r.shape == (12, 64)
    assert ud.dtype == th.bool and um.dtype == th.bool and (~um).any()       # truncation every 5 steps
    a_ref, lp_ref = O.actor_sample(s[3].numpy(), O.Mlp([port.actor.net[i].weight.detach().numpy() for i in (0, 2, 4)],
                                                      [port.actor.net[i].bias.detach().numpy() for i in (0, 2, 4)],
                                                      np.zeros(8, np.float32), np.ones(8, np.float32), np.zeros(2, np.float32)),
                                   np.zeros((64, 2), np.float32))
    mean = a_ref                                                             # eps = 0 -> action == mean
    np.testing.assert_allclose(lp[3].numpy(), O.gaussian_logprob(a[3].numpy(), mean, np.zeros(2, np.float32)), rtol=1e-5, atol=1e-5)


# ---- discrete policy (AgentDiscretePPO / ActorDiscretePPO), fixture ppo_discrete_small.npz ---------------------------
def _discrete_actor(g, prefix):
    a = mlp_from(g, prefix)
    a.action_std_log = None          # ActorDiscretePPO inherits the parameter but never uses it (its grad stays None)
    return a


def test_discrete_logprob_and_sampling_rule():
    g = load("ppo_discrete_small.npz")
    actor = _discrete_actor(g, "act0")
    H, N, A = g["actions"].shape[0], g["actions"].shape[1], g["logits"].shape[-1]
    for t in range(H):
        z = O.actor_mean(g["states"][t], actor)
        np.testing.assert_allclose(z, g["logits"][t], rtol=1e-5, atol=1e-5)
        L = O.categorical_logits(O.softmax(z))
        np.testing.assert_allclose(L[np.arange(N), g["actions"][t]], g["logprobs"][t], rtol=1e-5, atol=1e-5)   # dist.log_prob
    # the inverse-CDF draw: u just below / above a cumulative boundary picks the two neighbouring actions
    p = O.softmax(O.actor_mean(g["states"][0], actor))
    c = np.cumsum(p, axis=1)
    for k in range(A - 1):
        lo, _ = O.categorical_sample(g["states"][0], actor, (c[:, k] - 1e-6).astype(np.float32))
        hi, _ = O.categorical_sample(g["states"][0], actor, (c[:, k] + 1e-6).astype(np.float32))
        assert (lo <= k).all() and (hi >= k + 1).all()
    a, lp = O.categorical_sample(g["states"][0], actor, np.full(N, 0.999999, np.float32))
    assert a.max() <= A - 1 and np.isfinite(lp).all()


def test_discrete_update_net_weights_and_objectives():
    """the reference's AgentDiscretePPO.update_net on its own recorded ids: manual softmax / entropy backward vs autograd."""
    g = load("ppo_discrete_small.npz")
    hp = hyper(g)
    actor, critic = _discrete_actor(g, "act0"), mlp_from(g, "cri0")
    buf = (g["states"], g["actions"], g["unmasks"], g["logprobs"], g["advantages_norm"], g["reward_sums"])
    sa, sc = O.AdamState(), O.AdamState()
    objs = [O.ppo_minibatch_step_discrete(buf, ids, actor, critic, sa, sc, lr=hp["lr"], max_norm=hp["max_norm"],
                                          ratio_clip=hp["ratio_clip"], lambda_entropy=hp["lambda_entropy"]) for ids in g["ids"]]
    np.testing.assert_allclose(np.array(objs, dtype=np.float64).mean(axis=0), g["objs"], rtol=2e-4, atol=2e-6)
    ref_a, ref_c = _discrete_actor(g, "act1"), mlp_from(g, "cri1")
    for mine, ref in ((actor, ref_a), (critic, ref_c)):
        for p, q in zip(mine.trainable(), ref.trainable()):
            np.testing.assert_allclose(p, q, rtol=0, atol=5e-6)
    assert sum(float(np.abs(p - q).sum()) for p, q in zip(_discrete_actor(g, "act0").trainable(), ref_a.trainable())) > 0


# ---- PPO siblings: AgentA2C (reference golden) and the objective forms (torch autograd of the quoted expressions) ----------
@pytest.mark.parametrize("name", ["a2c_small.npz", "a2c_mid.npz"])
def test_a2c_update_net_weights_and_objectives(name):
    """AgentA2C.update_net (elegantrl/agents/AgentPPO.py:256-303) on a one-env buffer, replayed by the numpy restatement on the
    reference's recorded time indices: GAE + normalisation as AgentPPO, then the un-clipped objective per minibatch."""
    g = load(name)
    gamma, lam, lr, max_norm = (float(x) for x in g["hyper"])
    actor, critic = mlp_from(g, "act0"), mlp_from(g, "cri0")
    v = O.critic_value(g["states"], critic)
    nv = O.critic_value(g["last_state"], critic)
    np.testing.assert_allclose(v, g["values"], rtol=1e-5, atol=1e-6)
    adv, _, _ = O.gae_scan(g["rewards"].copy(), g["undones"].copy(), g["unmasks"], g["values"], nv, gamma, lam)
    np.testing.assert_allclose(adv, g["advantages"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(O.adv_normalize(g["advantages"]), g["advantages_norm"], rtol=1e-5, atol=1e-5)
    buf = (g["states"], g["actions"], g["unmasks"], g["logprobs"], g["advantages_norm"], g["reward_sums"])
    sa, sc = O.AdamState(), O.AdamState()
    objs = [O.ppo_minibatch_step(buf, ids, actor, critic, sa, sc, lr=lr, max_norm=max_norm, ratio_clip=0.25, lambda_entropy=0.0,
                                 objective="a2c") for ids in g["ids"]]
    objs = np.array(objs, dtype=np.float64).mean(axis=0)
    np.testing.assert_allclose(objs[:2], g["objs"][:2], rtol=2e-4, atol=2e-6)
    assert g["objs"][2] == 0 and objs[2] == 0
    for mine, ref in ((actor, mlp_from(g, "act1")), (critic, mlp_from(g, "cri1"))):
        for p, q in zip(mine.trainable(), ref.trainable()):
            np.testing.assert_allclose(p, q, rtol=0, atol=5e-6)


@pytest.mark.parametrize("objective", ["reference", "canonical", "a2c"])
def test_actor_objective_forms_against_torch_autograd(objective):
    """the three actor objectives of oracle/ppo_numpy.actor_objective (= csrc/ppo_objective.h) against torch autograd of the
    expressions they quote: AgentPPO.py:196-204, helloworld_PPO_single_file.py:337-339 (inside the same loss), AgentPPO.py:301."""
    import torch as th
    rng = np.random.default_rng(5)
    B, S, A, h1, h2, clip, lam = 96, 7, 3, 32, 32, 0.25, 0.01
    ws = [rng.standard_normal(s) * 0.3 for s in ((h1, S), (h2, h1), (A, h2))]
    bs = [rng.standard_normal(n) * 0.1 for n in (h1, h2, A)]
    actor = O.Mlp(ws, bs, rng.standard_normal(S) * 0.1, 1 + 0.2 * rng.random(S), rng.standard_normal(A) * 0.2 - 0.3)
    s, a = rng.standard_normal((B, S)), rng.standard_normal((B, A))
    adv, um = rng.standard_normal(B), rng.random(B) > 0.2
    mean0 = O.actor_mean(s, actor)
    lp_old = O.gaussian_logprob(a, mean0, actor.action_std_log) + 0.4 * rng.standard_normal(B)    # ratios on both sides of the clip
    obj_s, obj_e, gw, gb, gsl = O.actor_objective(s, a, lp_old, adv, um, actor, clip, lam, objective)

    t = lambda x: th.tensor(np.asarray(x), dtype=th.float64)  # noqa: E731
    W = [t(w).requires_grad_() for w in ws]
    Bv = [t(b).requires_grad_() for b in bs]
    sl = t(actor.action_std_log).requires_grad_()
    x = (t(s) - t(actor.state_avg)) / (t(actor.state_std) + 1e-4)
    h = th.nn.functional.gelu(x @ W[0].T + Bv[0])
    h = th.nn.functional.gelu(h @ W[1].T + Bv[1])
    mu = h @ W[2].T + Bv[2]
    dist = th.distributions.Normal(mu, sl.exp())
    logp_a = dist.log_prob(t(a))                       # (B, A)
    ent = dist.entropy().sum(1)
    tum, tadv = t(um.astype(np.float64)), t(adv)
    if objective == "a2c":
        obj = (tadv[:, None] * logp_a).mean()          # AgentPPO.py:301 with new_logprob of shape (B, A)
        loss, ref_e = -obj, 0.0
    else:
        ratio = (logp_a.sum(1) - t(lp_old)).exp()
        if objective == "canonical":
            surr = th.min(tadv * ratio, tadv * ratio.clamp(1 - clip, 1 + clip))
        else:
            surr = tadv * ratio * th.where(tadv > 0, 1 - clip, 1 + clip)
        obj = (surr * tum).mean()
        ref_e = (ent * tum).mean()
        loss = -(obj - ref_e * lam)
        ref_e = float(ref_e.detach())
    loss.backward()
    assert abs(float(obj) - obj_s) <= 1e-10 * max(1, abs(float(obj))) and abs(ref_e - obj_e) <= 1e-10
    for mine, ref in zip(list(gw) + list(gb) + [gsl], W + Bv + [sl]):
        np.testing.assert_allclose(mine, ref.grad.numpy().reshape(mine.shape), rtol=1e-9, atol=1e-12)
