"""Off-policy half of the hot path: AgentSAC's rollout / update loop on the HIP replay
kernels, pinned to tests/golden/sac_small.npz -- outputs of the reference's own AgentSAC (oracle/make_golden.py:make_sac)
with every random draw (minibatch ids, both rsample() noises per step) recorded so the steps can be replayed."""
import numpy as np
import pytest
import torch as th

from tests.helpers import load


def _load_nets(g, tag, act, cri):
    act.load_state_dict({k[len(f"act{tag}."):]: th.from_numpy(v) for k, v in g.items() if k.startswith(f"act{tag}.")})
    cri.load_state_dict({k[len(f"cri{tag}."):]: th.from_numpy(v) for k, v in g.items() if k.startswith(f"cri{tag}.")})


SAC_GOLDENS = ["sac_small.npz", "sac_fit_cum_r.npz"]      # the second: lambda_fit_cum_r = 0.3 (AgentSAC.py:66-68), seeded cum_rewards


@pytest.mark.parametrize("name", SAC_GOLDENS)
def test_torch_restatement_replays_the_reference(name):
    """oracle/sac_torch.py (CPU) against the reference-generated golden: objectives, actor, critics, target and alpha after
    each of the 3 recorded update steps."""
    from oracle.sac_torch import SacStepper
    th.set_grad_enabled(True)
    g = load(name)
    lam = float(g["lambda_fit_cum_r"][0]) if "lambda_fit_cum_r" in g else 0.0
    N, S, A, rows, B, n_upd, n_ens, h1, h2 = [int(x) for x in g["dims"]]
    gamma, lr, max_norm, reward_scale, tau, target_entropy = [float(x) for x in g["hyper"]]
    st = SacStepper([h1, h2], S, A, n_ens, lr, gamma, tau, max_norm)
    _load_nets(g, 0, st.act, st.cri)
    st.cri_target.load_state_dict(st.cri.state_dict())
    with th.no_grad():
        st.alpha_log[:] = th.from_numpy(g["alpha_log0"])
    st.reset_optimizers()
    ring = {k: th.from_numpy(g[f"ro_{k}"]) for k in ("states", "actions", "rewards", "undones", "unmasks")}
    L = rows - 1
    for t in range(n_upd):
        ids = th.from_numpy(g["ids"][t])
        i0, i1 = ids % L, ids // L
        batch = (ring["states"][i0, i1], ring["actions"][i0, i1], ring["rewards"][i0, i1], ring["undones"][i0, i1].float(),
                 ring["unmasks"][i0, i1].float(), ring["states"][i0 + 1, i1])
        cum = th.from_numpy(g["cum_rewards"])[i0, i1] if lam else None
        oc, oa = st.step(batch, th.from_numpy(g["eps_next"][t]), th.from_numpy(g["eps_cur"][t]), cum_reward=cum, lambda_fit_cum_r=lam)
        np.testing.assert_allclose([oc, oa], g["objs"][t], rtol=1e-5, atol=1e-7)
        for prefix, net in ((f"act{t + 1}", st.act), (f"cri{t + 1}", st.cri), (f"crit{t + 1}", st.cri_target)):
            for k, v in net.state_dict().items():
                np.testing.assert_allclose(v.numpy(), g[f"{prefix}.{k}"], rtol=0, atol=2e-6, err_msg=f"{prefix}.{k}")
        np.testing.assert_allclose(st.alpha_log.detach().numpy(), g[f"alpha_log{t + 1}"], rtol=0, atol=1e-6)


def _mod_setup(g):
    N, S, A, rows, B, n_upd, n_ens, h1, h2 = [int(x) for x in g["dims"]]
    gamma, lr, max_norm, reward_scale, tau, target_entropy, critic_tau, critic_value = [float(x) for x in g["hyper"]]
    return (N, S, A, rows, B, n_upd, n_ens, h1, h2), (gamma, lr, max_norm, reward_scale, tau, target_entropy, critic_tau, critic_value)


def test_torch_restatement_replays_the_reference_modsac():
    """oracle/sac_torch.py's ModSacStepper / ActorFixSAC (CPU) against tests/golden/sac_mod_small.npz -- outputs of the reference's own
    AgentModSAC (elegantrl/agents/AgentSAC.py:89-165, :201-243; oracle/make_golden.py:make_sac_mod): objectives (nan where the
    two-time-scale rule skipped the actor), actor, actor target, critics, critic target and alpha after each of the 4 recorded steps."""
    from oracle.sac_torch import ModSacStepper
    th.set_grad_enabled(True)
    g = load("sac_mod_small.npz")
    (N, S, A, rows, B, n_upd, n_ens, h1, h2), (gamma, lr, max_norm, reward_scale, tau, target_entropy, _, _) = _mod_setup(g)
    assert n_ens == 8 and abs(target_entropy + np.log(A)) < 1e-12                 # AgentModSAC's defaults (:93, :107)
    st = ModSacStepper([h1, h2], S, A, n_ens, lr, gamma, tau, max_norm)
    _load_nets(g, 0, st.act, st.cri)
    st.act_target.load_state_dict(st.act.state_dict())
    st.cri_target.load_state_dict(st.cri.state_dict())
    with th.no_grad():
        st.alpha_log[:] = th.from_numpy(g["alpha_log0"])
    st.reset_optimizers()
    with th.no_grad():                                                            # the rollout: ActorFixSAC.get_action (:217-224)
        a = st.act.get_action(th.from_numpy(g["ro_states"][0]), th.from_numpy(g["ro_eps"][0]))
        np.testing.assert_allclose(a.numpy(), g["ro_actions"][0], rtol=1e-5, atol=1e-6)
    ring = {k: th.from_numpy(g[f"ro_{k}"]) for k in ("states", "actions", "rewards", "undones", "unmasks")}
    L = rows - 1
    for t in range(n_upd):
        ids = th.from_numpy(g["ids"][t])
        i0, i1 = ids % L, ids // L
        batch = (ring["states"][i0, i1], ring["actions"][i0, i1], ring["rewards"][i0, i1], ring["undones"][i0, i1].float(),
                 ring["unmasks"][i0, i1].float(), ring["states"][i0 + 1, i1])
        oc, oa = st.step(batch, th.from_numpy(g["eps_next"][t]), th.from_numpy(g["eps_cur"][t]), update_t=t)
        assert np.isnan(oa) == (g["actor_updated"][t] == 0) == bool(np.isnan(g["objs"][t][1]))
        np.testing.assert_allclose([oc, oa], g["objs"][t], rtol=1e-5, atol=1e-7, equal_nan=True)
        for prefix, net in ((f"act{t + 1}", st.act), (f"actt{t + 1}", st.act_target), (f"cri{t + 1}", st.cri), (f"crit{t + 1}", st.cri_target)):
            for k, v in net.state_dict().items():
                np.testing.assert_allclose(v.numpy(), g[f"{prefix}.{k}"], rtol=0, atol=2e-6, err_msg=f"{prefix}.{k}")
        np.testing.assert_allclose(st.alpha_log.detach().numpy(), g[f"alpha_log{t + 1}"], rtol=0, atol=1e-6)
    assert list(g["actor_updated"]) == [1, 1, 0, 1]


def test_actor_fix_sac_module_matches_the_reference_on_cpu():
    """the product's ActorFixSAC module (what the Evaluator calls, what checkpoints hold): same parameter names as the reference's
    (its state_dict loads), same rollout action for the recorded noise"""
    from elegantrl_amd.agents.AgentSAC import ActorFixSAC
    g = load("sac_mod_small.npz")
    (N, S, A, rows, B, n_upd, n_ens, h1, h2), _ = _mod_setup(g)
    act = ActorFixSAC([h1, h2], S, A)
    act.load_state_dict({k[len("act0."):]: th.from_numpy(v) for k, v in g.items() if k.startswith("act0.")})
    with th.no_grad():
        a = act.get_action(th.from_numpy(g["ro_states"][5]), th.from_numpy(g["ro_eps"][5]))
        np.testing.assert_allclose(a.numpy(), g["ro_actions"][5], rtol=1e-5, atol=1e-6)
        assert act(th.from_numpy(g["ro_states"][5])).shape == (N, A)


@pytest.mark.gpu
def test_modsac_rollout_and_updates_replay_the_reference():
    """AgentModSAC on the HIP step (erl_sac_update_opt_f32, erl_sac_explore_action_opt_f32) replays the reference's own AgentModSAC run:
    the off-policy rollout, then 4 update steps with the recorded ids / noise -- objectives (nan on the step the two-time-scale rule
    skips), actor, ACTOR TARGET, critics, critic target, alpha after every step."""
    from elegantrl_amd.agents import AgentModSAC
    from elegantrl_amd.train import Config, ReplayBuffer
    g = load("sac_mod_small.npz")
    (N, S, A, rows, B, n_upd, n_ens, h1, h2), (gamma, lr, max_norm, reward_scale, tau, target_entropy, critic_tau, critic_value) = _mod_setup(g)
    dev = th.device("cuda:0")
    args = Config(AgentModSAC, None, {"env_name": "scripted", "num_envs": N, "max_step": 100, "state_dim": S, "action_dim": A,
                                      "if_discrete": False})
    assert args.if_off_policy
    args.net_dims = [h1, h2]
    args.batch_size, args.learning_rate, args.gamma, args.reward_scale, args.soft_update_tau = B, lr, gamma, reward_scale, tau
    args.clip_grad_norm = max_norm
    agent = AgentModSAC(args.net_dims, S, A, gpu_id=0, args=args)
    assert agent.num_ensembles == n_ens == 8 and abs(agent.target_entropy - target_entropy) < 1e-12
    assert agent.critic_tau == critic_tau and agent.critic_value == critic_value
    _load_nets(g, 0, agent.act, agent.cri)
    agent.act_target.load_state_dict(agent.act.state_dict())
    agent.cri_target.load_state_dict(agent.cri.state_dict())
    with th.no_grad():
        agent.alpha_log[:] = th.from_numpy(g["alpha_log0"]).to(dev)

    agent.last_state = th.from_numpy(g["first_state"]).to(dev)
    th.set_grad_enabled(False)
    items = agent._explore_vec_env(_ReplayEnv(g, dev), rows, noise=th.from_numpy(g["ro_eps"]).to(dev))
    for got, name in zip(items, ("ro_states", "ro_actions", "ro_rewards", "ro_undones", "ro_unmasks")):
        if got.dtype == th.bool:
            np.testing.assert_array_equal(got.cpu().numpy(), g[name])
        else:
            np.testing.assert_allclose(got.cpu().numpy(), g[name], rtol=2e-5, atol=2e-6)

    buf = ReplayBuffer(max_size=rows + 5, state_dim=S, action_dim=A, gpu_id=0, num_seqs=N)
    buf.update(tuple(th.from_numpy(g[n]).to(dev) for n in ("ro_states", "ro_actions", "ro_rewards", "ro_undones", "ro_unmasks")))
    th.set_grad_enabled(True)
    for t in range(n_upd):
        oc, oa = agent.update_objectives(buf, t, ids=th.from_numpy(g["ids"][t]).to(dev),
                                         noises=(th.from_numpy(g["eps_next"][t]).to(dev), th.from_numpy(g["eps_cur"][t]).to(dev)))
        assert np.isnan(oa) == (g["actor_updated"][t] == 0) and agent._last_actor_updated == bool(g["actor_updated"][t])
        np.testing.assert_allclose([oc, oa], g["objs"][t], rtol=2e-4, atol=2e-6, equal_nan=True)
        for prefix, net in ((f"act{t + 1}", agent.act), (f"actt{t + 1}", agent.act_target), (f"cri{t + 1}", agent.cri),
                            (f"crit{t + 1}", agent.cri_target)):
            for k, v in net.state_dict().items():
                np.testing.assert_allclose(v.cpu().numpy(), g[f"{prefix}.{k}"], rtol=0, atol=3e-5, err_msg=f"{prefix}.{k} after step {t}")
        np.testing.assert_allclose(agent.alpha_log.detach().cpu().numpy(), g[f"alpha_log{t + 1}"], rtol=0, atol=1e-5)
    th.set_grad_enabled(False)
    assert agent._actor_step == 3 and agent.act_optimizer.step_count == 3 and agent.cri_optimizer.step_count == 4


@pytest.mark.gpu
@pytest.mark.parametrize("net,E,B", [([256, 256], 8, 256), ([128, 64], 4, 512), ([64], 2, 100)], ids=["config3-net-8critics", "128x64", "one-hidden-layer"])
def test_modsac_step_matches_the_torch_restatement_at_larger_shapes(net, E, B):
    """AgentModSAC's HIP step against oracle/sac_torch.py's ModSacStepper (itself pinned to the reference's run) beyond the golden's
    [64, 32] / 8-critic / batch-64 shape: config 3's network with 8 critics, a [128, 64] net, and ONE hidden layer (ActorFixSAC's encoder is
    then a single raw linear layer: no activation at all).  Three steps on a random batch with injected noise: the third skips the actor."""
    from elegantrl_amd.agents import AgentModSAC
    from elegantrl_amd.train import Config
    from oracle.sac_torch import ModSacStepper
    S, A = 11, 3
    dev = th.device("cuda:0")
    g = th.Generator().manual_seed(len(net) * 100 + E)
    args = Config(AgentModSAC, None, {"env_name": "x", "num_envs": 4, "max_step": 100, "state_dim": S, "action_dim": A, "if_discrete": False})
    args.net_dims, args.batch_size, args.learning_rate, args.gamma, args.num_ensembles = list(net), B, 1e-3, 0.98, E
    agent = AgentModSAC(args.net_dims, S, A, gpu_id=0, args=args)
    th.set_grad_enabled(True)
    st = ModSacStepper(net, S, A, E, 1e-3, 0.98, float(agent.soft_update_tau), float(agent.clip_grad_norm))
    st.act.load_state_dict({k: v.detach().cpu() for k, v in agent.act.state_dict().items()})
    st.act_target.load_state_dict(st.act.state_dict())
    st.cri.load_state_dict({k: v.detach().cpu() for k, v in agent.cri.state_dict().items()})
    st.cri_target.load_state_dict(st.cri.state_dict())
    st.reset_optimizers()
    batch = (th.randn(B, S, generator=g), th.randn(B, A, generator=g).tanh(), th.randn(B, generator=g), (th.rand(B, generator=g) < 0.97).float(),
             (th.rand(B, generator=g) < 0.98).float(), th.randn(B, S, generator=g))
    dbatch = tuple(x.to(dev).contiguous() for x in batch)
    objs = th.zeros(2, device=dev)
    for t in range(3):
        e_next, e_cur = th.randn(B, A, generator=g), th.randn(B, A, generator=g)
        oc, oa = st.step(batch, e_next, e_cur, update_t=t)
        agent._update_on_batch(dbatch, objs, noises=(e_next.to(dev), e_cur.to(dev)), update_t=t)
        got = objs.cpu().numpy()
        assert np.isnan(oa) == np.isnan(got[1]) == (t == 2)
        np.testing.assert_allclose(got, [oc, oa], rtol=3e-4, atol=3e-6, equal_nan=True)
        for name, mine, ref in (("act", agent.act, st.act), ("act_target", agent.act_target, st.act_target), ("cri", agent.cri, st.cri),
                                ("cri_target", agent.cri_target, st.cri_target)):
            for k, v in ref.state_dict().items():
                np.testing.assert_allclose(mine.state_dict()[k].cpu().numpy(), v.numpy(), rtol=0, atol=4e-5, err_msg=f"{name}.{k} after step {t}")
        np.testing.assert_allclose(agent.alpha_log.detach().cpu().numpy(), st.alpha_log.detach().numpy(), rtol=0, atol=1e-5)
    th.set_grad_enabled(False)


@pytest.mark.gpu
def test_modsac_update_net_loop_and_checkpoint(tmp_path):
    """AgentModSAC end to end on a GPU-resident env: rollout -> ring -> update_net (the two-time-scale rule inside the loop: about a
    third of the steps skip the actor, their nan objectives stay out of the mean as AgentBase.py:186-188) -> save / load"""
    from elegantrl_amd.agents import AgentModSAC
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config, ReplayBuffer
    N, S, A = 64, 11, 3
    args = Config(AgentModSAC, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 50, "state_dim": S, "action_dim": A,
                                           "if_discrete": False})
    args.net_dims, args.batch_size, args.horizon_len, args.repeat_times = [64, 32], 128, 16, 32.0      # cur_size counts time rows (AgentBase.py:180)
    th.manual_seed(0)
    agent = AgentModSAC(args.net_dims, S, A, gpu_id=0, args=args)
    env = SynVecEnv(N, S, A, max_step=50, gpu_id=0, seed=1)
    agent.last_state = env.reset()[0]
    buf = ReplayBuffer(max_size=4096, state_dim=S, action_dim=A, gpu_id=0, num_seqs=N)
    for _ in range(2):
        buf.update(agent.explore_env(env, 16))
    w0 = agent.act.encoder_s[0].weight.detach().clone()
    t0 = agent.act_target.encoder_s[0].weight.detach().clone()
    oc, oa = agent.update_net(buf)
    times = int(buf.cur_size * 32.0 / 128)
    assert times >= 6 and np.isfinite([oc, oa]).all()
    assert 0 < agent.update_a < times and agent._actor_step == agent.update_a          # some steps skipped the actor
    assert not th.equal(agent.act.encoder_s[0].weight, w0) and not th.equal(agent.act_target.encoder_s[0].weight, t0)
    agent.save_or_load_agent(str(tmp_path), if_save=True)
    fresh = AgentModSAC(args.net_dims, S, A, gpu_id=0, args=args)
    fresh.save_or_load_agent(str(tmp_path), if_save=False)
    assert fresh._actor_step == agent._actor_step and fresh._step == agent._step
    assert th.equal(fresh.act_target.encoder_s[0].weight, agent.act_target.encoder_s[0].weight)
    assert th.equal(fresh._actor_target_flat, agent._actor_target_flat)


def test_actor_critic_modules_match_reference_rollout_on_cpu():
    """module-level math on CPU: stored action == tanh(mean + std * eps) for the recorded eps (AgentSAC.py:179-185)."""
    from elegantrl_amd.agents.AgentSAC import ActorSAC, CriticEnsemble
    g = load("sac_small.npz")
    N, S, A, rows, B, n_upd, n_ens, h1, h2 = [int(x) for x in g["dims"]]
    act, cri = ActorSAC([h1, h2], S, A), CriticEnsemble([h1, h2], S, A, n_ens)
    _load_nets(g, 0, act, cri)
    with th.no_grad():
        a = act.get_action(th.from_numpy(g["ro_states"][0]), th.from_numpy(g["ro_eps"][0]))
        np.testing.assert_allclose(a.numpy(), g["ro_actions"][0], rtol=1e-5, atol=1e-6)
        q = cri.get_q_values(th.from_numpy(g["ro_states"][3]), th.from_numpy(g["ro_actions"][3]))
        assert q.shape == (N, n_ens)
        np.testing.assert_allclose(cri(th.from_numpy(g["ro_states"][3]), th.from_numpy(g["ro_actions"][3])).numpy(),
                                   q.mean(1, keepdim=True).numpy(), rtol=1e-6)


class _ReplayEnv:
    """replays the recorded env transitions of the golden rollout (device tensors, the reference's 5-tuple protocol)."""

    def __init__(self, g, dev):
        self.g, self.dev, self.t = g, dev, 0
        self.num_envs = g["ro_states"].shape[1]

    def step(self, action):
        g, t = self.g, self.t
        nxt = g["ro_states"][t + 1] if t + 1 < g["ro_states"].shape[0] else g["ro_last_state"]
        self.t += 1
        reward = th.from_numpy(g["ro_rewards"][t] / 0.5).to(self.dev)          # un-scale (reward_scale = 0.5)
        return (th.from_numpy(nxt).to(self.dev), reward, th.from_numpy(~g["ro_undones"][t]).to(self.dev),
                th.from_numpy(~g["ro_unmasks"][t]).to(self.dev), {})


@pytest.mark.gpu
@pytest.mark.parametrize("name", SAC_GOLDENS)
def test_sac_rollout_and_updates_replay_the_reference(name):
    from elegantrl_amd.agents import AgentSAC
    from elegantrl_amd.train import Config, ReplayBuffer
    g = load(name)
    lam = float(g["lambda_fit_cum_r"][0]) if "lambda_fit_cum_r" in g else 0.0
    N, S, A, rows, B, n_upd, n_ens, h1, h2 = [int(x) for x in g["dims"]]
    gamma, lr, max_norm, reward_scale, tau, target_entropy = [float(x) for x in g["hyper"]]
    dev = th.device("cuda:0")
    args = Config(AgentSAC, None, {"env_name": "scripted", "num_envs": N, "max_step": 100, "state_dim": S, "action_dim": A,
                                   "if_discrete": False})
    assert args.if_off_policy
    args.net_dims = [h1, h2]
    args.batch_size, args.learning_rate, args.gamma, args.reward_scale, args.soft_update_tau = B, lr, gamma, reward_scale, tau
    args.clip_grad_norm = max_norm
    args.lambda_fit_cum_r = lam
    agent = AgentSAC(args.net_dims, S, A, gpu_id=0, args=args)
    assert abs(agent.target_entropy - target_entropy) < 1e-12 and agent.lambda_fit_cum_r == lam
    _load_nets(g, 0, agent.act, agent.cri)
    agent.cri_target.load_state_dict(agent.cri.state_dict())
    with th.no_grad():
        agent.alpha_log[:] = th.from_numpy(g["alpha_log0"]).to(dev)

    # ---- row 2b: off-policy rollout (stored action = squashed action, rewards scaled, flags inverted)
    agent.last_state = th.from_numpy(g["first_state"]).to(dev)
    th.set_grad_enabled(False)
    items = agent._explore_vec_env(_ReplayEnv(g, dev), rows, noise=th.from_numpy(g["ro_eps"]).to(dev))
    for got, name in zip(items, ("ro_states", "ro_actions", "ro_rewards", "ro_undones", "ro_unmasks")):
        if got.dtype == th.bool:
            np.testing.assert_array_equal(got.cpu().numpy(), g[name])
        else:
            np.testing.assert_allclose(got.cpu().numpy(), g[name], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(agent.last_state.cpu().numpy(), g["ro_last_state"], rtol=0, atol=0)

    # ---- rows 13-16: ring write (K8), sample (K9), update_objectives
    buf = ReplayBuffer(max_size=rows + 5, state_dim=S, action_dim=A, gpu_id=0, num_seqs=N)
    buf.update(tuple(th.from_numpy(g[n]).to(dev) for n in ("ro_states", "ro_actions", "ro_rewards", "ro_undones", "ro_unmasks")))
    assert (buf.p, buf.cur_size, buf.if_full) == (rows, rows, False)
    if lam:                       # the recorded contents of buffer.cum_rewards (the reference cannot fill it for AgentSAC: make_golden.py)
        buf.cum_rewards[:] = th.from_numpy(g["cum_rewards"]).to(dev)
    th.set_grad_enabled(True)
    for t in range(n_upd):
        oc, oa = agent.update_objectives(buf, t, ids=th.from_numpy(g["ids"][t]).to(dev),
                                         noises=(th.from_numpy(g["eps_next"][t]).to(dev), th.from_numpy(g["eps_cur"][t]).to(dev)))
        np.testing.assert_allclose([oc, oa], g["objs"][t], rtol=2e-4, atol=2e-6)
        for prefix, net in ((f"act{t + 1}", agent.act), (f"cri{t + 1}", agent.cri), (f"crit{t + 1}", agent.cri_target)):
            for k, v in net.state_dict().items():
                np.testing.assert_allclose(v.cpu().numpy(), g[f"{prefix}.{k}"], rtol=0, atol=3e-5, err_msg=f"{prefix}.{k}")
        np.testing.assert_allclose(agent.alpha_log.detach().cpu().numpy(), g[f"alpha_log{t + 1}"], rtol=0, atol=1e-5)
    th.set_grad_enabled(False)


@pytest.mark.gpu
@pytest.mark.parametrize("lambda_fit", [0.0, 0.5], ids=["default", "lambda_fit_cum_r"])
def test_sac_update_net_loop_on_hopper_shaped_ring(lambda_fit):
    """config-3 shapes end to end: GPU-resident synthetic env (S=11, A=3) -> off-policy rollout -> ring -> update_net; with
    lambda_fit_cum_r the loop first refreshes the newest rows' n-step returns (AgentBase.py:176-177, erl_cum_rewards_f32)."""
    from elegantrl_amd.agents import AgentSAC
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config, ReplayBuffer
    N, S, A, H = 64, 11, 3, 32
    args = Config(AgentSAC, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 50, "state_dim": S, "action_dim": A,
                                        "if_discrete": False})
    args.net_dims, args.batch_size, args.repeat_times = [64, 64], 256, 16.0   # update_times = int(cur_size * 16 / 256)
    args.lambda_fit_cum_r = lambda_fit
    th.manual_seed(0)
    agent = AgentSAC(args.net_dims, S, A, gpu_id=0, args=args)
    env = SynVecEnv(N, S, A, max_step=50, gpu_id=0, seed=3)
    agent.last_state = env.reset()[0]
    buf = ReplayBuffer(max_size=1000, state_dim=S, action_dim=A, gpu_id=0, num_seqs=N)
    th.set_grad_enabled(False)
    w0 = [p.detach().clone() for p in agent.act.parameters()]
    for _ in range(2):
        items = agent.explore_env(env, H)
        assert [tuple(x.shape) for x in items] == [(H, N, S), (H, N, A), (H, N), (H, N), (H, N)]
        assert items[1].abs().max() <= 1.0 and items[3].dtype == th.bool
        buf.update(items)
        if lambda_fit:
            buf.cum_rewards[buf.p - H:buf.p] = float("nan")          # the rows update_net must refresh before it samples
        oc, oa = agent.update_net(buf)
        assert np.isfinite([oc, oa]).all()
        if lambda_fit:             # the newest add_size rows were refreshed: discounted sums of finite rewards
            assert bool(th.isfinite(buf.cum_rewards[:buf.cur_size]).all())
    assert buf.cur_size == 2 * H and int(buf.cur_size * args.repeat_times / args.batch_size) == 4
    assert any(not th.equal(a, b) for a, b in zip(w0, agent.act.parameters()))


@pytest.mark.gpu
def test_train_agent_sac_pendulum_learns(tmp_path):
    """config-3 style loop through train_agent: off-policy rollout -> ring -> erl_sac_update_f32; files like the reference's
    run.py writes them, and the policy improves on Pendulum (a few thousand SAC steps)."""
    import os
    from elegantrl_amd import train_agent
    from elegantrl_amd.agents import AgentSAC
    from elegantrl_amd.envs import PendulumVecEnv
    from elegantrl_amd.train import Config
    args = Config(AgentSAC, PendulumVecEnv, {"env_name": "Pendulum-v1", "num_envs": 64, "max_step": 200, "state_dim": 3,
                                             "action_dim": 1, "if_discrete": False})
    args.net_dims = [128, 64]
    args.horizon_len, args.batch_size, args.repeat_times = 25, 256, 256.0 / 25 * 2   # ~2 SAC steps per collected row-block... cur_size based
    args.buffer_size = 20_000
    args.gamma, args.reward_scale, args.learning_rate = 0.97, 2 ** -2, 3e-4
    args.break_step, args.eval_per_step, args.eval_times = 25 * 60, 25 * 20, 8
    args.cwd, args.gpu_id, args.random_seed = str(tmp_path / "run"), 0, 0
    train_agent(args, if_single_process=True)
    files = os.listdir(args.cwd)
    assert "act.pth" in files and "cri.pth" in files and "cri_target.pth" in files and "recorder.npy" in files
    rec = np.load(os.path.join(args.cwd, "recorder.npy"))
    assert np.isfinite(rec[:, :4]).all()
    assert rec[-1, 1] > rec[0, 1] + 50, f"no learning progress: first eval {rec[0, 1]:.1f}, last {rec[-1, 1]:.1f}"
    actor = th.load(os.path.join(args.cwd, "act.pth"), weights_only=False)
    assert actor(th.zeros((2, 3), device="cuda:0")).shape == (2, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("S,A,hidden,E,B", [(5, 1, (32,), 1, 37), (11, 3, (256, 256), 4, 256), (17, 6, (64, 48, 32), 2, 130),
                                           (3, 1, (128, 64), 4, 64)])
def test_sac_update_matches_torch_restatement_on_other_shapes(S, A, hidden, E, B):
    """erl_sac_update_f32 vs oracle/sac_torch.py (itself pinned to the reference golden) on random batches for network
    shapes the golden does not cover: 1..3 hidden layers, 1..4 ensembles, scalar actions, the demos' [256, 256]."""
    from elegantrl_amd import ops
    from oracle.sac_torch import SacStepper
    dev = th.device("cuda:0")
    th.set_grad_enabled(True)                                   # train_agent() (previous test) leaves autograd switched off
    th.manual_seed(S * 100 + E)
    st = SacStepper(list(hidden), S, A, E, lr=1e-3, gamma=0.97, tau=5e-3, max_norm=3.0)
    with th.no_grad():                                          # make target != critic and log_std span the clamp range
        for p in st.cri_target.parameters():
            p.add_(0.05 * th.randn_like(p))
        st.act.net_a[0].bias[A:] = th.linspace(-18.0, 3.0, A) if A > 1 else th.tensor([0.3])
    spec = ops.SacSpec(S, A, hidden, E)

    def flat(module, slices):
        sd = dict(module.named_parameters())
        return th.cat([sd[name].detach().reshape(-1) for name, _, _ in slices]).to(dev).contiguous()

    pa, pc, pt = flat(st.act, spec.actor_slices()), flat(st.cri, spec.critic_slices()), flat(st.cri_target, spec.critic_slices())
    assert pa.numel() == spec.actor_count and pc.numel() == spec.critic_count
    alpha = st.alpha_log.detach().clone().to(dev)
    mom = [th.zeros_like(pa), th.zeros_like(pa), th.zeros_like(pc), th.zeros_like(pc), th.zeros(1, device=dev), th.zeros(1, device=dev)]
    objs = th.zeros(2, device=dev)
    for step in range(1, 4):
        batch = (th.randn(B, S), th.randn(B, A).tanh(), th.randn(B), (th.rand(B) > 0.1).float(), (th.rand(B) > 0.1).float(),
                 th.randn(B, S))
        e_next, e_cur = th.randn(B, A), th.randn(B, A)
        ref = st.step(batch, e_next, e_cur)
        ops.sac_update(spec, pa, pc, pt, alpha, mom, [x.to(dev).contiguous() for x in batch], step, gamma=0.97,
                       target_entropy=st.target_entropy, tau=5e-3, lr=1e-3, max_norm=3.0, objs_out=objs,
                       noises=(e_next.to(dev), e_cur.to(dev)))
        np.testing.assert_allclose(objs.cpu().numpy(), ref, rtol=3e-4, atol=3e-6)
        # Adam turns a gradient g into a step lr * g / (|g| + 1e-8): for the handful of weights whose gradient is at the
        # fp32 noise floor (|g| ~ 1e-9, summation order of the MFMA GEMMs vs torch) the step can differ by up to lr per update.
        # Bar: >= 99.5 % of every block within 5e-5, no element further than the accumulated Adam step bound.
        for got, module, slices in ((pa, st.act, spec.actor_slices()), (pc, st.cri, spec.critic_slices()),
                                    (pt, st.cri_target, spec.critic_slices())):
            diff = np.abs(got.cpu().numpy() - flat(module, slices).cpu().numpy())
            assert (diff <= 5e-5).mean() >= 0.995, f"{(diff > 5e-5).mean():.4%} of the block is off"
            assert diff.max() <= 2.2 * step * 1e-3
        np.testing.assert_allclose(alpha.cpu().numpy(), st.alpha_log.detach().numpy(), rtol=0, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("B,E", [(256, 4), (100, 2), (64, 8)])
def test_sac_split_training_pass_agrees_with_the_unsplit_one(B, E, monkeypatch):
    """round 6: the critic's training pass splits every 256-wide decoder over four workgroups that exchange their shares of q INSIDE the
    launch (CriticArgs::qx) and add their shares of dEnc by last-arriver (ERL_SAC_TRAIN_SPLIT=0: one workgroup per tile and decoder, as
    before).  Same inputs, three steps each: objectives, critic / target / actor weights and the returned td errors agree to summation
    order (the shares of q and dEnc are added in slice order instead of feature order), repeated launches of the split form are
    bit-identical to each other (fixed orders everywhere), no exchange wait timed out."""
    from elegantrl_amd import _hip, ops
    dev = th.device("cuda:0")
    S, A, hidden = 11, 3, (256, 256)
    spec = ops.SacSpec(S, A, hidden, E)
    g = th.Generator(device=dev).manual_seed(B + E)
    init = [0.05 * th.randn(n, device=dev, generator=g) for n in (spec.actor_count, spec.critic_count, spec.critic_count)]
    batches = [((th.randn((B, S), device=dev, generator=g), th.randn((B, A), device=dev, generator=g).tanh(), th.randn(B, device=dev, generator=g),
                 (th.rand(B, device=dev, generator=g) > 0.1).float(), (th.rand(B, device=dev, generator=g) > 0.1).float(),
                 th.randn((B, S), device=dev, generator=g)), th.randn((B, A), device=dev, generator=g), th.randn((B, A), device=dev, generator=g))
               for _ in range(3)]

    def run(train_split):
        monkeypatch.setenv("ERL_SAC_TRAIN_SPLIT", "1" if train_split else "0")
        pa, pc, pt = [x.clone() for x in init]
        alpha = th.full((1,), -1.0, device=dev)
        mom = [th.zeros_like(pa), th.zeros_like(pa), th.zeros_like(pc), th.zeros_like(pc), th.zeros(1, device=dev), th.zeros(1, device=dev)]
        objs, td, out = th.zeros(2, device=dev), th.zeros(B, device=dev), []
        for step, (batch, e_next, e_cur) in enumerate(batches, 1):
            ops.sac_update(spec, pa, pc, pt, alpha, mom, list(batch), step, gamma=0.97, target_entropy=-float(A), tau=5e-3, lr=1e-3, max_norm=3.0,
                           objs_out=objs, noises=(e_next, e_cur), td_error_out=td)
            out.append((objs.clone(), td.clone()))
        th.cuda.synchronize()
        _hip.check_async_faults()
        return pa, pc, pt, out

    a, b, a2 = run(True), run(False), run(True)
    for x, y in zip(a[:3], a2[:3]):
        assert th.equal(x, y)                                   # the split form is deterministic
    for (oa, ta), (ob, tb) in zip(a[3], b[3]):
        np.testing.assert_allclose(oa.cpu().numpy(), ob.cpu().numpy(), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(ta.cpu().numpy(), tb.cpu().numpy(), rtol=2e-4, atol=1e-6)
    for x, y in zip(a[:3], b[:3]):
        d = (x - y).abs().cpu().numpy()
        assert (d <= 2e-5).mean() >= 0.995 and d.max() <= 6.6e-3      # (Adam's step for a gradient at the fp32 noise floor: see above)


@pytest.mark.gpu
@pytest.mark.parametrize("B,E,hidden", [(256, 4, (256, 256)), (100, 2, (256, 256)), (64, 8, (256, 256)), (130, 5, (64, 48))])
def test_sac_weight_gradient_table_forms_are_bit_identical(B, E, hidden, monkeypatch):
    """dw_table_kernel (csrc/sac_fused.hip), ERL_SAC_DW: 0 (default) requests a summed operand's matrices (the encoder's gradient: one dEnc
    per decoder) in one round trip; 1 is the earlier form (one round trip per further matrix); 2 / 3 the same two with the temperature's
    Adam step / clamp in a workgroup of their own behind the last tile; 4 a deeper instantiation; 5 / 6 = 0 / 1 with the tiles handed out in
    XCD-contiguous order.  Same loads, same order of every sum:
    weights, moments, temperature and objectives must be the SAME BITS after three steps."""
    from elegantrl_amd import _hip, ops
    dev = th.device("cuda:0")
    S, A = 11, 3
    spec = ops.SacSpec(S, A, hidden, E)
    g = th.Generator(device=dev).manual_seed(7 * B + E)
    init = [0.05 * th.randn(n, device=dev, generator=g) for n in (spec.actor_count, spec.critic_count, spec.critic_count)]
    batches = [((th.randn((B, S), device=dev, generator=g), th.randn((B, A), device=dev, generator=g).tanh(), th.randn(B, device=dev, generator=g),
                 (th.rand(B, device=dev, generator=g) > 0.1).float(), (th.rand(B, device=dev, generator=g) > 0.1).float(),
                 th.randn((B, S), device=dev, generator=g)), th.randn((B, A), device=dev, generator=g), th.randn((B, A), device=dev, generator=g))
               for _ in range(3)]

    def run(form):
        monkeypatch.setenv("ERL_SAC_DW", form)
        pa, pc, pt = [x.clone() for x in init]
        alpha = th.full((1,), 1.9, device=dev)                   # (close to the clamp's upper edge: the clamp has work to do)
        mom = [th.zeros_like(pa), th.zeros_like(pa), th.zeros_like(pc), th.zeros_like(pc), th.zeros(1, device=dev), th.zeros(1, device=dev)]
        objs, out = th.zeros(2, device=dev), []
        for step, (batch, e_next, e_cur) in enumerate(batches, 1):
            ops.sac_update(spec, pa, pc, pt, alpha, mom, list(batch), step, gamma=0.97, target_entropy=-float(A), tau=5e-3, lr=1e-3, max_norm=3.0,
                           objs_out=objs, noises=(e_next, e_cur))
            out.append(objs.clone())
        th.cuda.synchronize()
        _hip.check_async_faults()
        return [pa, pc, pt, alpha] + mom + out

    ref = run("1")
    for form in ("0", "2", "3", "4", "5", "6"):
        for x, y in zip(ref, run(form)):
            assert th.equal(x, y), f"ERL_SAC_DW={form}"


@pytest.mark.gpu
@pytest.mark.parametrize("B,E", [(256, 4), (100, 2), (4096, 2), (37, 1)])
def test_sac_head_share_meeting_forms_are_bit_identical(B, E, monkeypatch):
    """The split actor forward's slices meet either by last arriver (shares to memory, acknowledged, an arrival counter, the last one fetches:
    ERL_SAC_YX=0) or by owner (round 6: {share, nonce} granules, the tile's last-dispatched slice polls and adds in slice order: the default).
    Same sums in the same order: actions, log-probs and everything downstream -- weights, moments, temperature, objectives -- are the SAME BITS
    after three steps; no wait timed out (B = 4096: 2048 workgroups, far more than are resident at once)."""
    from elegantrl_amd import _hip, ops
    dev = th.device("cuda:0")
    S, A, hidden = 11, 3, (256, 256)
    spec = ops.SacSpec(S, A, hidden, E)
    g = th.Generator(device=dev).manual_seed(11 * B + E)
    init = [0.05 * th.randn(n, device=dev, generator=g) for n in (spec.actor_count, spec.critic_count, spec.critic_count)]
    batches = [((th.randn((B, S), device=dev, generator=g), th.randn((B, A), device=dev, generator=g).tanh(), th.randn(B, device=dev, generator=g),
                 (th.rand(B, device=dev, generator=g) > 0.1).float(), (th.rand(B, device=dev, generator=g) > 0.1).float(),
                 th.randn((B, S), device=dev, generator=g)), th.randn((B, A), device=dev, generator=g), th.randn((B, A), device=dev, generator=g))
               for _ in range(3)]

    def run(form):
        monkeypatch.setenv("ERL_SAC_YX", form)
        pa, pc, pt = [x.clone() for x in init]
        alpha = th.full((1,), -1.0, device=dev)
        mom = [th.zeros_like(pa), th.zeros_like(pa), th.zeros_like(pc), th.zeros_like(pc), th.zeros(1, device=dev), th.zeros(1, device=dev)]
        objs, out = th.zeros(2, device=dev), []
        for step, (batch, e_next, e_cur) in enumerate(batches, 1):
            ops.sac_update(spec, pa, pc, pt, alpha, mom, list(batch), step, gamma=0.97, target_entropy=-float(A), tau=5e-3, lr=1e-3, max_norm=3.0,
                           objs_out=objs, noises=(e_next, e_cur))
            out.append(objs.clone())
        th.cuda.synchronize()
        _hip.check_async_faults()
        return [pa, pc, pt, alpha] + mom + out

    ref = run("0")
    for x, y in zip(ref, run("1")):
        assert th.equal(x, y)
    for x, y in zip(ref, run("1")):                               # (twice: the nonce moves on, older granules are never mistaken for new ones)
        assert th.equal(x, y)


@pytest.mark.gpu
def test_sac_update_with_importance_weights_and_td_errors():
    """prioritised replay through the SAC step (AgentSAC.py:58-62): obj_critic = mean(td_error * is_weight), td_error comes back
    per sample; against oracle/sac_torch.py with the same weights."""
    from elegantrl_amd import ops
    from oracle.sac_torch import SacStepper
    dev = th.device("cuda:0")
    th.set_grad_enabled(True)
    th.manual_seed(77)
    S, A, hidden, E, B = 11, 3, (64, 32), 4, 96
    st = SacStepper(list(hidden), S, A, E, lr=1e-3, gamma=0.97, tau=5e-3, max_norm=3.0)
    spec = ops.SacSpec(S, A, hidden, E)

    def flat(module, slices):
        sd = dict(module.named_parameters())
        return th.cat([sd[name].detach().reshape(-1) for name, _, _ in slices]).to(dev).contiguous()

    pa, pc, pt = flat(st.act, spec.actor_slices()), flat(st.cri, spec.critic_slices()), flat(st.cri_target, spec.critic_slices())
    alpha = st.alpha_log.detach().clone().to(dev)
    mom = [th.zeros_like(pa), th.zeros_like(pa), th.zeros_like(pc), th.zeros_like(pc), th.zeros(1, device=dev), th.zeros(1, device=dev)]
    objs, td = th.zeros(2, device=dev), th.zeros(B, device=dev)
    for step in range(1, 3):
        batch = (th.randn(B, S), th.randn(B, A).tanh(), th.randn(B), (th.rand(B) > 0.1).float(), (th.rand(B) > 0.1).float(),
                 th.randn(B, S))
        w = th.rand(B) * 0.9 + 0.1
        e_next, e_cur = th.randn(B, A), th.randn(B, A)
        ref = st.step(batch, e_next, e_cur, is_weight=w)
        ops.sac_update(spec, pa, pc, pt, alpha, mom, [x.to(dev).contiguous() for x in batch], step, gamma=0.97,
                       target_entropy=st.target_entropy, tau=5e-3, lr=1e-3, max_norm=3.0, objs_out=objs,
                       noises=(e_next.to(dev), e_cur.to(dev)), is_weight=w.to(dev), td_error_out=td)
        np.testing.assert_allclose(objs.cpu().numpy(), ref, rtol=3e-4, atol=3e-6)
        np.testing.assert_allclose(td.cpu().numpy(), st.td_error.numpy(), rtol=3e-4, atol=1e-6)
        diff = np.abs(pc.cpu().numpy() - flat(st.cri, spec.critic_slices()).cpu().numpy())
        assert (diff <= 5e-5).mean() >= 0.995 and diff.max() <= 2.2 * step * 1e-3


@pytest.mark.gpu
def test_agent_sac_with_prioritised_replay_runs_and_updates_priorities():
    """AgentSAC(if_use_per=True) + ReplayBuffer(if_use_per=True): update_net draws prioritised batches and writes the td errors
    back -- the sampled leaves leave the initial priority 10."""
    from elegantrl_amd.agents import AgentSAC
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config, ReplayBuffer
    dev = th.device("cuda:0")
    N, S, A, H = 8, 11, 3, 16
    args = Config(AgentSAC, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 50, "state_dim": S, "action_dim": A,
                                        "if_discrete": False})
    args.net_dims, args.horizon_len, args.batch_size, args.if_use_per = [64, 32], H, 64, True
    args.repeat_times = 4 * 64 / 64.0
    agent = AgentSAC(args.net_dims, S, A, gpu_id=0, args=args)
    env = SynVecEnv(N, S, A, max_step=50, gpu_id=0, seed=1)
    agent.last_state = env.reset()[0]
    buf = ReplayBuffer(max_size=64, state_dim=S, action_dim=A, gpu_id=0, num_seqs=N, if_use_per=True, args=args)
    buf.update(agent.explore_env(env, H))
    buf.update(agent.explore_env(env, H))
    leaves0 = buf.sum_trees.sum.view(N, -1)[:, buf.sum_trees.leaves:buf.sum_trees.leaves + buf.cur_size].clone()
    assert th.all(leaves0 == 10.0)
    objs = agent.update_net(buf)
    assert all(np.isfinite(o) for o in objs)
    leaves1 = buf.sum_trees.sum.view(N, -1)[:, buf.sum_trees.leaves:buf.sum_trees.leaves + buf.cur_size]
    changed = (leaves1 != 10.0)
    assert int(changed.sum()) > 20 and float(leaves1.max()) <= 10.0 and float(leaves1.min()) > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("net,N", [((256, 256), 64), ((64, 48, 32), 50)], ids=["one-launch-form", "layered-form"])
def test_sac_rollout_rows_written_by_the_explore_launch(net, N):
    """the off-policy rollout (AgentBase.py:130-170) lets the explore kernel write `actions[t]` AND `states[t]` into the buffer rows:
    the same five tensors, bit for bit, as the loop that copies them (explore_action without `out` / `out_state`)."""
    from elegantrl_amd.agents import AgentSAC
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config
    S, A, H = 11, 3, 9

    def run(rows: bool):
        args = Config(AgentSAC, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 5, "state_dim": S, "action_dim": A, "if_discrete": False})
        args.net_dims, args.random_seed, args.fused_rollout = list(net), 3, False        # (the per-step loop is what is under test)
        th.manual_seed(5)
        agent = AgentSAC(args.net_dims, S, A, gpu_id=0, args=args)
        if not rows:                                            # the signature without the row arguments: the loop copies
            inner = agent.explore_action
            agent.explore_action = lambda state, noise=None: inner(state, noise)
        env = SynVecEnv(N, S, A, max_step=5, gpu_id=0, seed=1)
        agent.last_state = env.reset()[0]
        noise = th.randn((H, N, A), device="cuda:0", generator=th.Generator(device="cuda:0").manual_seed(2))
        return agent._explore_vec_env(env, H, noise=noise), agent.last_state.clone()
    (a, la), (b, lb) = run(True), run(False)
    for name, x, y in zip(("states", "actions", "rewards", "undones", "unmasks"), a, b):
        assert x.shape == y.shape and th.equal(x, y), name
    assert th.equal(la, lb)
    assert a[0].abs().sum().item() > 0 and not th.equal(a[0][0], a[0][-1])


@pytest.mark.gpu
def test_sac_update_net_draws_its_sample_ids_ahead_from_the_same_distribution():
    """AgentSAC.update_net draws the ids of all its steps with one th.randint (sample_ids_ahead, default on) instead of one per step:
    both loops run, stay finite and visit the whole ring (a statistical statement: the streams differ by construction)."""
    from elegantrl_amd.agents import AgentSAC
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config, ReplayBuffer
    N, S, A, H = 8, 11, 3, 32
    for ahead in (True, False):
        args = Config(AgentSAC, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 50, "state_dim": S, "action_dim": A, "if_discrete": False})
        args.net_dims, args.horizon_len, args.batch_size, args.sample_ids_ahead = [64, 32], H, 64, ahead
        args.sample_in_step = False                         # (the spy below watches buffer.sample)
        args.repeat_times = 8.0
        agent = AgentSAC(args.net_dims, S, A, gpu_id=0, args=args)
        assert agent.sample_ids_ahead is ahead
        env = SynVecEnv(N, S, A, max_step=50, gpu_id=0, seed=1)
        agent.last_state = env.reset()[0]
        buf = ReplayBuffer(max_size=H, state_dim=S, action_dim=A, gpu_id=0, num_seqs=N, args=args)
        buf.update(agent.explore_env(env, H))
        seen = []
        inner = buf.sample

        def spy(batch_size, ids=None, reuse=False):
            out = inner(batch_size, ids=ids, reuse=reuse)
            seen.append((ids is not None, buf.ids0.clone(), buf.ids1.clone()))
            return out
        buf.sample = spy
        oc, oa = agent.update_net(buf)
        assert np.isfinite(oc) and np.isfinite(oa)
        assert len(seen) == int(buf.cur_size * 8.0 / 64) == 4 and all(flag is ahead for flag, _, _ in seen)
        ids0 = th.cat([s[1] for s in seen]).cpu().numpy()
        ids1 = th.cat([s[2] for s in seen]).cpu().numpy()
        assert ids0.min() >= 0 and ids0.max() <= buf.cur_size - 2 and ids1.min() >= 0 and ids1.max() <= N - 1


@pytest.mark.gpu
@pytest.mark.parametrize("N,S,A,net,max_step,H,scale", [(64, 11, 3, (256, 256), 5, 9, 1.0), (50, 17, 6, (64, 48), 4, 7, 0.25),
                                                        (100, 56, 8, (128, 256), 1000, 6, 2.0), (16, 3, 1, (256, 64), 3, 12, 1.0)])
@pytest.mark.parametrize("inject", [True, False], ids=["injected-noise", "philox"])
def test_sac_persistent_rollout_is_bit_identical_to_the_per_step_loop(N, S, A, net, max_step, H, scale, inject):
    """erl_sac_rollout_synenv_f32 (one launch per explore_env) against the loop it replaces (explore launch + env launch per step,
    AgentBase.py:130-170): states, actions, rewards, undones, unmasks, the final state and the env's counters, over two consecutive
    rollouts (state / counters / rng counter carry over), with resets (max_step < H), N not a multiple of the 16-env tile, reward scaling."""
    from elegantrl_amd.agents import AgentSAC
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config

    def make(fused):
        args = Config(AgentSAC, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": max_step, "state_dim": S, "action_dim": A,
                                            "if_discrete": False})
        args.net_dims, args.random_seed, args.reward_scale, args.fused_rollout = list(net), 3, scale, fused
        th.manual_seed(5)
        agent = AgentSAC(args.net_dims, S, A, gpu_id=0, args=args)
        env = SynVecEnv(N, S, A, max_step=max_step, gpu_id=0, seed=1)
        agent.last_state = env.reset()[0]
        return agent, env
    (fa, fe), (pa, pe) = make(True), make(False)
    assert th.equal(fa._actor_flat, pa._actor_flat) and th.equal(fe.state, pe.state)
    launches = []
    inner = fe.fused_rollout_offpolicy
    fe.fused_rollout_offpolicy = lambda *a, **k: (launches.append(1), inner(*a, **k))[1]
    g = th.Generator(device="cuda:0").manual_seed(2)
    for it in range(2):
        noise = th.randn((H, N, A), device="cuda:0", generator=g) if inject else None
        f_items = fa._explore_vec_env(fe, H, noise=noise)
        p_items = pa._explore_vec_env(pe, H, noise=noise)
        assert len(launches) == it + 1                                        # the one-launch path really ran on one side only
        for name, x, y in zip(("states", "actions", "rewards", "undones", "unmasks"), f_items, p_items):
            assert x.dtype == y.dtype and x.shape == y.shape, name
            assert th.equal(x, y), f"{name} differs at rollout {it}: {(x != y).sum().item()} elements"
        assert th.equal(fa.last_state, pa.last_state) and th.equal(fe.state, pe.state)
        assert th.equal(fe.step_count, pe.step_count) and th.equal(fe.episode, pe.episode)
        assert fa.rng_counter == pa.rng_counter
        if max_step < H:
            assert (~f_items[4]).any(), "the case is meant to contain truncations"
    # somebody moves the env between two rollouts: the agent's last_state wins (it is copied into the env's live buffer), as in the loop
    fe.state.add_(1.0)
    fe.state_epoch += 1
    pe.state.add_(1.0)
    pe.state_epoch += 1
    f_items, p_items = fa._explore_vec_env(fe, 3), pa._explore_vec_env(pe, 3)
    assert th.equal(f_items[1][0], p_items[1][0])                             # the first actions come from the agent's last_state either way


@pytest.mark.gpu
@pytest.mark.parametrize("N,net,max_step,H,scale", [(64, (256, 256), 7, 20, 1.0), (50, (64, 48), 200, 9, 0.5), (4096, (128, 64), 5, 6, 1.0)])
@pytest.mark.parametrize("inject", [True, False], ids=["injected-noise", "philox"])
def test_sac_persistent_rollout_on_pendulum_is_bit_identical_to_the_per_step_loop(N, net, max_step, H, scale, inject):
    """erl_sac_rollout_pendulum_f32 against explore launch + erl_pendulum_step_f32 per step: the five tensors, the final observation, theta /
    theta_dot and the counters, two consecutive rollouts, truncation resets inside the horizon."""
    from elegantrl_amd.agents import AgentSAC
    from elegantrl_amd.envs import PendulumVecEnv
    from elegantrl_amd.train import Config

    def make(fused):
        args = Config(AgentSAC, PendulumVecEnv, {"env_name": "Pendulum-v1", "num_envs": N, "max_step": max_step, "state_dim": 3, "action_dim": 1,
                                                 "if_discrete": False})
        args.net_dims, args.random_seed, args.reward_scale, args.fused_rollout = list(net), 3, scale, fused
        th.manual_seed(5)
        agent = AgentSAC(args.net_dims, 3, 1, gpu_id=0, args=args)
        env = PendulumVecEnv(N, max_step=max_step, gpu_id=0, seed=1)
        agent.last_state = env.reset()[0]
        return agent, env
    (fa, fe), (pa, pe) = make(True), make(False)
    launches = []
    inner = fe.fused_rollout_offpolicy
    fe.fused_rollout_offpolicy = lambda *a, **k: (launches.append(1), inner(*a, **k))[1]
    g = th.Generator(device="cuda:0").manual_seed(2)
    for it in range(2):
        noise = th.randn((H, N, 1), device="cuda:0", generator=g) if inject else None
        f_items = fa._explore_vec_env(fe, H, noise=noise)
        p_items = pa._explore_vec_env(pe, H, noise=noise)
        assert len(launches) == it + 1
        for name, x, y in zip(("states", "actions", "rewards", "undones", "unmasks"), f_items, p_items):
            assert x.dtype == y.dtype and x.shape == y.shape, name
            assert th.equal(x, y), f"{name} differs at rollout {it}: {(x != y).sum().item()} elements"
        assert th.equal(fa.last_state, pa.last_state) and th.equal(fe.state, pe.state) and th.equal(fe.phys, pe.phys)
        assert th.equal(fe.step_count, pe.step_count) and th.equal(fe.episode, pe.episode)
        if max_step < H:
            assert (~f_items[4]).any()


@pytest.mark.gpu
@pytest.mark.parametrize("net", [(256, 256), (64, 48), (64, 48, 32)], ids=["fused-split", "fused", "layered"])
def test_sac_update_net_with_the_sample_inside_the_step_is_bit_identical(net):
    """AgentSAC.update_net hands ring + ids to the step (erl_sac_update_ring_f32: the gather of ReplayBuffer.sample rides in the fused step's
    first launch, or runs as erl_replay_sample_f32 in front of the layered step) instead of sampling first: same weights, moments, targets,
    temperature, objectives and the same staged batch / ids0 / ids1, bit for bit."""
    from elegantrl_amd.agents import AgentSAC
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config, ReplayBuffer
    N, S, A, H = 16, 11, 3, 40

    def run(in_step, interleaved=True):
        args = Config(AgentSAC, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 50, "state_dim": S, "action_dim": A, "if_discrete": False})
        args.net_dims, args.horizon_len, args.batch_size, args.sample_in_step, args.random_seed = list(net), H, 256, in_step, 3
        args.replay_interleaved = interleaved
        args.repeat_times = 32.0                            # update_times = int(cur_size * repeat_times / batch_size): 5, then 10
        th.manual_seed(5)
        agent = AgentSAC(args.net_dims, S, A, gpu_id=0, args=args)
        assert agent.sample_in_step is in_step
        env = SynVecEnv(N, S, A, max_step=50, gpu_id=0, seed=1)
        agent.last_state = env.reset()[0]
        buf = ReplayBuffer(max_size=2 * H, state_dim=S, action_dim=A, gpu_id=0, num_seqs=N, args=args)
        buf.update(agent.explore_env(env, H))
        th.manual_seed(9)                                   # the same id draws on both sides
        out = [agent.update_net(buf)]
        buf.update(agent.explore_env(env, H))               # the ring wraps; a second update on the full ring
        out.append(agent.update_net(buf))
        return agent, buf, out
    (a, ba, oa), (b, bb, ob) = run(True), run(False)
    assert oa == ob and all(np.isfinite(x) for pair in oa for x in pair)
    for name in ("_actor_flat", "_critic_flat", "_target_flat", "alpha_log"):
        assert th.equal(getattr(a, name), getattr(b, name)), name
    for x, y in zip((a.act_optimizer.exp_avg, a.cri_optimizer.exp_avg_sq), (b.act_optimizer.exp_avg, b.cri_optimizer.exp_avg_sq)):
        assert th.equal(x, y)
    assert th.equal(ba.ids0, bb.ids0) and th.equal(ba.ids1, bb.ids1)
    for x, y in zip(ba._stage.out, bb._stage.out):          # the last staged batch
        assert th.equal(x, y)
    assert a._step == b._step == 15
    # ... and the same with the whole loop from ONE C call against one call per step (args.update_loop_in_c; the default above is the loop)
    def run_per_step():
        import os
        os.environ["ERL_SAC_LOOP_IN_C"] = "0"
        try:
            return run(True)
        finally:
            del os.environ["ERL_SAC_LOOP_IN_C"]
    e, be, oe = run_per_step()
    assert a.update_loop_in_c and not e.update_loop_in_c and oe == oa and e._step == 15
    for name in ("_actor_flat", "_critic_flat", "_target_flat", "alpha_log"):
        assert th.equal(getattr(a, name), getattr(e, name)), name
    assert th.equal(ba.ids0, be.ids0) and th.equal(ba.ids1, be.ids1)
    for x, y in zip(ba._stage.out, be._stage.out):
        assert th.equal(x, y)
    # ... and the same again from five PLANAR ring tensors (args.replay_interleaved = False: the reference's layout) -- the in-step gather
    # reads either layout (ErlRingSample.row_floats), the bits do not depend on it
    c, bc, oc = run(True, interleaved=False)
    assert ba._ring is not None and bc._ring is None and oc == oa
    for name in ("_actor_flat", "_critic_flat", "_target_flat", "alpha_log"):
        assert th.equal(getattr(a, name), getattr(c, name)), name
    assert th.equal(ba.ids0, bc.ids0) and th.equal(ba.ids1, bc.ids1)
    for x, y in zip(ba._stage.out, bc._stage.out):
        assert th.equal(x, y)
