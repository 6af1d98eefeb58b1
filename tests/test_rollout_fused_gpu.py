"""The persistent H-step rollout (erl_rollout_synenv_f32 / erl_rollout_pendulum_f32, csrc/rollout_fused.hip) against the
per-step path it replaces (erl_rollout_step_f32 + erl_synenv_step_f32 / erl_pendulum_step_f32 per step, then the torch
`rewards *= reward_scale` / `logical_not` ops): the six rollout buffers, the final state and the env's counters must be
BIT-IDENTICAL under the same injected noise and under the same Philox keys -- the per-step path itself is pinned to the
reference's AgentPPO._explore_vec_env (elegantrl/agents/AgentPPO.py:87-129) by tests/test_agent_gpu.py.  The values the
fused kernel leaves for update_net are checked against the K2 value pre-pass (different summation order: tolerance)."""
import numpy as np
import pytest
import torch as th

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make(env_kind, N, S, A, net, max_step, fused, reward_scale=1.0, seed=3, lr=1e-3):
    from elegantrl_amd.agents import AgentPPO
    from elegantrl_amd.envs import PendulumVecEnv, SynVecEnv
    from elegantrl_amd.train import Config
    args = Config(AgentPPO, None, {"env_name": env_kind, "num_envs": N, "max_step": max_step, "state_dim": S, "action_dim": A,
                                   "if_discrete": False})
    args.net_dims, args.reward_scale, args.random_seed, args.learning_rate = list(net), reward_scale, 7, lr
    args.fused_rollout = fused
    th.manual_seed(seed)
    agent = AgentPPO(args.net_dims, S, A, gpu_id=0, args=args)
    with th.no_grad():                                     # non-trivial normalisation vectors and action std
        g = th.Generator(device=DEV).manual_seed(seed + 1)
        agent.act.state_avg[:] = 0.1 * th.randn(S, device=DEV, generator=g)
        agent.act.state_std[:] = 1.0 + 0.2 * th.rand(S, device=DEV, generator=g)
        agent.cri.state_avg[:] = 0.1 * th.randn(S, device=DEV, generator=g)
        agent.cri.state_std[:] = 1.0 + 0.2 * th.rand(S, device=DEV, generator=g)
        agent.act.action_std_log[:] = -0.3 + 0.1 * th.randn(A, device=DEV, generator=g)
    if env_kind == "pendulum":
        env = PendulumVecEnv(N, max_step=max_step, gpu_id=0, seed=5)
    else:
        env = SynVecEnv(N, S, A, max_step=max_step, gpu_id=0, seed=5)
    agent.last_state = env.reset()[0]
    return agent, env, args


CASES = [
    ("syn", 4096, 64, 8, (128, 128), 5, 32, 1.0),          # BASELINE config 4 shape, short episodes: truncations + resets
    ("syn", 1000, 60, 8, (128, 128), 7, 12, 0.25),         # config-5 obs width, N not a multiple of 16, reward scaling
    ("syn", 50, 17, 3, (64, 32), 4, 9, 2.0),               # unaligned state_dim (scalar load path), small net
    ("syn", 8192, 64, 8, (128, 128), 1000, 8, 1.0),        # config-5 env count: two workgroup rounds per CU
    ("pendulum", 4096, 3, 1, (128, 64), 16, 40, 0.25),     # config 2 (Pendulum, net [128, 64]), truncation resets
    ("pendulum", 100, 3, 1, (64, 64), 200, 10, 1.0),
    ("pendulum", 64, 3, 1, (128, 64), 50, 130, 1.0),       # horizon > 128: the epilogue reads its inputs back from the buffers (args.fused_gae forced on)
]


@pytest.mark.parametrize("kind,N,S,A,net,max_step,H,scale", CASES)
@pytest.mark.parametrize("inject", [True, False], ids=["injected-noise", "philox"])
def test_fused_rollout_is_bit_identical_to_the_per_step_path(kind, N, S, A, net, max_step, H, scale, inject):
    fa, fe, _ = _make(kind, N, S, A, net, max_step, True, scale)
    pa, pe, _ = _make(kind, N, S, A, net, max_step, False, scale)
    assert th.equal(fa._flat, pa._flat) and th.equal(fe.state, pe.state)
    g = th.Generator(device=DEV).manual_seed(11)
    for it in range(2):                                    # twice: the env state / counters / rng counter carry over
        noise = th.randn((H, N, A), device=DEV, generator=g) if inject else None
        f_items = fa._explore_vec_env(fe, H, noise=noise)
        p_items = pa._explore_vec_env(pe, H, noise=noise)
        assert fa._rollout_cache is not None and pa._rollout_cache is None       # the fused path really ran on one side only
        names = ("states", "actions", "logprobs", "rewards", "undones", "unmasks")
        for n, a, b in zip(names, f_items, p_items):
            assert a.dtype == b.dtype and a.shape == b.shape, n
            assert th.equal(a, b), f"{n} differs at iteration {it}: {(a != b).sum().item()} elements"
        assert f_items[4].dtype == th.bool and f_items[5].dtype == th.bool
        assert th.equal(fa.last_state, pa.last_state) and th.equal(fe.state, pe.state)
        assert th.equal(fe.step_count, pe.step_count) and th.equal(fe.episode, pe.episode)
        if kind == "pendulum":
            assert th.equal(fe.phys, pe.phys)
        assert fa.rng_counter == pa.rng_counter == (it + 1) * H
        if max_step < H:
            assert (~f_items[5]).any(), "the case is meant to contain truncations"
        # the critic's values of the visited states, as update_net's pre-pass computes them (other summation order)
        v_ref = pa.get_values(p_items[0])
        nv_ref = pa.get_values(pa.last_state)
        c = fa._rollout_cache
        np.testing.assert_allclose(c["values"].cpu().numpy(), v_ref.cpu().numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(c["next_value"].cpu().numpy(), nv_ref.cpu().numpy(), rtol=2e-5, atol=2e-5)


def test_update_net_consumes_the_values_of_the_fused_rollout_only_while_they_are_valid(monkeypatch):
    """update_net must skip the value pre-pass exactly when the buffer is the one the fused rollout produced and the critic
    has not changed since; any other buffer / a stepped or edited critic recomputes (AgentPPO.py:141-143, :219-220)."""
    N, S, A, H, B = 512, 64, 8, 8, 1024
    agent, env, args = _make("syn", N, S, A, (128, 128), 50, True)
    agent.batch_size, agent.repeat_times = B, 2 * B / H
    calls = []
    real = agent.get_values
    monkeypatch.setattr(agent, "get_values", lambda s: (calls.append(tuple(s.shape)), real(s))[1])
    items = agent.explore_env(env, H)
    agent.update_net(list(items))
    assert calls == []                                     # pre-pass and bootstrap both came from the rollout
    items = agent.explore_env(env, H)
    agent.update_net([x.clone() for x in items])           # e.g. the Learner's concatenated buffers: not the rollout's tensors
    assert calls == [(H, N, S), (N, S)]
    calls.clear()
    items = agent.explore_env(env, H)
    with th.no_grad():
        agent.cri.net[0].bias.add_(0.01)                   # critic edited after the rollout
    agent.update_net(list(items))
    assert calls == [(H, N, S), (N, S)]
    calls.clear()
    items = agent.explore_env(env, H)
    agent.last_state = agent.last_state.clone()            # run.py sets last_state: another tensor -> bootstrap is recomputed
    agent.update_net(list(items))
    assert calls == [(H, N, S), (N, S)]


def test_fused_and_per_step_training_agree(monkeypatch):
    """three PPO iterations (rollout + update) with and without the fused rollout, same Philox keys and minibatch ids: the
    rollouts stay bit-identical as long as the weights do; the weights differ only through the values' summation order."""
    N, S, A, H, B = 1024, 64, 8, 16, 4096
    fa, fe, _ = _make("syn", N, S, A, (128, 128), 20, True)
    pa, pe, _ = _make("syn", N, S, A, (128, 128), 20, False)
    for ag in (fa, pa):
        ag.batch_size, ag.repeat_times = B, 3 * B / H
    g = th.Generator(device=DEV).manual_seed(2)
    for it in range(3):
        ids = th.randint(H * N, (3, B), device=DEV, generator=g)
        fi, pi = fa.explore_env(fe, H), pa.explore_env(pe, H)
        if it == 0:
            assert all(th.equal(a, b) for a, b in zip(fi, pi))
        lf, lp = fa.update_net(list(fi), ids=ids), pa.update_net(list(pi), ids=ids)
        np.testing.assert_allclose(lf, lp, rtol=2e-3, atol=1e-5)
        np.testing.assert_allclose(fa._flat.cpu().numpy(), pa._flat.cpu().numpy(), rtol=0, atol=2e-4 * (it + 1))
    assert np.isfinite(fa._flat.cpu().numpy()).all()


def test_fused_rollout_rejects_what_it_cannot_run():
    from elegantrl_amd import _hip
    L = _hip.lib()
    assert L.erl_rollout_fused_supported(64, 128, 128, 8) == 1 and L.erl_rollout_fused_supported(3, 128, 64, 1) == 1
    assert L.erl_rollout_fused_supported(65, 128, 128, 8) == 0          # state tile is 4 k-tiles of 16
    assert L.erl_rollout_fused_supported(64, 128, 100, 8) == 0
    # an agent whose shape is outside the fused kernel's range silently keeps the per-step launches
    agent, env, _ = _make("syn", 64, 100, 4, (128, 128), 10, True)
    items = agent.explore_env(env, 4)
    assert agent._rollout_cache is None and items[0].shape == (4, 64, 100)


# ---- the rollout's epilogue (round 4): the agent's own last_state, get_advantages + its statistics in the same launch ---------
@pytest.mark.parametrize("kind,N,S,A,net,max_step,H,scale", CASES)
@pytest.mark.parametrize("vtrace", [True, False], ids=["vtrace", "alt"])
def test_rollout_epilogue_is_the_exact_gae_scan(kind, N, S, A, net, max_step, H, scale, vtrace):
    """advantages / reward sums left by the persistent rollout == erl_gae_scan_f32(EXACT) on the same buffers BIT FOR BIT (the
    scan shares its step with gae.hip: csrc/gae_step.h), the five raw sums agree to fp64 rounding, the caller's rewards / undones
    come back as explore_env returns them in the reference (untouched), and last_state is a tensor of the agent's own."""
    from elegantrl_amd import ops
    agent, env, _ = _make(kind, N, S, A, net, max_step, True, scale)
    agent.if_use_v_trace = vtrace
    agent._fused_gae_explicit = True                       # (as args.fused_gae = True: beyond 128 steps the default leaves get_advantages to the scan kernels)
    for it in range(2):
        items = agent._explore_vec_env(env, H)
        states, actions, logprobs, rewards, undones, unmasks = items
        c = agent._rollout_cache
        assert c is not None and "adv" in c
        assert agent.last_state.data_ptr() != env.state.data_ptr() and th.equal(agent.last_state, env.state)
        r0, u0 = rewards.clone(), undones.clone()
        stats = th.zeros(8, dtype=th.float64, device=DEV)
        adv, ret = ops.gae_scan(rewards.clone(), undones.clone(), unmasks, c["values"], c["next_value"], float(agent.gamma),
                                float(agent.lambda_gae_adv), use_v_trace=vtrace, mutate=True, algo="exact", stats=stats)
        assert th.equal(c["adv"], adv), f"advantages differ in {(c['adv'] != adv).sum().item()} elements"
        assert th.equal(c["ret"], ret)
        assert th.equal(rewards, r0) and th.equal(undones, u0)            # explore_env's outputs are not mutated by the epilogue
        # the epilogue leaves per-workgroup partial sums; erl_adv_stats_fold_f32 (or the update loop's first launch) folds them
        folded = ops.adv_stats_fold(c["parts"], c["n_parts"], H, N, th.full((8,), -1.0, dtype=th.float64, device=DEV))
        np.testing.assert_allclose(folded.cpu().numpy()[:5], stats.cpu().numpy()[:5], rtol=1e-12, atol=1e-9)
        assert folded[5:].abs().sum().item() == 0
        if max_step < H:
            assert (~unmasks).any()


def test_the_epilogue_is_left_out_beyond_128_steps_unless_asked_for():
    """default: horizons above 128 steps leave get_advantages to the scan kernels (cheaper there: profiles/r06_c2_fused_gae_ab.txt);
    args.fused_gae = True keeps the epilogue at any horizon."""
    agent, env, _ = _make("pendulum", 64, 3, 1, (128, 64), 50, True)
    agent._explore_vec_env(env, 130)
    assert agent._rollout_cache is not None and "adv" not in agent._rollout_cache
    agent._explore_vec_env(env, 128)
    assert "adv" in agent._rollout_cache
    agent._fused_gae_explicit = True
    agent._explore_vec_env(env, 130)
    assert "adv" in agent._rollout_cache


def test_update_net_with_the_rollout_epilogue_matches_the_separate_launches():
    """one iteration with args.fused_gae on and off (same weights, noise, minibatch ids): the buffers leave update_net with
    get_advantages' side effect applied either way (rewards[trunc] += values[trunc], undones[trunc] = False: bitwise), objectives
    and weights agree to the last bits the statistics' summation order can move."""
    N, S, A, H, B = 1024, 64, 8, 16, 4096
    out = {}
    for fused_gae in (True, False):
        agent, env, _ = _make("syn", N, S, A, (128, 128), 6, True)
        agent.fused_gae = fused_gae
        agent.batch_size, agent.repeat_times = B, 3 * B / H
        g = th.Generator(device=DEV).manual_seed(2)
        noise = th.randn((H, N, A), device=DEV, generator=g)
        ids = th.randint(H * N, (3, B), device=DEV, generator=g)
        items = agent._explore_vec_env(env, H, noise=noise)
        assert ("adv" in agent._rollout_cache) == fused_gae
        raw = [x.clone() for x in items]
        objs = agent.update_net(list(items), ids=ids)
        out[fused_gae] = (raw, [x.clone() for x in items], objs, agent._flat.clone())
    (raw_f, after_f, objs_f, w_f), (raw_s, after_s, objs_s, w_s) = out[True], out[False]
    for a, b in zip(raw_f, raw_s):
        assert th.equal(a, b)
    assert (~raw_f[5]).any()                                                # truncations present
    assert not th.equal(after_f[3], raw_f[3]) and not th.equal(after_f[4], raw_f[4])     # the side effect happened ...
    for a, b in zip(after_f, after_s):
        assert th.equal(a, b)                                               # ... identically on both paths
    np.testing.assert_allclose(objs_f, objs_s, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(w_f.cpu().numpy(), w_s.cpu().numpy(), rtol=0, atol=1e-6)


def test_last_state_copy_back_is_skipped_only_while_both_sides_are_untouched():
    """agent.last_state is the rollout kernel's own copy; the next explore_env copies it into the env's live buffer only if
    somebody changed either side in between (AgentPPO._last_state_token)."""
    agent, env, _ = _make("syn", 256, 64, 8, (128, 128), 50, True)
    agent.explore_env(env, 4)
    assert agent._last_state_token is not None
    ls = agent.last_state
    with th.no_grad():
        ls.add_(1.0)                                       # the caller edits its last_state in place: must reach the env
    items = agent.explore_env(env, 2)
    assert th.equal(items[0][0], ls)                        # the rollout started from the edited state
    agent.last_state = th.zeros_like(agent.last_state)     # run.py-style assignment of another tensor
    items = agent.explore_env(env, 2)
    assert float(items[0][0].abs().max()) == 0.0
    e0 = env.state_epoch
    agent.explore_env(env, 2)
    assert env.state_epoch == e0 + 1                       # only the rollout itself moved the env: no copy-back in between
