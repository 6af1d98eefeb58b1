"""Generic-shape MLP path (erl_mlpn_*: any depth / width; own MFMA GEMMs with fused epilogues) against the
depth-generic numpy oracle, and AgentPPO end to end on the reference's larger demo shapes (256, 128) / (256, 128, 64)
(examples/demo_A2C_PPO.py:117,171).  Same bars as the fused kernels: rtol 1e-4 vs the fp32 oracle."""
import numpy as np
import pytest
import torch as th

from oracle import ppo_numpy as O
from tests.test_kernels_gpu import cu, flat_params, oracle_flat_grads, ppo_case

pytestmark = pytest.mark.gpu

NET_SHAPES = [(24, (256, 128), 6), (17, (256, 128, 64), 5), (8, (96,), 3), (64, (128, 128), 8), (5, (32, 48, 64, 16), 1)]


@pytest.fixture(scope="module")
def ops():
    from elegantrl_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def dev():
    return th.device("cuda:0")


def random_net_n(rng, dims, with_std):
    ws, bs = [], []
    for i, o in zip(dims[:-1], dims[1:]):
        ws.append((rng.standard_normal((o, i)) / np.sqrt(i)).astype(np.float32))
        bs.append((0.1 * rng.standard_normal(o)).astype(np.float32))
    S, out = dims[0], dims[-1]
    return O.Mlp(ws, bs, (0.1 * rng.standard_normal(S)).astype(np.float32), (1.0 + 0.2 * rng.random(S)).astype(np.float32),
                 (-0.3 + 0.1 * rng.standard_normal(out)).astype(np.float32) if with_std else None)


@pytest.mark.parametrize("S,hidden,A", NET_SHAPES)
def test_mlpn_spec_matches_module_layout(ops, S, hidden, A):
    spec = ops.MlpSpecN([S, *hidden, A], True)
    n = sum(o * i + o for i, o in zip([S, *hidden], [*hidden, A])) + A
    assert spec.count == n
    off = 0
    for name, o, shape in spec.slices():
        assert o == off
        off += int(np.prod(shape))
    assert off == n


@pytest.mark.parametrize("S,hidden,A", NET_SHAPES)
@pytest.mark.parametrize("rows", [1, 300, 4096])
def test_mlpn_value_forward(ops, dev, S, hidden, A, rows):
    rng = np.random.default_rng(rows + S)
    critic = random_net_n(rng, [S, *hidden, 1], False)
    x = rng.standard_normal((rows, S), dtype=np.float32)
    spec = ops.MlpSpecN([S, *hidden, 1], False)
    v = ops.mlpn_value_forward(cu(flat_params(critic), dev), spec, cu(critic.state_avg, dev), cu(critic.state_std, dev), cu(x, dev))
    ref = O.critic_value(x.astype(np.float64), critic.astype(np.float64))
    np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("S,hidden,A", NET_SHAPES)
def test_mlpn_rollout_step(ops, dev, S, hidden, A):
    rng = np.random.default_rng(S)
    N = 777
    actor = random_net_n(rng, [S, *hidden, A], True)
    x = rng.standard_normal((N, S), dtype=np.float32)
    eps = rng.standard_normal((N, A), dtype=np.float32)
    spec = ops.MlpSpecN([S, *hidden, A], True)
    o_s, o_a, o_l, o_e = (th.zeros((N, S), device=dev), th.zeros((N, A), device=dev), th.zeros(N, device=dev), th.zeros((N, A), device=dev))
    ops.mlpn_rollout_step(cu(flat_params(actor), dev), spec, cu(actor.state_avg, dev), cu(actor.state_std, dev), cu(x, dev),
                          noise=cu(eps, dev), out_state=o_s, out_action=o_a, out_logprob=o_l, out_env_action=o_e)
    a_ref, lp_ref = O.actor_sample(x.astype(np.float64), actor.astype(np.float64), eps.astype(np.float64))
    np.testing.assert_array_equal(o_s.cpu().numpy(), x)
    np.testing.assert_allclose(o_a.cpu().numpy(), a_ref, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(o_l.cpu().numpy(), lp_ref, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(o_e.cpu().numpy(), np.tanh(a_ref), rtol=1e-4, atol=2e-5)
    # Philox path: same stream as the fused kernel (seed, counter, env, action-dim)
    o_a2 = th.zeros((N, A), device=dev)
    ops.mlpn_rollout_step(cu(flat_params(actor), dev), spec, cu(actor.state_avg, dev), cu(actor.state_std, dev), cu(x, dev), seed=5,
                          counter=9, out_action=o_a2)
    z = (o_a2.cpu().numpy() - O.actor_mean(x.astype(np.float64), actor.astype(np.float64))) / np.exp(actor.action_std_log)
    assert abs(z.mean()) < 0.1 and abs(z.std() - 1.0) < 0.1


@pytest.mark.parametrize("S,hidden,A", NET_SHAPES)
@pytest.mark.parametrize("B", [64, 1000, 2321])      # 2321 = 9 chunks of 256 rows + 17: the split weight-gradient reduction
def test_mlpn_ppo_step_gradients(ops, dev, S, hidden, A, B):
    rng = np.random.default_rng(S + B)
    H, N = 9, 50
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = buf_ids[:6], buf_ids[6]
    actor, critic = random_net_n(rng, [S, *hidden, A], True), random_net_n(rng, [S, *hidden, 1], False)
    spec = ops.MlpSpecN([S, *hidden, A], True)
    Pa, Pc = spec.count, ops.MlpSpecN([S, *hidden, 1], False).count
    flat = th.full((Pa + Pc + 4,), float("nan"), device=dev)
    ops.mlpn_ppo_step(cu(flat_params(actor), dev), cu(flat_params(critic), dev), cu(actor.state_avg, dev), cu(actor.state_std, dev),
                      cu(critic.state_avg, dev), cu(critic.state_std, dev), spec, *[cu(x, dev) for x in buf], cu(ids, dev), 0.25, 0.001,
                      1.0 / B, flat)
    got = flat.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    ga, gc, objs = oracle_flat_grads(buf, ids, actor, critic, 0.25, 0.001, np.float64)
    for name, g, ref in (("actor", got[:Pa], ga), ("critic", got[Pa:Pa + Pc], gc)):
        scale = np.abs(ref).max()
        assert np.abs(g - ref).max() <= 1e-4 * scale + 1e-7, f"{name} grad err {np.abs(g - ref).max():.3e} (scale {scale:.3e})"
    np.testing.assert_allclose(got[Pa + Pc:Pa + Pc + 3], objs, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("objective,code", [("canonical", 1), ("a2c", 2)])
def test_mlpn_ppo_step_objective_forms(ops, dev, objective, code):
    """the layered path's objective kernel with the textbook clip and AgentA2C's objective (csrc/ppo_objective.h)."""
    rng = np.random.default_rng(len(objective))
    S, hidden, A, B, H, N = 11, [256, 128, 64], 3, 200, 9, 50
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = list(buf_ids[:6]), buf_ids[6]
    buf[3] = (buf[3] + 0.5 * rng.standard_normal(buf[3].shape)).astype(np.float32)
    actor, critic = random_net_n(rng, [S, *hidden, A], True), random_net_n(rng, [S, *hidden, 1], False)
    spec = ops.MlpSpecN([S, *hidden, A], True)
    Pa, Pc = spec.count, ops.MlpSpecN([S, *hidden, 1], False).count
    flat = th.full((Pa + Pc + 4,), float("nan"), device=dev)
    ops.mlpn_ppo_step(cu(flat_params(actor), dev), cu(flat_params(critic), dev), cu(actor.state_avg, dev), cu(actor.state_std, dev),
                      cu(critic.state_avg, dev), cu(critic.state_std, dev), spec, *[cu(x, dev) for x in buf], cu(ids, dev), 0.25, 0.001,
                      1.0 / B, flat, objective=code)
    got = flat.cpu().numpy().astype(np.float64)
    ga, gc, objs = oracle_flat_grads(buf, ids, actor, critic, 0.25, 0.001, np.float64, objective)
    for name, g, ref in (("actor", got[:Pa], ga), ("critic", got[Pa:Pa + Pc], gc)):
        scale = np.abs(ref).max()
        assert np.abs(g - ref).max() <= 1e-4 * scale + 1e-7, f"{name} grad err {np.abs(g - ref).max():.3e} (scale {scale:.3e})"
    np.testing.assert_allclose(got[Pa + Pc:Pa + Pc + 3], objs, rtol=1e-4, atol=1e-6)


def test_mlpn_agrees_with_fused_kernels_on_a_fused_shape(ops, dev):
    """on [128, 128] both paths exist: same gradient within fp32 summation-order noise."""
    rng = np.random.default_rng(3)
    S, A, B, H, N = 64, 8, 1024, 9, 300
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = [cu(x, dev) for x in buf_ids[:6]], cu(buf_ids[6], dev)
    actor, critic = random_net_n(rng, [S, 128, 128, A], True), random_net_n(rng, [S, 128, 128, 1], False)
    pa, pc = cu(flat_params(actor), dev), cu(flat_params(critic), dev)
    norms = [cu(x, dev) for x in (actor.state_avg, actor.state_std, critic.state_avg, critic.state_std)]
    stride, n_slabs = ops.ppo_slab_stride(S, 128, 128, A), ops.ppo_num_slabs(B)
    slabs, g1, g2 = th.zeros((n_slabs, stride), device=dev), th.zeros(stride, device=dev), th.zeros(stride, device=dev)
    ops.ppo_step(pa, pc, *norms, S, 128, 128, A, *buf, ids, 0.25, 0.001, 1.0 / B, slabs, n_slabs)
    ops.grad_reduce(slabs, n_slabs, stride, g1)
    ops.mlpn_ppo_step(pa, pc, *norms, ops.MlpSpecN([S, 128, 128, A], True), *buf, ids, 0.25, 0.001, 1.0 / B, g2)
    scale = g1.abs().max().item()
    assert (g1 - g2).abs().max().item() <= 2e-5 * scale + 1e-7


@pytest.mark.parametrize("hidden", [(256, 128), (256, 128, 64)])
def test_agent_ppo_on_reference_demo_shapes(hidden):
    """AgentPPO with the reference demos' larger nets: rollout, value pre-pass, GAE and two minibatches against the oracle."""
    from elegantrl_amd.agents import AgentPPO
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config
    N, S, A, H, B = 128, 24, 4, 8, 256
    dev = th.device("cuda:0")
    args = Config(AgentPPO, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 5, "state_dim": S, "action_dim": A,
                                        "if_discrete": False})
    args.net_dims = list(hidden)
    args.horizon_len, args.batch_size, args.repeat_times, args.learning_rate = H, B, 2 * B / H, 1e-3
    th.manual_seed(1)
    agent = AgentPPO(args.net_dims, S, A, gpu_id=0, args=args)
    assert not agent._fused
    env = SynVecEnv(N, S, A, max_step=5, gpu_id=0, seed=2)
    agent.last_state = env.reset()[0]
    g = th.Generator(device=dev).manual_seed(4)
    noise = th.randn((H, N, A), device=dev, generator=g)
    items = agent._explore_vec_env(env, H, noise=noise)
    states, actions, logprobs, rewards, undones, unmasks = [x.clone() for x in items]

    def to_mlp(net, with_std):
        f = lambda t: t.detach().cpu().numpy().astype(np.float64)   # noqa: E731
        idx = range(0, 2 * (len(hidden) + 1), 2)
        return O.Mlp([f(net.net[i].weight) for i in idx], [f(net.net[i].bias) for i in idx], f(net.state_avg), f(net.state_std),
                     f(net.action_std_log).reshape(-1) if with_std else None)

    actor, critic = to_mlp(agent.act, True), to_mlp(agent.cri, False)
    f64 = lambda t: t.cpu().numpy().astype(np.float64)   # noqa: E731
    a_ref, lp_ref = O.actor_sample(f64(states[0]), actor, f64(noise[0]))
    np.testing.assert_allclose(actions[0].cpu().numpy(), a_ref, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(logprobs[0].cpu().numpy(), lp_ref, rtol=1e-4, atol=1e-4)
    ids = th.randint(H * N, (2, B), device=dev, generator=g)
    objs = agent.update_net(list(items), ids=ids)
    v = O.critic_value(f64(states), critic)
    nv = O.critic_value(f64(agent.last_state), critic)
    adv, _, _ = O.gae_scan(f64(rewards), undones.cpu().numpy(), unmasks.cpu().numpy(), v, nv, args.gamma, 0.95)
    buf = (f64(states), f64(actions), unmasks.cpu().numpy(), f64(logprobs), O.adv_normalize(adv), O.reward_sums(adv, v))
    sa, sc = O.AdamState(), O.AdamState()
    ref = np.mean([O.ppo_minibatch_step(buf, i.cpu().numpy(), actor, critic, sa, sc, lr=args.learning_rate, max_norm=3.0,
                                        ratio_clip=0.25, lambda_entropy=0.001) for i in ids], axis=0)
    np.testing.assert_allclose(np.array(objs), ref, rtol=5e-4, atol=5e-6)
    np.testing.assert_allclose(agent.act.net[0].weight.detach().cpu().numpy(), actor.weights[0], rtol=0, atol=2e-5)
    np.testing.assert_allclose(agent.cri.net[2 * len(hidden)].weight.detach().cpu().numpy(), critic.weights[-1], rtol=0, atol=2e-5)


@pytest.mark.parametrize("S,h2,A,N", [(8, 128, 2, 4096), (24, 64, 6, 777), (33, 32, 1, 50), (64, 128, 16, 1000), (3, 96, 3, 16), (61, 128, 8, 16384),
                                      (17, (128, 64), 5, 777), (64, (128, 128), 8, 4096), (6, (32, 96), 1, 33)])
def test_wide_rollout_step_is_one_launch_and_matches_fp64(ops, dev, S, h2, A, N):
    """net_dims = (256, h2) (examples/demo_A2C_PPO.py:117): erl_mlpn_rollout_step_f32 takes the one-launch latency form (csrc/rollout_wide.hip) --
    actions, log-probs and tanh(actions) against the fp64 restatement (split-bf16 hidden layers: fp32-class), the state row bit for bit,
    aligned and unaligned state_dim, N not a multiple of the 16-env tile, every second-layer width."""
    mid = list(h2) if isinstance(h2, tuple) else [h2]                      # (h2,) or (h2, h3): examples/demo_A2C_PPO.py:171, :224
    rng = np.random.default_rng(S + sum(mid))
    actor = random_net_n(rng, [S, 256, *mid, A], True)
    x = rng.standard_normal((N, S), dtype=np.float32)
    eps = rng.standard_normal((N, A), dtype=np.float32)
    spec = ops.MlpSpecN([S, 256, *mid, A], True)
    o_s, o_a, o_l, o_e = (th.zeros((N, S), device=dev), th.zeros((N, A), device=dev), th.zeros(N, device=dev), th.zeros((N, A), device=dev))
    ops.mlpn_rollout_step(cu(flat_params(actor), dev), spec, cu(actor.state_avg, dev), cu(actor.state_std, dev), cu(x, dev),
                          noise=cu(eps, dev), out_state=o_s, out_action=o_a, out_logprob=o_l, out_env_action=o_e)
    a_ref, lp_ref = O.actor_sample(x.astype(np.float64), actor.astype(np.float64), eps.astype(np.float64))
    np.testing.assert_array_equal(o_s.cpu().numpy(), x)
    scale = max(1.0, float(np.abs(a_ref).max()))
    err = float(np.abs(o_a.cpu().numpy() - a_ref).max())
    assert err <= 1e-5 * scale, err
    np.testing.assert_allclose(o_l.cpu().numpy(), lp_ref, rtol=2e-5, atol=5e-5)
    np.testing.assert_allclose(o_e.cpu().numpy(), np.tanh(a_ref), rtol=0, atol=2e-5)
    # optional outputs may be absent; the Philox draws are those of every other rollout kernel: (seed, counter, env, action-dim)
    o_a2 = th.zeros((N, A), device=dev)
    ops.mlpn_rollout_step(cu(flat_params(actor), dev), spec, cu(actor.state_avg, dev), cu(actor.state_std, dev), cu(x, dev), seed=5, counter=9,
                          out_action=o_a2)
    z = (o_a2.cpu().numpy() - O.actor_mean(x.astype(np.float64), actor.astype(np.float64))) / np.exp(actor.action_std_log)
    if N >= 777:
        assert abs(z.mean()) < 0.1 and abs(z.std() - 1.0) < 0.1
    z2 = th.zeros((N, A), device=dev)
    ops.mlpn_rollout_step(cu(flat_params(actor), dev), spec, cu(actor.state_avg, dev), cu(actor.state_std, dev), cu(x, dev), seed=5, counter=9,
                          out_action=z2)
    assert th.equal(o_a2, z2)                   # deterministic in (seed, counter)


@pytest.mark.parametrize("S,mid,rows", [(8, (128,), 131072), (24, (64,), 20001), (33, (32,), 50), (64, (128,), 4097), (17, (128, 64), 9000),
                                        (64, (128, 128), 70000), (6, (32, 96), 17)])
def test_wide_value_forward_is_one_launch_and_matches_fp64(ops, dev, S, mid, rows):
    """net_dims = (256, h2[, h3]): erl_mlpn_value_forward_f32 takes the persistent-tile form (csrc/rollout_wide.hip value_wide_kernel: weights
    split once, kept in registers, workgroups walk 16-row tiles) -- against the fp64 restatement over more rows than one pass of the grid,
    row counts that are not multiples of the tile, aligned and unaligned state_dim, two and three hidden layers."""
    rng = np.random.default_rng(S + rows)
    critic = random_net_n(rng, [S, 256, *mid, 1], False)
    x = rng.standard_normal((rows, S), dtype=np.float32)
    spec = ops.MlpSpecN([S, 256, *mid, 1], False)
    v = ops.mlpn_value_forward(cu(flat_params(critic), dev), spec, cu(critic.state_avg, dev), cu(critic.state_std, dev), cu(x, dev))
    ref = O.critic_value(x.astype(np.float64), critic.astype(np.float64))
    err = float(np.abs(v.cpu().numpy() - ref).max())
    assert v.shape == (rows,) and err <= 4e-6 * max(1.0, float(np.abs(ref).max())), err
