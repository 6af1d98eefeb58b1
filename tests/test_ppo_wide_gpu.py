"""GPU parity tests of the PPO minibatch kernel for the reference's wider demo networks, net_dims = (256, 64 | 128)
(csrc/ppo_step_wd_impl.h; examples/demo_A2C_PPO.py:117 trains (256, 128)): against the fp64 restatement of
AgentPPO.update_objectives (elegantrl/agents/AgentPPO.py:173-204), against the layered erl_mlpn_* path it replaces for this
shape, and through the C update loop (quarter images of W2 kept current by clip + Adam)."""
import numpy as np
import pytest
import torch as th

from oracle import ppo_numpy as O
from tests.test_kernels_gpu import cu, flat_params, oracle_flat_grads, ppo_case, random_net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from elegantrl_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def dev():
    return th.device("cuda:0")


def blocks(S, h1, h2, out, with_std):
    """(name, offset, length) of a network's parameter blocks in flat order"""
    names = [("W1", h1 * S), ("b1", h1), ("W2", h2 * h1), ("b2", h2), ("W3", out * h2), ("b3", out)] + ([("std", out)] if with_std else [])
    res, o = [], 0
    for n, ln in names:
        res.append((n, o, ln))
        o += ln
    return res


def block_errors(got, ref, S, h1, h2, out, with_std):
    """max |got - ref| per parameter block, relative to the WHOLE gradient's scale (so that a wrong block stands out by name)"""
    scale = max(1e-30, np.abs(ref).max())
    return {n: float(np.abs(got[o:o + ln] - ref[o:o + ln]).max() / scale) for n, o, ln in blocks(S, h1, h2, out, with_std)}


def wide_step(ops, dev, S, h1, h2, A, B, buf, ids, actor, critic, H, N, objective=0):
    Pa, Pc = ops.MlpSpec(S, h1, h2, A, True).count, ops.MlpSpec(S, h1, h2, 1, False).count
    n_slabs, stride = ops.ppo_num_slabs(B), ops.ppo_slab_stride(S, h1, h2, A)
    assert stride >= Pa + Pc + 4 and stride % 32 == 0
    P = cu(np.concatenate([flat_params(actor), flat_params(critic)]), dev)       # [actor | critic]: the stand-alone call builds the images from it
    slabs = th.full((n_slabs, stride), float("nan"), device=dev)
    flat = th.zeros(stride, device=dev)
    ops.ppo_step(P[:Pa], P[Pa:], cu(actor.state_avg, dev), cu(actor.state_std, dev), cu(critic.state_avg, dev), cu(critic.state_std, dev),
                 S, h1, h2, A, *[cu(x, dev) for x in buf], cu(ids, dev), 0.25, 0.001, 1.0 / B, slabs, n_slabs, objective=objective)
    ops.grad_reduce(slabs, n_slabs, stride, flat)
    got = flat.cpu().numpy().astype(np.float64)
    return got, Pa, Pc


WIDE_SHAPES = [(64, 256, 128, 8), (24, 256, 128, 4), (3, 256, 128, 1), (60, 256, 128, 8), (17, 256, 64, 6), (8, 256, 64, 2), (33, 256, 128, 5)]


@pytest.mark.parametrize("S,h1,h2,A", WIDE_SHAPES)
@pytest.mark.parametrize("B", [200, 1024, 1])
def test_wide_step_against_fp64(ops, dev, S, h1, h2, A, B):
    """gradients of both networks and the three logged objectives against the fp64 restatement: within fp32 rounding of the gradient's
    scale (2e-6: the [128,128] kernels' own errors on such cases reach 8.5e-7; one bf16 rounding would be 4e-3)"""
    assert ops.ppo_arith_in_use(S, h1, h2, A) == "split"
    rng = np.random.default_rng(11 * S + B)
    H, N = 9, 50
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = buf_ids[:6], buf_ids[6]
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    ga, gc, objs = oracle_flat_grads(buf, ids, actor, critic, 0.25, 0.001, np.float64)
    got, Pa, Pc = wide_step(ops, dev, S, h1, h2, A, B, buf, ids, actor, critic, H, N)
    assert np.isfinite(got).all(), "a slab slot was left unwritten"
    assert not np.any(got[Pa + Pc + 4:])
    ea = block_errors(got[:Pa], ga, S, h1, h2, A, True)
    ec = block_errors(got[Pa:Pa + Pc], gc, S, h1, h2, 1, False)
    eo = np.abs(got[Pa + Pc:Pa + Pc + 3] - objs) / np.maximum(1e-30, np.abs(objs).max())
    print(f"S={S} net=({h1},{h2}) A={A} B={B}: actor {ea}\n  critic {ec}\n  objectives {eo}")
    assert max(ea.values()) <= 2e-6, f"actor gradient: {ea}"
    assert max(ec.values()) <= 2e-6, f"critic gradient: {ec}"
    assert eo.max() <= 2e-6, f"objectives: {eo}"


@pytest.mark.parametrize("B", [200, 128 * 5 + 17, 128 * 11 + 1, 128 * 35 + 3, 16384])
def test_wide_step_workgroup_maps_agree(ops, dev, B, monkeypatch):
    """the (256, h2) and (256, 128, h3) kernels under both workgroup maps (csrc/ppo_step.h k6_wg_map; ERL_K6_WG_MAP is read per launch): the
    same summed gradient row bit for bit, incl. slab counts that are no multiple of 4 and the full-size launch (which, unforced, is the one
    that measures both maps on this device: include/erl_hip.h erl_ppo_wg_map_info with ERL_PPO_WG_FAMILY_WIDE)"""
    from elegantrl_amd import _hip
    S, h1, h2, A, H, N = 24, 256, 128, 4, 9, 2000
    rng = np.random.default_rng(B)
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = buf_ids[:6], buf_ids[6]
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    got, got3 = {}, {}
    for wg_map in ("0", "1", "2"):
        monkeypatch.setenv("ERL_K6_WG_MAP", wg_map)
        got[wg_map] = wide_step(ops, dev, S, h1, h2, A, B, buf, ids, actor, critic, H, N)[0]
        got3[wg_map] = wide3_step(ops, dev, 17, 64, 5, B, H, N, 3 + B)[0]
        assert np.isfinite(got[wg_map]).all() and np.isfinite(got3[wg_map]).all()
        assert _hip.ppo_wg_map_info(wide=True)["map"] == int(wg_map)
    assert np.array_equal(got["0"], got["1"]) and np.array_equal(got3["0"], got3["1"])
    assert np.array_equal(got["0"], got["2"]) and np.array_equal(got3["0"], got3["2"])
    monkeypatch.delenv("ERL_K6_WG_MAP")
    auto = wide_step(ops, dev, S, h1, h2, A, B, buf, ids, actor, critic, H, N)[0]
    assert np.array_equal(auto, got["0"])
    info = _hip.ppo_wg_map_info(wide=True)
    if B == 16384:
        print("wide kernels' workgroup map on this box:", info)
        assert info["map"] in (0, 2) and info["us_map0"] and info["us_map2"]


@pytest.mark.parametrize("S,h2,A", [(24, 128, 4), (8, 64, 2)])
@pytest.mark.parametrize("objective", ["canonical", "a2c"])
def test_wide_step_objective_forms(ops, dev, S, h2, A, objective):
    """the textbook clipped surrogate and AgentA2C's un-clipped objective (AgentPPO.py:296-303; examples/demo_A2C_PPO.py trains A2C on
    the same (256, 128) network) through the wide kernel, ratios on both sides of the clip"""
    h1, B, H, N = 256, 300, 9, 50
    rng = np.random.default_rng(S + len(objective))
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = list(buf_ids[:6]), buf_ids[6]
    buf[3] = (buf[3] + 0.5 * rng.standard_normal(buf[3].shape)).astype(np.float32)
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    ga, gc, objs = oracle_flat_grads(buf, ids, actor, critic, 0.25, 0.001, np.float64, objective)
    got, Pa, Pc = wide_step(ops, dev, S, h1, h2, A, B, buf, ids, actor, critic, H, N, objective={"canonical": 1, "a2c": 2}[objective])
    ea = block_errors(got[:Pa], ga, S, h1, h2, A, True)
    ec = block_errors(got[Pa:Pa + Pc], gc, S, h1, h2, 1, False)
    assert max(ea.values()) <= 2e-6 and max(ec.values()) <= 2e-6, (ea, ec)
    np.testing.assert_allclose(got[Pa + Pc:Pa + Pc + 3], objs, rtol=1e-5, atol=1e-6)


def test_wide_step_matches_the_layered_path(ops, dev):
    """the kernel replaces erl_mlpn_ppo_step_f32 for this shape: same summed gradient row (the layered path's fp32-MFMA GEMMs and
    this kernel's split arithmetic differ by fp32 rounding only)"""
    S, h1, h2, A, B, H, N = 24, 256, 128, 4, 512, 9, 100
    rng = np.random.default_rng(5)
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = buf_ids[:6], buf_ids[6]
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    got, Pa, Pc = wide_step(ops, dev, S, h1, h2, A, B, buf, ids, actor, critic, H, N)
    spec = ops.MlpSpecN([S, h1, h2, A], True)
    assert spec.count == Pa
    g = th.zeros(Pa + Pc + 4, device=dev)
    ops.mlpn_ppo_step(cu(flat_params(actor), dev), cu(flat_params(critic), dev), cu(actor.state_avg, dev), cu(actor.state_std, dev),
                      cu(critic.state_avg, dev), cu(critic.state_std, dev), spec, *[cu(x, dev) for x in buf], cu(ids, dev), 0.25, 0.001, 1.0 / B, g)
    lay = g.cpu().numpy().astype(np.float64)
    for name, sl in (("actor", slice(0, Pa)), ("critic", slice(Pa, Pa + Pc)), ("objectives", slice(Pa + Pc, Pa + Pc + 3))):
        d = np.abs(got[sl] - lay[sl]).max() / np.abs(lay[sl]).max()
        assert d < 5e-6, (name, d)


def test_wide_step_at_demo_batch_size(ops, dev):
    """a full-size minibatch (128 slabs per network: 16384 samples out of 32 x 4096) at the Ant-like shape S = 64, A = 8"""
    S, h1, h2, A, H, N, B = 64, 256, 128, 8, 32, 4096, 16384
    rng = np.random.default_rng(2025)
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = buf_ids[:6], buf_ids[6]
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    ga, gc, objs = oracle_flat_grads(buf, ids, actor, critic, 0.25, 0.001, np.float64)
    got, Pa, Pc = wide_step(ops, dev, S, h1, h2, A, B, buf, ids, actor, critic, H, N)
    assert np.isfinite(got).all()
    ea = block_errors(got[:Pa], ga, S, h1, h2, A, True)
    ec = block_errors(got[Pa:Pa + Pc], gc, S, h1, h2, 1, False)
    eo = np.abs(got[Pa + Pc:Pa + Pc + 3] - objs).max() / np.abs(objs).max()
    print(f"actor {ea}\ncritic {ec}\nobjectives {eo:.2e}")
    assert max(ea.values()) < 1e-6 and max(ec.values()) < 1e-6 and eo < 1e-6


@pytest.mark.parametrize("S,A,B,h2", [(64, 8, 512, 128), (24, 4, 300, 128), (3, 1, 128, 128), (17, 5, 200, 64)])
def test_wide_update_loop_keeps_the_quarter_images_current(ops, dev, S, A, B, h2):
    """the C update loop builds the W1 image and the four column-quarter images of W2 once and lets clip + Adam refresh them element by
    element; the stand-alone erl_ppo_step_f32 rebuilds them from the fp32 weights at every call: same bits after six Adam steps --
    weights, moments, gradient rows"""
    h1 = 256
    rng = np.random.default_rng(S + B)
    H, N, T = 9, 400, 6
    buf = ppo_case(rng, H, N, S, A, B)[:6]
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    Pa, Pc = ops.MlpSpec(S, h1, h2, A, True).count, ops.MlpSpec(S, h1, h2, 1, False).count
    stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)
    ids = cu(rng.integers(0, H * N, (T, B)), dev)
    P0 = cu(np.concatenate([flat_params(actor), flat_params(critic)]), dev)
    norm = [cu(actor.state_avg, dev), cu(actor.state_std, dev), cu(critic.state_avg, dev), cu(critic.state_std, dev)]
    tb = [cu(x, dev) for x in buf]
    groups = [(0, Pa), (Pa, Pc)]
    P, M1, M2 = P0.clone(), th.zeros_like(P0), th.zeros_like(P0)
    slabs, rows = th.zeros((n_slabs, stride), device=dev), th.zeros((T, stride), device=dev)
    for k in range(T):
        ops.ppo_step(P[:Pa], P[Pa:], *norm, S, h1, h2, A, *tb, ids[k], 0.25, 0.001, 1.0 / B, slabs, n_slabs)
        ops.grad_reduce_partials(slabs, n_slabs, stride, rows[k], groups)
        ops.clip_adam_partials(P, rows[k], M1, M2, stride, groups, k + 1, 1e-3, 3.0)
    Pc_, M1c, M2c = P0.clone(), th.zeros_like(P0), th.zeros_like(P0)
    slabs_c, rows_c = th.zeros((n_slabs, stride), device=dev), th.zeros((T, stride), device=dev)
    ops.ppo_update(Pc_, M1c, M2c, *norm, S, h1, h2, A, *tb, ids, 0.25, 0.001, slabs_c, rows_c, 1, 1e-3, 3.0)
    assert th.isfinite(P).all() and not th.equal(P, P0)
    for name, x, y in (("rows", rows, rows_c), ("weights", P, Pc_), ("exp_avg", M1, M1c), ("exp_avg_sq", M2, M2c)):
        assert th.equal(x, y), f"{name}: max difference {(x - y).abs().max().item():.3e}"


def test_agent_wide_against_the_layered_update():
    """AgentPPO at net_dims (256, 128): the same rollout, then update_net through the fused minibatch kernel and through the layered path
    it replaces (args.wide_fused = False) -- same objectives and weights up to fp32 rounding through Adam"""
    from elegantrl_amd.agents import AgentPPO
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config
    N, S, A, H, B = 128, 24, 4, 8, 256
    dev = th.device("cuda:0")
    out = {}
    for wide in (True, False):
        args = Config(AgentPPO, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 5, "state_dim": S, "action_dim": A, "if_discrete": False})
        args.net_dims = [256, 128]
        args.horizon_len, args.batch_size, args.repeat_times, args.learning_rate = H, B, 2 * B / H, 1e-3
        args.wide_fused = wide
        th.manual_seed(1)
        agent = AgentPPO(args.net_dims, S, A, gpu_id=0, args=args)
        assert agent._wide == wide and not agent._fused
        env = SynVecEnv(N, S, A, max_step=5, gpu_id=0, seed=2)
        agent.last_state = env.reset()[0]
        g = th.Generator(device=dev).manual_seed(4)
        noise = th.randn((H, N, A), device=dev, generator=g)
        items = agent._explore_vec_env(env, H, noise=noise)
        ids = th.randint(H * N, (2, B), device=dev, generator=g)
        objs = agent.update_net(list(items), ids=ids)
        out[wide] = (np.array(objs), agent._flat.detach().cpu().numpy().copy())
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=2e-5, atol=2e-6)
    # Adam's first steps move every weight by ~lr whatever the gradient's size: elements whose gradient sits at rounding level may differ by 2 lr
    d = np.abs(out[True][1] - out[False][1])
    assert np.quantile(d, 0.999) < 2e-5 and d.max() < 4.1e-3, (np.quantile(d, 0.999), d.max())


def test_agent_a2c_wide_against_the_layered_update():
    """AgentA2C (whole time rows per minibatch, un-clipped objective) at net_dims (256, 128) through the fused kernel and through the
    layered path: same objectives, same weights up to fp32 rounding through Adam"""
    from elegantrl_amd.agents import AgentA2C
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config
    N, S, A, H = 1, 24, 4, 256
    dev = th.device("cuda:0")
    out = {}
    for wide in (True, False):
        args = Config(AgentA2C, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 50, "state_dim": S, "action_dim": A, "if_discrete": False})
        args.net_dims = [256, 128]
        args.horizon_len, args.batch_size, args.repeat_times, args.learning_rate = H, 128, 1.0, 1e-3
        args.wide_fused = wide
        th.manual_seed(3)
        agent = AgentA2C(args.net_dims, S, A, gpu_id=0, args=args)
        assert agent._wide == wide
        env = SynVecEnv(N, S, A, max_step=50, gpu_id=0, seed=5)
        agent.last_state = env.reset()[0]
        g = th.Generator(device=dev).manual_seed(6)
        noise = th.randn((H, N, A), device=dev, generator=g)
        items = agent._explore_vec_env(env, H, noise=noise)
        th.manual_seed(7)                                   # (the agent draws its minibatch rows with the global generator)
        objs = agent.update_net(list(items))
        out[wide] = (np.array(objs), agent._flat.detach().cpu().numpy().copy())
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=5e-5, atol=5e-6)
    d = np.abs(out[True][1] - out[False][1])
    assert np.quantile(d, 0.999) < 2e-5 and d.max() < 4.1e-3, (np.quantile(d, 0.999), d.max())


# ---- three hidden layers: net_dims (256, 128, 64 | 128) behind erl_mlpn_ppo_step_f32 ------------------------------------------------
def blocks3(S, h3, out, with_std):
    names = [("W1", 256 * S), ("b1", 256), ("W2", 128 * 256), ("b2", 128), ("W3", h3 * 128), ("b3", h3), ("W4", out * h3), ("b4", out)]
    names += [("std", out)] if with_std else []
    res, o = [], 0
    for n, ln in names:
        res.append((n, o, ln))
        o += ln
    return res


def block_errors3(got, ref, S, h3, out, with_std):
    scale = max(1e-30, np.abs(ref).max())
    return {n: float(np.abs(got[o:o + ln] - ref[o:o + ln]).max() / scale) for n, o, ln in blocks3(S, h3, out, with_std)}


def wide3_step(ops, dev, S, h3, A, B, H, N, seed, objective=0, noisy_logprobs=False):
    from tests.test_mlpn_gpu import random_net_n
    rng = np.random.default_rng(seed)
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = list(buf_ids[:6]), buf_ids[6]
    if noisy_logprobs:
        buf[3] = (buf[3] + 0.5 * rng.standard_normal(buf[3].shape)).astype(np.float32)
    dims = [S, 256, 128, h3]
    actor, critic = random_net_n(rng, dims + [A], True), random_net_n(rng, dims + [1], False)
    spec = ops.MlpSpecN(dims + [A], True)
    Pa, Pc = spec.count, ops.MlpSpecN(dims + [1], False).count
    flat = th.full((Pa + Pc + 4,), float("nan"), device=dev)
    ops.mlpn_ppo_step(cu(flat_params(actor), dev), cu(flat_params(critic), dev), cu(actor.state_avg, dev), cu(actor.state_std, dev),
                      cu(critic.state_avg, dev), cu(critic.state_std, dev), spec, *[cu(x, dev) for x in buf], cu(ids, dev), 0.25, 0.001,
                      1.0 / B, flat, objective=objective)
    got = flat.cpu().numpy().astype(np.float64)
    return got, Pa, Pc, buf, ids, actor, critic


WIDE3_SHAPES = [(64, 128, 8), (24, 64, 4), (17, 64, 5), (3, 128, 1), (60, 64, 8), (8, 128, 2)]


@pytest.mark.parametrize("S,h3,A", WIDE3_SHAPES)
@pytest.mark.parametrize("B", [200, 1024, 1])
def test_wide3_step_against_fp64(ops, dev, S, h3, A, B):
    """net_dims (256, 128, h3) (examples/demo_A2C_PPO.py:171, :224): erl_mlpn_ppo_step_f32 routes this shape to the fused kernel; both
    networks' gradients per parameter block and the three objectives against the fp64 restatement"""
    got, Pa, Pc, buf, ids, actor, critic = wide3_step(ops, dev, S, h3, A, B, 9, 50, 13 * S + B)
    assert np.isfinite(got).all(), "a slab slot was left unwritten"
    ga, gc, objs = oracle_flat_grads(buf, ids, actor, critic, 0.25, 0.001, np.float64)
    ea = block_errors3(got[:Pa], ga, S, h3, A, True)
    ec = block_errors3(got[Pa:Pa + Pc], gc, S, h3, 1, False)
    eo = np.abs(got[Pa + Pc:Pa + Pc + 3] - objs) / np.maximum(1e-30, np.abs(objs).max())
    print(f"S={S} net=(256,128,{h3}) A={A} B={B}: actor {ea}\n  critic {ec}\n  objectives {eo}")
    assert max(ea.values()) <= 2e-6, f"actor gradient: {ea}"
    assert max(ec.values()) <= 2e-6, f"critic gradient: {ec}"
    assert eo.max() <= 2e-6, f"objectives: {eo}"
    assert got[Pa + Pc + 3] == 0.0


@pytest.mark.parametrize("objective", ["canonical", "a2c"])
def test_wide3_step_objective_forms(ops, dev, objective):
    got, Pa, Pc, buf, ids, actor, critic = wide3_step(ops, dev, 24, 64, 4, 300, 9, 50, len(objective), {"canonical": 1, "a2c": 2}[objective], True)
    ga, gc, objs = oracle_flat_grads(buf, ids, actor, critic, 0.25, 0.001, np.float64, objective)
    ea, ec = block_errors3(got[:Pa], ga, 24, 64, 4, True), block_errors3(got[Pa:Pa + Pc], gc, 24, 64, 1, False)
    assert max(ea.values()) <= 2e-6 and max(ec.values()) <= 2e-6, (ea, ec)
    np.testing.assert_allclose(got[Pa + Pc:Pa + Pc + 3], objs, rtol=1e-5, atol=1e-6)


def test_wide3_step_at_demo_batch_size(ops, dev):
    """a full-size minibatch (128 slabs per network) at (256, 128, 128), S = 64, A = 8"""
    got, Pa, Pc, buf, ids, actor, critic = wide3_step(ops, dev, 64, 128, 8, 16384, 32, 4096, 77)
    ga, gc, objs = oracle_flat_grads(buf, ids, actor, critic, 0.25, 0.001, np.float64)
    ea, ec = block_errors3(got[:Pa], ga, 64, 128, 8, True), block_errors3(got[Pa:Pa + Pc], gc, 64, 128, 1, False)
    eo = np.abs(got[Pa + Pc:Pa + Pc + 3] - objs).max() / np.abs(objs).max()
    print(f"actor {ea}\ncritic {ec}\nobjectives {eo:.2e}")
    assert max(ea.values()) < 1e-6 and max(ec.values()) < 1e-6 and eo < 1e-6
