"""GPU parity tests: every HIP kernel (through the C ABI) against the CPU oracle on seeded inputs.

Bars (BASELINE.json north_star / SURVEY.md 8c): indices and gathered rows bit-exact; advantages and
returns within 1e-5 * max(1, |ref|) fp32 (bit-exact in ERL_GAE_ALGO_EXACT mode); MLP / loss / gradient
math within rtol 1e-4 of the fp32 oracle (MFMA accumulation order differs from torch's GEMM).
"""
import numpy as np
import pytest
import torch as th

from elegantrl_amd import _hip
from oracle import c_oracle
from oracle import ppo_numpy as O
from tests.helpers import PPO_GOLDENS, dims, hyper, load, mlp_from

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from elegantrl_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def dev():
    return th.device("cuda:0")


def cu(a, dev, dtype=None):
    t = th.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


def rel_close(x, ref, tol):
    x, ref = np.asarray(x, np.float64), np.asarray(ref, np.float64)
    err = np.abs(x - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= tol, f"max rel err {err.max():.3e} > {tol:.1e} at {np.unravel_index(err.argmax(), err.shape)}"


def gae_inputs(H, N, seed, p_done=0.01, p_trunc=0.005):
    rng = np.random.default_rng(seed)
    r = rng.standard_normal((H, N), dtype=np.float32)
    v = rng.standard_normal((H, N), dtype=np.float32)
    u = rng.random((H, N)) >= p_done
    m = rng.random((H, N)) >= p_trunc
    nv = rng.standard_normal(N, dtype=np.float32)
    return r, u, m, v, nv


def flat_params(net: O.Mlp):
    return np.concatenate([p.reshape(-1) for p in net.trainable()]).astype(np.float32)


# ------------------------------------------------------------------------------------------------
def test_mfma_tile_mapping():
    from elegantrl_amd import _hip
    assert _hip.selftest_mfma() < 1e-5


@pytest.mark.parametrize("H,N", [(32, 4096), (1, 64), (7, 1), (33, 130), (200, 4096), (5, 1000)])
@pytest.mark.parametrize("vtrace", [True, False])
def test_gae_exact_is_bitwise_equal_to_oracle(ops, dev, H, N, vtrace):
    r, u, m, v, nv = gae_inputs(H, N, seed=H * 1000 + N, p_done=0.05, p_trunc=0.03)
    adv_o, ret_o, r_o, u_o = c_oracle.gae(r, u, m, v, nv, 0.99, 0.95, use_v_trace=vtrace)
    tr, tu, tm, tv, tnv = cu(r, dev), cu(u, dev), cu(m, dev), cu(v, dev), cu(nv, dev)
    adv, ret = ops.gae_scan(tr, tu, tm, tv, tnv, 0.99, 0.95, use_v_trace=vtrace, mutate=True, algo="exact")
    np.testing.assert_array_equal(adv.cpu().numpy(), adv_o)
    np.testing.assert_array_equal(ret.cpu().numpy(), ret_o)
    np.testing.assert_array_equal(tr.cpu().numpy(), r_o)     # in-place truncation fix-up, like the reference
    np.testing.assert_array_equal(tu.cpu().numpy(), u_o)


def test_gae_exact_edge_rows(ops, dev):
    H, N = 16, 128
    r, u, m, v, nv = gae_inputs(H, N, seed=3)
    u[:, :32] = False          # every step terminal
    m[:, 32:64] = False        # every step truncated
    u[:, 64:96] = True
    m[:, 64:96] = True         # never done
    adv_o, ret_o, _, _ = c_oracle.gae(r, u, m, v, nv, 0.99, 0.95)
    adv, ret = ops.gae_scan(cu(r, dev), cu(u, dev), cu(m, dev), cu(v, dev), cu(nv, dev), 0.99, 0.95, algo="exact")
    np.testing.assert_array_equal(adv.cpu().numpy(), adv_o)
    np.testing.assert_array_equal(ret.cpu().numpy(), ret_o)


@pytest.mark.parametrize("H,N", [(256, 1000), (513, 260), (128, 4096), (1024, 512), (40, 77), (9, 64)])
@pytest.mark.parametrize("vtrace", [True, False])
def test_gae_chunked_within_1e5(ops, dev, H, N, vtrace):
    r, u, m, v, nv = gae_inputs(H, N, seed=H + N)
    adv_o, ret_o, r_o, u_o = c_oracle.gae(r, u, m, v, nv, 0.99, 0.95, use_v_trace=vtrace)
    tr, tu = cu(r, dev), cu(u, dev)
    adv, ret = ops.gae_scan(tr, tu, cu(m, dev), cu(v, dev), cu(nv, dev), 0.99, 0.95, use_v_trace=vtrace, algo="chunked")
    rel_close(adv.cpu().numpy(), adv_o, 1e-5)
    rel_close(ret.cpu().numpy(), ret_o, 1e-5)
    np.testing.assert_array_equal(tu.cpu().numpy(), u_o)
    np.testing.assert_allclose(tr.cpu().numpy(), r_o, rtol=0, atol=1e-6)


@pytest.mark.parametrize("H,N", [(256, 1000), (513, 260), (128, 4096), (1024, 512), (40, 76), (9, 64), (4, 4), (2048, 256)])
@pytest.mark.parametrize("vtrace", [True, False])
def test_gae_lookback_within_1e5(ops, dev, H, N, vtrace):
    """single-pass decoupled look-back scan (N % 4 == 0): same bar as the chunked scan, plus the in-place
    truncation fix-up of rewards / undones."""
    r, u, m, v, nv = gae_inputs(H, N, seed=H + N)
    adv_o, ret_o, r_o, u_o = c_oracle.gae(r, u, m, v, nv, 0.99, 0.95, use_v_trace=vtrace)
    tr, tu = cu(r, dev), cu(u, dev)
    adv, ret = ops.gae_scan(tr, tu, cu(m, dev), cu(v, dev), cu(nv, dev), 0.99, 0.95, use_v_trace=vtrace, algo="lookback")
    rel_close(adv.cpu().numpy(), adv_o, 1e-5)
    rel_close(ret.cpu().numpy(), ret_o, 1e-5)
    np.testing.assert_array_equal(tu.cpu().numpy(), u_o)
    np.testing.assert_allclose(tr.cpu().numpy(), r_o, rtol=0, atol=1e-6)


@pytest.mark.parametrize("H,N", [(64, 36), (100, 44), (255, 260), (256, 4100), (129, 32), (200, 4096), (130, 8200)])
def test_gae_one_workgroup_per_32_envs_against_c_oracle(ops, dev, H, N, monkeypatch):
    """64 <= H <= 256 (round 6: gae_tall_kernel -- a workgroup holds the whole horizon of 32 envs, 32 time chunks composed in LDS; or of
    16 envs, 64 chunks, when 32 would leave CUs without a workgroup): ragged env groups, horizons that do not fill the last chunk, long
    undone chains so that the carry crosses every chunk, the statistics partials, the in-place truncation fix-up; both widths forced
    (ERL_GAE_TALL_ENVS), the library's own choice, and the slab form (ERL_GAE_TALL=0) on the same inputs, all within the same bar."""
    r, u, m, v, nv = gae_inputs(H, N, seed=H * 3 + N, p_done=0.002, p_trunc=0.002)
    adv_o, ret_o, r_o, u_o = c_oracle.gae(r, u, m, v, nv, 0.99, 0.95, use_v_trace=True)
    for tall, envs in (("1", "32"), ("1", "16"), ("1", None), ("0", None)):
        monkeypatch.setenv("ERL_GAE_TALL", tall)
        if envs is None:
            monkeypatch.delenv("ERL_GAE_TALL_ENVS", raising=False)
        else:
            monkeypatch.setenv("ERL_GAE_TALL_ENVS", envs)
        tr, tu = cu(r, dev), cu(u, dev)
        stats = th.zeros(8, dtype=th.float64, device=dev)
        adv, ret = ops.gae_scan(tr, tu, cu(m, dev), cu(v, dev), cu(nv, dev), 0.99, 0.95, use_v_trace=True, algo="lookback", stats=stats)
        adv_np = adv.cpu().numpy()
        rel_close(adv_np, adv_o, 1e-5)
        rel_close(ret.cpu().numpy(), ret_o, 1e-5)
        np.testing.assert_array_equal(tu.cpu().numpy(), u_o)
        np.testing.assert_allclose(tr.cpu().numpy(), r_o, rtol=0, atol=1e-6)
        s_, sub = stats.cpu().numpy(), adv_np[::4, ::4].astype(np.float64)
        np.testing.assert_allclose(s_[0], adv_np.astype(np.float64).sum(), rtol=1e-9, atol=1e-6)
        assert s_[1] == H * N and s_[4] == sub.size
        np.testing.assert_allclose(s_[2], sub.sum(), rtol=1e-9, atol=1e-6)
        np.testing.assert_allclose(s_[3], (sub * sub).sum(), rtol=1e-9)
    th.cuda.synchronize()
    from elegantrl_amd import _hip
    _hip.check_async_faults()


@pytest.mark.parametrize("H,N", [(2048, 4096), (200, 4096), (32, 32768)], ids=["baseline-2048x4096", "config2-200x4096", "32x32768"])
@pytest.mark.parametrize("vtrace", [True, False])
def test_gae_lookback_at_baseline_sizes_against_c_oracle(ops, dev, H, N, vtrace):
    """the metric's own sizes (SURVEY.md 8d sweep; 2048 x 4096 = 151 MB) checked DIRECTLY against oracle/gae_scan.c, not
    against another kernel of this library: |x - ref| <= 1e-5 max(1, |ref|) (north star), flags bit-exact."""
    r, u, m, v, nv = gae_inputs(H, N, seed=H * 7 + N)
    adv_o, ret_o, r_o, u_o = c_oracle.gae(r, u, m, v, nv, 0.99, 0.95, use_v_trace=vtrace)
    tr, tu = cu(r, dev), cu(u, dev)
    adv, ret = ops.gae_scan(tr, tu, cu(m, dev), cu(v, dev), cu(nv, dev), 0.99, 0.95, use_v_trace=vtrace, algo="lookback")
    rel_close(adv.cpu().numpy(), adv_o, 1e-5)
    rel_close(ret.cpu().numpy(), ret_o, 1e-5)
    np.testing.assert_array_equal(tu.cpu().numpy(), u_o)
    np.testing.assert_allclose(tr.cpu().numpy(), r_o, rtol=0, atol=1e-6)
    th.cuda.synchronize()
    from elegantrl_amd import _hip
    _hip.check_async_faults()                                  # no look-back wait timed out


def test_gae_lookback_tables_are_per_stream(ops, dev):
    """the look-back scan's library-owned granule table is keyed by (device, stream) (ABI 17; round 4: one per device, a second
    stream took the memset path): scans interleaved on three streams -- two side streams and the default one, different sizes,
    several rounds so that nonces advance independently -- all match oracle/gae_scan.c"""
    from elegantrl_amd import _hip
    shapes = [(256, 1024), (1024, 512), (128, 4096)]
    cases = []
    for k, (H, N) in enumerate(shapes):
        r, u, m, v, nv = gae_inputs(H, N, seed=77 + k)
        adv_o, ret_o, _, _ = c_oracle.gae(r, u, m, v, nv, 0.99, 0.95)
        cases.append(((r, u, m, v, nv), adv_o, ret_o))
    streams = [th.cuda.Stream(device=dev), th.cuda.Stream(device=dev), th.cuda.current_stream(dev)]
    outs = []
    th.cuda.synchronize()
    for rnd in range(3):
        for (inp, _, _), st in zip(cases, streams):
            with th.cuda.stream(st):
                t = [cu(x, dev) for x in inp]
                outs.append((rnd, ops.gae_scan(*t, 0.99, 0.95, mutate=False, algo="lookback")))
    th.cuda.synchronize()
    for i, (rnd, (adv, ret)) in enumerate(outs):
        _, adv_o, ret_o = cases[i % 3]
        rel_close(adv.cpu().numpy(), adv_o, 1e-5)
        rel_close(ret.cpu().numpy(), ret_o, 1e-5)
    _hip.check_async_faults()


def test_gae_lookback_timeout_is_reported_not_silent(ops, dev, monkeypatch):
    """a predecessor slab that never publishes (fault injection: granules written under a foreign nonce) makes the bounded
    look-back wait expire: the affected advantages are NaN AND the fault reaches the host through the ABI
    (erl_async_fault_count / _hip.check_async_faults), which AgentPPO.update_net checks at its host sync."""
    from elegantrl_amd import _hip
    th.cuda.synchronize()
    assert _hip.lib().erl_async_fault_count(1) >= 0            # start clean
    r, u, m, v, nv = gae_inputs(512, 256, seed=5, p_done=0.0, p_trunc=0.0)
    args = (cu(m, dev), cu(v, dev), cu(nv, dev), 0.99, 0.95)
    monkeypatch.setenv("ERL_GAE_LB_FAULT", "1")
    monkeypatch.setenv("ERL_GAE_LB_SPIN", "64")
    adv, _ = ops.gae_scan(cu(r, dev), cu(u, dev), *args, algo="lookback")
    th.cuda.synchronize()
    assert th.isnan(adv).any()
    with pytest.raises(_hip.HipExtensionError, match="look-back"):
        _hip.check_async_faults()
    _hip.check_async_faults()                                  # the counter was cleared by the failing check
    monkeypatch.delenv("ERL_GAE_LB_FAULT")
    monkeypatch.delenv("ERL_GAE_LB_SPIN")
    adv, _ = ops.gae_scan(cu(r, dev), cu(u, dev), *args, algo="lookback")
    th.cuda.synchronize()
    assert not th.isnan(adv).any()
    _hip.check_async_faults()


@pytest.mark.parametrize("L,W,delay", [(4, 8, 1), (4, 8, 6), (2, 4, 3), (8, 16, 2), (16, 8, 2)])
def test_gae_lookback_with_delayed_publishers(ops, dev, L, W, delay, monkeypatch):
    """the look-back walk reads four granules of two (L >= 8) or four slabs per round with 16-byte loads, all issued and waited for in ONE
    asm statement (round 6; ADVICE r05: the wait was a statement of its own), and trusts a granule only when its own nonce matches.
    ERL_GAE_LB_DELAY makes every second slab sleep `delay` x 128 x 64 clocks before it publishes anything, so readers poll stale tables
    for long and meet aggregates and inclusive values in every mix: long undone chains, results within 1e-5 of oracle/gae_scan.c, no
    timeout, five launches each (nonces advance)."""
    from elegantrl_amd import _hip
    monkeypatch.setenv("ERL_GAE_LB_L", str(L))
    monkeypatch.setenv("ERL_GAE_LB_W", str(W))
    monkeypatch.setenv("ERL_GAE_LB_DELAY", str(delay))
    H, N = 1500, 1028
    r, u, m, v, nv = gae_inputs(H, N, seed=L * 1000 + W * 10 + delay, p_done=0.0003, p_trunc=0.0003)
    adv_o, ret_o, _, _ = c_oracle.gae(r, u, m, v, nv, 0.99, 0.95, use_v_trace=True)
    tm, tv, tnv = cu(m, dev), cu(v, dev), cu(nv, dev)
    for _ in range(5):
        adv, ret = ops.gae_scan(cu(r, dev), cu(u, dev), tm, tv, tnv, 0.99, 0.95, algo="lookback")
        rel_close(adv.cpu().numpy(), adv_o, 1e-5)
        rel_close(ret.cpu().numpy(), ret_o, 1e-5)
    th.cuda.synchronize()
    _hip.check_async_faults()


def test_gae_auto_short_horizon_many_envs_is_within_1e5_of_the_exact_kernel(ops, dev):
    """AUTO sends H < 64 with more than 8192 envs to the ONE-slab look-back form (no granules; 5.3 against 18.9 us at 32 x 32768) and keeps
    the bit-exact lane-per-env kernel up to 8192 envs: the former within 1e-5 of the latter and of oracle/gae_scan.c, the latter bitwise."""
    for H, N, exact in [(32, 32768, False), (32, 8192, True), (48, 16384, False)]:
        r, u, m, v, nv = gae_inputs(H, N, seed=H + N)
        adv_o, ret_o, _, _ = c_oracle.gae(r, u, m, v, nv, 0.99, 0.95, use_v_trace=True)
        t = lambda: (cu(r, dev), cu(u, dev), cu(m, dev), cu(v, dev), cu(nv, dev))   # noqa: E731
        adv_a, ret_a = ops.gae_scan(*t(), 0.99, 0.95)
        adv_e, ret_e = ops.gae_scan(*t(), 0.99, 0.95, algo="exact")
        np.testing.assert_array_equal(adv_e.cpu().numpy(), adv_o)
        if exact:
            np.testing.assert_array_equal(adv_a.cpu().numpy(), adv_o)
            np.testing.assert_array_equal(ret_a.cpu().numpy(), ret_o)
        else:
            rel_close(adv_a.cpu().numpy(), adv_e.cpu().numpy(), 1e-5)
            rel_close(ret_a.cpu().numpy(), ret_o, 1e-5)


def test_cum_rewards_reference_golden_bitwise(ops, dev):
    """erl_cum_rewards_f32 against the returns the reference's AgentBase.get_cumulative_rewards produced (4 appends through
    ReplayBuffer.update_cum_rewards, one of them the p < add_size branch): bit-exact."""
    g = load("cum_rewards.npz")
    gamma = float(g["gamma"][0])
    for k in range(len(g["adds"])):
        p0, p1 = [int(x) for x in g[f"slice{k}"]]
        out = ops.cum_rewards(cu(g[f"rewards{k}"][p0:p1], dev), cu(g[f"undones{k}"][p0:p1], dev), cu(g[f"next_value{k}"], dev), gamma)
        np.testing.assert_array_equal(out.cpu().numpy(), g[f"direct{k}"])


@pytest.mark.parametrize("H,N", [(512, 64), (1, 1), (37, 130), (4096, 4), (9, 1000)])
def test_cum_rewards_matches_numpy_restatement_bitwise(ops, dev, H, N):
    rng = np.random.default_rng(H + N)
    r = rng.standard_normal((H, N), dtype=np.float32)
    u = (rng.random((H, N)) > 0.05).astype(np.float32)
    nv = rng.standard_normal(N, dtype=np.float32)
    ref = O.cum_rewards(r, u, nv, 0.97)
    out = ops.cum_rewards(cu(r, dev), cu(u, dev), cu(nv, dev), 0.97)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("L,W", [(2, 1), (4, 3), (8, 16), (16, 8), (4, 16)])
def test_gae_lookback_every_tiling(ops, dev, L, W, monkeypatch):
    """every (steps per lane, waves per workgroup) instantiation, long undone chains (no episode ends) so that the
    carry really crosses every slab boundary; repeated to catch ordering races in the look-back."""
    monkeypatch.setenv("ERL_GAE_LB_L", str(L))
    monkeypatch.setenv("ERL_GAE_LB_W", str(W))
    H, N = 777, 1028
    r, u, m, v, nv = gae_inputs(H, N, seed=L * 100 + W, p_done=0.0005, p_trunc=0.0005)
    adv_o, ret_o, _, _ = c_oracle.gae(r, u, m, v, nv, 0.99, 0.95, use_v_trace=True)
    tm, tv, tnv = cu(m, dev), cu(v, dev), cu(nv, dev)
    for _ in range(5):
        adv, ret = ops.gae_scan(cu(r, dev), cu(u, dev), tm, tv, tnv, 0.99, 0.95, algo="lookback")
        rel_close(adv.cpu().numpy(), adv_o, 1e-5)
        rel_close(ret.cpu().numpy(), ret_o, 1e-5)


def test_gae_lookback_misaligned_falls_back(ops, dev):
    """N % 4 != 0 cannot use 16-byte rows: the entry point silently uses the chunked scan (same tolerance)."""
    r, u, m, v, nv = gae_inputs(64, 77, seed=3)
    adv_o, _, _, _ = c_oracle.gae(r, u, m, v, nv, 0.99, 0.95, use_v_trace=True)
    adv, _ = ops.gae_scan(cu(r, dev), cu(u, dev), cu(m, dev), cu(v, dev), cu(nv, dev), 0.99, 0.95, algo="lookback")
    rel_close(adv.cpu().numpy(), adv_o, 1e-5)


def test_gae_full_size_property_linearity(ops, dev):
    """BASELINE size (2048 x 4096, 151 MB): with no terminal/truncation GAE is linear in (r, v, next_v);
    check adv(x + y) == adv(x) + adv(y) and agreement of the exact and chunked algorithms."""
    H, N = 2048, 4096
    g = th.Generator(device=dev).manual_seed(0)
    ones = th.ones((H, N), dtype=th.bool, device=dev)

    def run(r, v, nv, algo):
        return ops.gae_scan(r.clone(), ones.clone(), ones, v, nv, 0.99, 0.95, algo=algo)[0]

    r1, v1 = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
    r2, v2 = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
    n1, n2 = th.randn(N, device=dev, generator=g), th.randn(N, device=dev, generator=g)
    a1, a2, a12 = run(r1, v1, n1, "chunked"), run(r2, v2, n2, "chunked"), run(r1 + r2, v1 + v2, n1 + n2, "chunked")
    assert (a12 - (a1 + a2)).abs().max().item() < 2e-4
    e1 = run(r1, v1, n1, "exact")
    err = ((a1 - e1).abs() / e1.abs().clamp_min(1.0)).max().item()
    assert err <= 1e-5, err
    l1, l12 = run(r1, v1, n1, "lookback"), run(r1 + r2, v1 + v2, n1 + n2, "lookback")
    err = ((l1 - e1).abs() / e1.abs().clamp_min(1.0)).max().item()
    assert err <= 1e-5, err
    assert (l12 - (l1 + run(r2, v2, n2, "lookback"))).abs().max().item() < 2e-4


@pytest.mark.parametrize("name", PPO_GOLDENS)
def test_gae_on_reference_golden(ops, dev, name):
    g = load(name)
    hp, d = hyper(g), dims(g)
    tr, tu = cu(g["rewards"], dev), cu(g["undones"], dev)
    adv, ret = ops.gae_scan(tr, tu, cu(g["unmasks"], dev), cu(g["values"], dev), cu(g["next_value"], dev), hp["gamma"],
                            hp["lam"], use_v_trace=d["vtrace"], algo="exact")
    rel_close(adv.cpu().numpy(), g["advantages"], 1e-5)
    rel_close(ret.cpu().numpy(), g["reward_sums"], 1e-5)
    np.testing.assert_array_equal(tu.cpu().numpy(), g["undones_after"])


@pytest.mark.parametrize("H,N", [(32, 4096), (12, 8), (33, 130), (200, 1024)])
@pytest.mark.parametrize("algo", ["exact", "chunked", "lookback"])
def test_adv_stats_and_normalize(ops, dev, H, N, algo):
    r, u, m, v, nv = gae_inputs(H, N, seed=5)
    stats = th.zeros(8, dtype=th.float64, device=dev)
    adv, _ = ops.gae_scan(cu(r, dev), cu(u, dev), cu(m, dev), cu(v, dev), cu(nv, dev), 0.99, 0.95, algo=algo, stats=stats)
    adv_np = adv.cpu().numpy()
    s = stats.cpu().numpy()
    sub = adv_np[::4, ::4].astype(np.float64)
    np.testing.assert_allclose(s[0], adv_np.astype(np.float64).sum(), rtol=1e-9, atol=1e-6)
    assert s[1] == H * N and s[4] == sub.size
    np.testing.assert_allclose(s[2], sub.sum(), rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(s[3], (sub * sub).sum(), rtol=1e-9)
    s2 = ops.adv_stats(adv).cpu().numpy()
    np.testing.assert_allclose(s2[:5], s[:5], rtol=1e-9, atol=1e-6)
    out = ops.adv_normalize(adv, stats)
    rel_close(out.cpu().numpy(), O.adv_normalize(adv_np), 1e-5)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,N,S,A,B", [(32, 4096, 64, 8, 16384), (12, 8, 6, 2, 16), (7, 13, 3, 1, 100)])
def test_ppo_gather_bit_exact(ops, dev, H, N, S, A, B):
    rng = np.random.default_rng(1)
    states = rng.standard_normal((H, N, S), dtype=np.float32)
    actions = rng.standard_normal((H, N, A), dtype=np.float32)
    um = rng.random((H, N)) > 0.1
    lp, adv, rs = (rng.standard_normal((H, N), dtype=np.float32) for _ in range(3))
    ids = rng.integers(0, H * N, size=B).astype(np.int64)
    ids[:3] = [0, H * N - 1, H]                                   # extremes
    (o_s, o_a, o_u, o_l, o_ad, o_r), (i0, i1) = ops.ppo_gather(cu(states, dev), cu(actions, dev), cu(um, dev), cu(lp, dev),
                                                             cu(adv, dev), cu(rs, dev), cu(ids, dev))
    r0, r1 = O.split_ids(ids, H)
    np.testing.assert_array_equal(i0.cpu().numpy(), r0)
    np.testing.assert_array_equal(i1.cpu().numpy(), r1)
    s0, s1 = ops.split_ids(cu(ids, dev), H)
    np.testing.assert_array_equal(s0.cpu().numpy(), r0)
    np.testing.assert_array_equal(s1.cpu().numpy(), r1)
    np.testing.assert_array_equal(o_s.cpu().numpy(), states[r0, r1])
    np.testing.assert_array_equal(o_a.cpu().numpy(), actions[r0, r1])
    np.testing.assert_array_equal(o_u.cpu().numpy(), um[r0, r1])
    np.testing.assert_array_equal(o_l.cpu().numpy(), lp[r0, r1])
    np.testing.assert_array_equal(o_ad.cpu().numpy(), adv[r0, r1])
    np.testing.assert_array_equal(o_r.cpu().numpy(), rs[r0, r1])


@pytest.mark.parametrize("name", ["replay_ring.npz", "replay_ring_discrete.npz"])
def test_replay_ring_against_reference_golden(ops, dev, name):
    """K8 / K9 against the reference's own ReplayBuffer run, continuous and discrete (uint8 action ring, int32 actions in)."""
    g = load(name)
    discrete = "discrete" in name
    max_size, S, A, num_seqs = [int(x) for x in g["dims"]]
    bs = th.zeros((max_size, num_seqs, S), device=dev)
    ba = th.zeros((max_size, num_seqs), dtype=th.uint8, device=dev) if discrete else th.zeros((max_size, num_seqs, A), device=dev)
    br, bu, bm = (th.zeros((max_size, num_seqs), device=dev) for _ in range(3))
    p = 0
    ring = O.Ring(max_size, S, A, num_seqs, if_discrete=discrete)
    for k, add in enumerate(g["adds"]):
        items_np = tuple(g[f"in{k}_{n}"] for n in ("states", "actions", "rewards", "undones", "unmasks"))
        ops.replay_write(bs, ba, br, bu, bm, [cu(x, dev) for x in items_np], p)
        ring.update(items_np)
        p = ring.p
        for t, n in zip((bs, ba, br, bu, bm), ("states", "actions", "rewards", "undones", "unmasks")):
            np.testing.assert_array_equal(t.cpu().numpy(), g[f"buf{k}_{n}"])
        out, (i0, i1) = ops.replay_sample(bs, ba, br, bu, bm, cu(g[f"ids{k}"], dev), ring.cur_size - 1)
        np.testing.assert_array_equal(i0.cpu().numpy(), g[f"ids0_{k}"])
        np.testing.assert_array_equal(i1.cpu().numpy(), g[f"ids1_{k}"])
        for t, n in zip(out, ("state", "action", "reward", "undone", "unmask", "next_state")):
            np.testing.assert_array_equal(t.cpu().numpy(), g[f"out{k}_{n}"])
            assert t.dtype == (th.uint8 if discrete and n == "action" else th.float32)


def test_replay_buffer_class_discrete_and_reused_stage(dev):
    """ReplayBuffer(if_discrete=True) end to end on the reference golden, and sample(reuse=True): same values, same storage."""
    from elegantrl_amd.train import ReplayBuffer
    g = load("replay_ring_discrete.npz")
    max_size, S, _, num_seqs = [int(x) for x in g["dims"]]
    buf = ReplayBuffer(max_size=max_size, state_dim=S, action_dim=1, gpu_id=0, num_seqs=num_seqs, if_discrete=True)
    for t in (buf.states, buf.actions, buf.rewards, buf.undones, buf.unmasks):
        t.zero_()
    assert buf.actions.dtype == th.uint8 and buf.actions.shape == (max_size, num_seqs)
    ptrs = set()
    for k, add in enumerate(g["adds"]):
        buf.update(tuple(cu(g[f"in{k}_{n}"], dev) for n in ("states", "actions", "rewards", "undones", "unmasks")))
        assert [buf.p, buf.cur_size, int(buf.if_full), buf.add_size] == list(g[f"cursor{k}"])
        np.testing.assert_array_equal(buf.actions.cpu().numpy(), g[f"buf{k}_actions"])
        for reuse in (False, True):
            out = buf.sample(16, ids=cu(g[f"ids{k}"], dev), reuse=reuse)
            for t, n in zip(out, ("state", "action", "reward", "undone", "unmask", "next_state")):
                np.testing.assert_array_equal(t.cpu().numpy(), g[f"out{k}_{n}"])
            np.testing.assert_array_equal(buf.ids0.cpu().numpy(), g[f"ids0_{k}"])
        ptrs.add(out[0].data_ptr())
    assert len(ptrs) == 1                     # the reused stage is one allocation for the whole run


def test_replay_large_random_vs_oracle(ops, dev):
    rng = np.random.default_rng(2)
    max_size, S, A, num_seqs = 1000, 11, 3, 4
    ring = O.Ring(max_size, S, A, num_seqs)
    bs = th.zeros((max_size, num_seqs, S), device=dev)
    ba = th.zeros((max_size, num_seqs, A), device=dev)
    br, bu, bm = (th.zeros((max_size, num_seqs), device=dev) for _ in range(3))
    for add in (300, 512, 188, 1, 999, 1000):
        items = (rng.standard_normal((add, num_seqs, S), dtype=np.float32), rng.standard_normal((add, num_seqs, A), dtype=np.float32),
                 rng.standard_normal((add, num_seqs), dtype=np.float32), (rng.random((add, num_seqs)) > 0.1).astype(np.float32),
                 (rng.random((add, num_seqs)) > 0.1).astype(np.float32))
        ops.replay_write(bs, ba, br, bu, bm, [cu(x, dev) for x in items], ring.p)
        ring.update(items)
        np.testing.assert_array_equal(bs.cpu().numpy(), ring.states)
        np.testing.assert_array_equal(bu.cpu().numpy(), ring.undones)
        ids = rng.integers(0, (ring.cur_size - 1) * num_seqs, size=4096).astype(np.int64)
        out, (i0, i1) = ops.replay_sample(bs, ba, br, bu, bm, cu(ids, dev), ring.cur_size - 1)
        ref, (r0, r1) = ring.sample(ids)
        np.testing.assert_array_equal(i0.cpu().numpy(), r0)
        np.testing.assert_array_equal(i1.cpu().numpy(), r1)
        for t, x in zip(out, ref):
            np.testing.assert_array_equal(t.cpu().numpy(), x)


def test_interleaved_ring_against_reference_golden(ops, dev):
    """K8 / K9 on the interleaved block (ops.ReplayRing: [seq][time][state | action | reward | undone | unmask | pad]) against the
    reference's own ReplayBuffer run: the ring's contents THROUGH THE STRIDED VIEWS, the index split and every sampled row bit for bit;
    the planar kernels on the same inputs leave the same bits."""
    g = load("replay_ring.npz")
    max_size, S, A, num_seqs = [int(x) for x in g["dims"]]
    ring = ops.ReplayRing(max_size, num_seqs, S, A, dev)
    assert ring.row_floats == (S + A + 3 + 3) // 4 * 4 and ring.block.shape == (num_seqs, max_size, ring.row_floats)
    assert ring.states.shape == (max_size, num_seqs, S) and ring.actions.shape == (max_size, num_seqs, A) and ring.rewards.shape == (max_size, num_seqs)
    assert ring.states.data_ptr() == ring.block.data_ptr() and not ring.states.is_contiguous()
    ref = O.Ring(max_size, S, A, num_seqs)
    for k, add in enumerate(g["adds"]):
        items_np = tuple(g[f"in{k}_{n}"] for n in ("states", "actions", "rewards", "undones", "unmasks"))
        ring.write([cu(x, dev) for x in items_np], ref.p)
        ref.update(items_np)
        for t, n in zip((ring.states, ring.actions, ring.rewards, ring.undones, ring.unmasks), ("states", "actions", "rewards", "undones", "unmasks")):
            np.testing.assert_array_equal(t.cpu().numpy(), g[f"buf{k}_{n}"])
        out, (i0, i1) = ring.sample(cu(g[f"ids{k}"], dev), ref.cur_size - 1)
        np.testing.assert_array_equal(i0.cpu().numpy(), g[f"ids0_{k}"])
        np.testing.assert_array_equal(i1.cpu().numpy(), g[f"ids1_{k}"])
        for t, n in zip(out, ("state", "action", "reward", "undone", "unmask", "next_state")):
            np.testing.assert_array_equal(t.cpu().numpy(), g[f"out{k}_{n}"])
            assert t.dtype == th.float32
    assert (ring.block[:, :, S + A + 3:] == 0).all()                      # the pad columns are never written


@pytest.mark.parametrize("S,A,num_seqs,max_size", [(11, 3, 4, 1000), (3, 1, 1, 257), (64, 8, 3, 300), (17, 6, 2, 64)])
def test_interleaved_ring_large_random_vs_oracle(ops, dev, S, A, num_seqs, max_size):
    """several row widths (S + A + 3 = 17, 7, 75, 26 floats: with and without pad), wraps, bool and float flags, batches from one
    workgroup's worth to many: writes and samples of the interleaved ring against oracle/ppo_numpy.Ring, bit for bit"""
    rng = np.random.default_rng(S * 100 + A)
    ref = O.Ring(max_size, S, A, num_seqs)
    ring = ops.ReplayRing(max_size, num_seqs, S, A, dev)
    for k, add in enumerate((max_size // 3, max_size // 2 + 1, max_size // 5, 1, max_size - 1, max_size)):
        flags_f32 = k % 2 == 0
        ud, um = rng.random((add, num_seqs)) > 0.1, rng.random((add, num_seqs)) > 0.1
        items = (rng.standard_normal((add, num_seqs, S), dtype=np.float32), rng.standard_normal((add, num_seqs, A), dtype=np.float32),
                 rng.standard_normal((add, num_seqs), dtype=np.float32), ud.astype(np.float32), um.astype(np.float32))
        dev_items = [cu(x, dev) for x in items]
        if not flags_f32:
            dev_items[3], dev_items[4] = cu(ud, dev), cu(um, dev)        # torch.bool, as the rollout produces them
        ring.write(dev_items, ref.p)
        ref.update(items)
        np.testing.assert_array_equal(ring.states.cpu().numpy(), ref.states)
        np.testing.assert_array_equal(ring.actions.cpu().numpy(), ref.actions)
        np.testing.assert_array_equal(ring.undones.cpu().numpy(), ref.undones)
        for B in (7, 256, 5000):
            ids = rng.integers(0, (ref.cur_size - 1) * num_seqs, size=B).astype(np.int64)
            ids[0], ids[-1] = 0, (ref.cur_size - 1) * num_seqs - 1
            out, (i0, i1) = ring.sample(cu(ids, dev), ref.cur_size - 1)
            want, (r0, r1) = ref.sample(ids)
            np.testing.assert_array_equal(i0.cpu().numpy(), r0)
            np.testing.assert_array_equal(i1.cpu().numpy(), r1)
            for t, x in zip(out, want):
                np.testing.assert_array_equal(t.cpu().numpy(), x)


def test_replay_buffer_views_of_the_interleaved_ring_behave_like_the_reference_tensors(dev, tmp_path):
    """the class keeps the reference's attributes (elegantrl/train/replay_buffer.py:40-58) as strided views of the block: shapes, dtypes,
    slicing, advanced indexing and ASSIGNMENT work as on the reference's planar tensors; `args.replay_interleaved = False` keeps planar
    tensors and both layouts sample the same bits; save -> load round-trips across layouts (files hold contiguous rows, not the block)."""
    from elegantrl_amd.train import Config, ReplayBuffer
    max_size, S, A, Q = 50, 11, 3, 4
    planar_args = Config()
    planar_args.replay_interleaved = False
    a = ReplayBuffer(max_size=max_size, state_dim=S, action_dim=A, gpu_id=0, num_seqs=Q)
    b = ReplayBuffer(max_size=max_size, state_dim=S, action_dim=A, gpu_id=0, num_seqs=Q, args=planar_args)
    assert a._ring is not None and b._ring is None and b.states.is_contiguous()
    for buf in (a, b):
        assert buf.states.shape == (max_size, Q, S) and buf.actions.shape == (max_size, Q, A)
        assert buf.rewards.shape == buf.undones.shape == buf.unmasks.shape == buf.cum_rewards.shape == (max_size, Q)
        assert all(t.dtype == th.float32 for t in (buf.states, buf.actions, buf.rewards, buf.undones, buf.unmasks))
    g = th.Generator(device=dev).manual_seed(4)
    for add in (20, 25, 17):                                              # the third one wraps
        items = (th.randn((add, Q, S), device=dev, generator=g), th.randn((add, Q, A), device=dev, generator=g),
                 th.randn((add, Q), device=dev, generator=g), th.rand((add, Q), device=dev, generator=g) > 0.1,
                 th.rand((add, Q), device=dev, generator=g) > 0.1)
        a.update(items)
        b.update(items)
        assert (a.p, a.cur_size, a.if_full) == (b.p, b.cur_size, b.if_full)
        for x, y in zip((a.states, a.actions, a.rewards, a.undones, a.unmasks), (b.states, b.actions, b.rewards, b.undones, b.unmasks)):
            n = a.cur_size
            assert th.equal(x[:n], y[:n])
        ids = th.randint((a.cur_size - 1) * Q, (64,), device=dev, generator=g)
        for x, y in zip(a.sample(64, ids=ids), b.sample(64, ids=ids)):
            assert th.equal(x, y)
        assert th.equal(a.ids0, b.ids0) and th.equal(a.ids1, b.ids1)
    # assignment through a view lands in the block (what run.py-style code that writes buffer.rewards[...] relies on)
    a.rewards[3:5] = 7.0
    a.states[0, 1] = th.arange(S, device=dev, dtype=th.float32)
    assert (a._ring.block[:, 3:5, S + A] == 7.0).all() and th.equal(a._ring.block[1, 0, :S], th.arange(S, device=dev, dtype=th.float32))
    b.rewards[3:5] = 7.0
    b.states[0, 1] = th.arange(S, device=dev, dtype=th.float32)
    # save from the interleaved buffer, load into a planar one and back: unrolled order, contiguous files
    a.save_or_load_history(str(tmp_path), if_save=True)
    st = th.load(str(tmp_path / "replay_buffer_states.pth"))
    assert st.is_contiguous() and st.shape == (a.cur_size, Q, S) and st.untyped_storage().nbytes() == st.numel() * 4
    c = ReplayBuffer(max_size=max_size, state_dim=S, action_dim=A, gpu_id=0, num_seqs=Q, args=planar_args)
    c.save_or_load_history(str(tmp_path), if_save=False)
    d = ReplayBuffer(max_size=max_size, state_dim=S, action_dim=A, gpu_id=0, num_seqs=Q)
    d.save_or_load_history(str(tmp_path), if_save=False)
    assert c.cur_size == d.cur_size == a.cur_size
    for x, y in zip((c.states, c.actions, c.rewards, c.undones), (d.states, d.actions, d.rewards, d.undones)):
        assert th.equal(x[:c.cur_size], y[:c.cur_size])
    # update_cum_rewards reads slices of the views
    a.update_cum_rewards(lambda rewards, undones: rewards * 2 + undones)
    b.update_cum_rewards(lambda rewards, undones: rewards * 2 + undones)
    p1 = a.p if a.p >= a.add_size else a.max_size
    assert th.equal(a.cum_rewards[p1 - a.add_size:p1], b.cum_rewards[p1 - b.add_size:p1])


@pytest.mark.parametrize("num_seqs,max_size", [(1, 1_000_000), (64, 15_625)])
def test_replay_full_size_config3(ops, dev, num_seqs, max_size):
    """BASELINE config 3 ring (1e6 transitions, Hopper-shaped S=11, A=3): size-independent properties.
    (1) index split bit-exact vs integer arithmetic; (2) every sampled row equals the ring row it names (torch advanced
    indexing on the same device buffers) incl. next_state = states[ids0 + 1]; (3) a wrapped write lands in
    [p, max_size) + [0, p'), leaves every other row untouched, and casts bool flags to 0/1 floats."""
    S, A, B = 11, 3, 4096
    g = th.Generator(device=dev).manual_seed(3)
    bs = th.randn((max_size, num_seqs, S), device=dev, generator=g)
    ba = th.randn((max_size, num_seqs, A), device=dev, generator=g)
    br = th.randn((max_size, num_seqs), device=dev, generator=g)
    bu = (th.rand((max_size, num_seqs), device=dev, generator=g) > 0.1).float()
    bm = (th.rand((max_size, num_seqs), device=dev, generator=g) > 0.1).float()
    sample_len = max_size - 1                                    # full ring
    ids = th.randint(sample_len * num_seqs, (B,), device=dev, generator=g)
    ids[0], ids[1] = 0, sample_len * num_seqs - 1                # extremes
    out, (i0, i1) = ops.replay_sample(bs, ba, br, bu, bm, ids, sample_len)
    assert th.equal(i0, ids % sample_len) and th.equal(i1, ids // sample_len)
    for got, buf in zip(out[:5], (bs, ba, br, bu, bm)):
        assert th.equal(got, buf[i0, i1])
    assert th.equal(out[5], bs[i0 + 1, i1])
    # wrapped append of 1000 rows starting 300 rows before the end
    add, p = 1000, max_size - 300
    items = [th.randn((add, num_seqs, S), device=dev, generator=g), th.randn((add, num_seqs, A), device=dev, generator=g),
             th.randn((add, num_seqs), device=dev, generator=g), th.rand((add, num_seqs), device=dev, generator=g) > 0.5,
             th.rand((add, num_seqs), device=dev, generator=g) > 0.5]
    before = bs.clone()
    ops.replay_write(bs, ba, br, bu, bm, items, p)
    assert th.equal(bs[p:], items[0][:300]) and th.equal(bs[:700], items[0][300:])
    assert th.equal(bs[700:p], before[700:p])
    assert th.equal(bu[p:], items[3][:300].float()) and th.equal(bm[:700], items[4][300:].float())
    assert th.equal(br[:700], items[2][300:])
    # the same ring as ONE interleaved block: filled through the views, sampled by the row kernel -- the same bits as the planar sample
    ring = ops.ReplayRing(max_size, num_seqs, S, A, dev)
    ring.states.copy_(bs); ring.actions.copy_(ba); ring.rewards.copy_(br); ring.undones.copy_(bu); ring.unmasks.copy_(bm)   # noqa: E702
    out_p, _ = ops.replay_sample(bs, ba, br, bu, bm, ids, sample_len)
    out_r, (j0, j1) = ring.sample(ids, sample_len)
    assert th.equal(j0, i0) and th.equal(j1, i1)
    for x, y in zip(out_r, out_p):
        assert th.equal(x, y)
    ring.write(items, p)                                                  # (a wrapped append on top of what the planar write left)
    assert th.equal(ring.states, bs) and th.equal(ring.undones, bu) and th.equal(ring.rewards, br)


# ------------------------------------------------------------------------------------------------
def random_net(rng, S, h1, h2, out, with_std):
    def lin(o, i):
        return (rng.standard_normal((o, i)) / np.sqrt(i)).astype(np.float32), (0.1 * rng.standard_normal(o)).astype(np.float32)
    (w1, b1), (w2, b2), (w3, b3) = lin(h1, S), lin(h2, h1), lin(out, h2)
    return O.Mlp([w1, w2, w3], [b1, b2, b3], (0.1 * rng.standard_normal(S)).astype(np.float32),
                 (1.0 + 0.2 * rng.random(S)).astype(np.float32),
                 (-0.3 + 0.1 * rng.standard_normal(out)).astype(np.float32) if with_std else None)


MLP_SHAPES = [(64, 128, 128, 8), (60, 128, 128, 8), (3, 128, 64, 1), (17, 128, 128, 5), (6, 64, 32, 2), (128, 32, 96, 16)]


@pytest.mark.parametrize("S,h1,h2,A", MLP_SHAPES)
@pytest.mark.parametrize("rows", [1, 64, 1000])
def test_value_forward(ops, dev, S, h1, h2, A, rows):
    rng = np.random.default_rng(S + rows)
    net = random_net(rng, S, h1, h2, 1, False)
    x = rng.standard_normal((rows, S), dtype=np.float32)
    spec = ops.MlpSpec(S, h1, h2, 1, False)
    v = ops.value_forward(cu(flat_params(net), dev), spec, cu(net.state_avg, dev), cu(net.state_std, dev), cu(x, dev))
    ref = O.critic_value(x.astype(np.float64), net.astype(np.float64))
    np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("S,h1,h2,A", MLP_SHAPES)
@pytest.mark.parametrize("N", [4096, 37, 16384 + 21])     # <= 16384 envs: latency form (split), above: throughput form
def test_rollout_step_injected_noise(ops, dev, S, h1, h2, A, N):
    rng = np.random.default_rng(S * 7 + N)
    net = random_net(rng, S, h1, h2, A, True)
    x = rng.standard_normal((N, S), dtype=np.float32)
    eps = rng.standard_normal((N, A), dtype=np.float32)
    spec = ops.MlpSpec(S, h1, h2, A, True)
    o_s, o_a, o_e = th.zeros((N, S), device=dev), th.zeros((N, A), device=dev), th.zeros((N, A), device=dev)
    o_l = th.zeros(N, device=dev)
    ops.rollout_step(cu(flat_params(net), dev), spec, cu(net.state_avg, dev), cu(net.state_std, dev), cu(x, dev),
                     noise=cu(eps, dev), out_state=o_s, out_action=o_a, out_logprob=o_l, out_env_action=o_e)
    a_ref, lp_ref = O.actor_sample(x.astype(np.float64), net.astype(np.float64), eps.astype(np.float64))
    np.testing.assert_array_equal(o_s.cpu().numpy(), x)
    np.testing.assert_allclose(o_a.cpu().numpy(), a_ref, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(o_l.cpu().numpy(), lp_ref, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(o_e.cpu().numpy(), np.tanh(a_ref), rtol=1e-4, atol=2e-5)


def test_rollout_step_philox_noise_is_standard_normal_and_deterministic(ops, dev):
    S, h1, h2, A, N = 64, 128, 128, 8, 4096
    rng = np.random.default_rng(0)
    net = random_net(rng, S, h1, h2, A, True)
    net.action_std_log[:] = 0.0
    spec = ops.MlpSpec(S, h1, h2, A, True)
    P, avg, sd = cu(flat_params(net), dev), cu(net.state_avg, dev), cu(net.state_std, dev)
    x = cu(rng.standard_normal((N, S), dtype=np.float32), dev)

    def run(counter):
        o_a, o_l = th.zeros((N, A), device=dev), th.zeros(N, device=dev)
        ops.rollout_step(P, spec, avg, sd, x, seed=123, counter=counter, out_action=o_a, out_logprob=o_l)
        return o_a.cpu().numpy(), o_l.cpu().numpy()

    a0, l0 = run(5)
    a0b, _ = run(5)
    a1, _ = run(6)
    np.testing.assert_array_equal(a0, a0b)
    mean = O.actor_mean(x.cpu().numpy(), net)
    eps0, eps1 = a0 - mean, a1 - mean      # std = 1
    assert abs(eps0.mean()) < 0.02 and abs(eps0.std() - 1.0) < 0.02
    assert abs(np.corrcoef(eps0.ravel(), eps1.ravel())[0, 1]) < 0.02
    assert abs(np.corrcoef(eps0[:, 0], eps0[:, 1])[0, 1]) < 0.06
    np.testing.assert_allclose(l0, O.gaussian_logprob(a0, mean, net.action_std_log), rtol=1e-4, atol=1e-4)


def ppo_case(rng, H, N, S, A, B):
    states = rng.standard_normal((H, N, S), dtype=np.float32)
    actions = rng.standard_normal((H, N, A), dtype=np.float32) * 0.7
    um = rng.random((H, N)) > 0.1
    lp = (-1.0 * A + 0.3 * rng.standard_normal((H, N))).astype(np.float32)
    adv = rng.standard_normal((H, N), dtype=np.float32)
    rs = rng.standard_normal((H, N), dtype=np.float32)
    ids = rng.integers(0, H * N, size=B).astype(np.int64)
    return states, actions, um, lp, adv, rs, ids


def oracle_flat_grads(buf, ids, actor, critic, clip, lam_ent, dt, objective="reference"):
    states, actions, um, lp, adv, rs = buf
    H = states.shape[0]
    i0, i1 = O.split_ids(ids, H)
    s, a = states[i0, i1].astype(dt), actions[i0, i1].astype(dt)
    oc, gw, gb = O.critic_objective(s, rs[i0, i1].astype(dt), um[i0, i1], critic.astype(dt))
    gc = np.concatenate([x.reshape(-1) for pair in zip(gw, gb) for x in pair])
    os_, oe, gw, gb, gsl = O.actor_objective(s, a, lp[i0, i1].astype(dt), adv[i0, i1].astype(dt), um[i0, i1], actor.astype(dt),
                                             clip, lam_ent, objective)
    ga = np.concatenate([x.reshape(-1) for pair in zip(gw, gb) for x in pair] + [gsl.reshape(-1)])
    return ga, gc, np.array([oc, os_, oe], dtype=np.float64)


# more shapes of the one-wave-per-SIMD kernel (S <= 64, h1, h2 in {64, 128}, A <= 8): one K tile (S <= 32), action counts that
# do not fill the 4 x 4 output blocks or the 16-byte action load, a single action; round 3: the other three (h1, h2) pairs
# (two waves per row tile in the weight gradients, W2 / W3 copied through registers instead of LDS-DMA) and unaligned state
# rows (S % 4 != 0: MLP_SHAPES' Pendulum shape (3, 128, 64, 1) and (17, 128, 128, 5) take this kernel's element-wise loaders)
W4_SHAPES = [(32, 128, 128, 3), (16, 128, 128, 4), (64, 128, 128, 1), (4, 128, 128, 8), (48, 128, 128, 6),
             (64, 128, 64, 8), (60, 128, 64, 8), (32, 64, 128, 4), (12, 64, 64, 3), (3, 64, 64, 1), (7, 64, 128, 2), (33, 128, 64, 5)]


@pytest.mark.parametrize("S,h1,h2,A", MLP_SHAPES + W4_SHAPES)
@pytest.mark.parametrize("B", [64, 200, 1024, 1000, 1])
def test_ppo_step_gradients(ops, dev, S, h1, h2, A, B):
    rng = np.random.default_rng(S + B)
    n_slabs = ops.ppo_num_slabs(B)
    assert n_slabs == (B + 127) // 128
    H, N = 9, 50
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = buf_ids[:6], buf_ids[6]
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    stride = ops.ppo_slab_stride(S, h1, h2, A)
    Pa, Pc = ops.MlpSpec(S, h1, h2, A, True).count, ops.MlpSpec(S, h1, h2, 1, False).count
    assert stride == (Pa + Pc + 4 + 31) // 32 * 32
    slabs = th.full((n_slabs, stride), float("nan"), device=dev)
    flat = th.zeros(stride, device=dev)
    ops.ppo_step(cu(flat_params(actor), dev), cu(flat_params(critic), dev), cu(actor.state_avg, dev), cu(actor.state_std, dev),
                 cu(critic.state_avg, dev), cu(critic.state_std, dev), S, h1, h2, A, *[cu(x, dev) for x in buf], cu(ids, dev),
                 0.25, 0.001, 1.0 / B, slabs, n_slabs)
    ops.grad_reduce(slabs, n_slabs, stride, flat)
    got = flat.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all(), "a slab slot was left unwritten"
    ga, gc, objs = oracle_flat_grads(buf, ids, actor, critic, 0.25, 0.001, np.float64)
    for name, g, ref in (("actor", got[:Pa], ga), ("critic", got[Pa:Pa + Pc], gc)):
        scale = np.abs(ref).max()
        err = np.abs(g - ref).max()
        assert err <= 1e-4 * scale + 1e-7, f"{name} grad err {err:.3e} (scale {scale:.3e})"
    np.testing.assert_allclose(got[Pa + Pc:Pa + Pc + 3], objs, rtol=1e-4, atol=1e-6)


S3_SHAPES = [(64, 128, 128, 8), (32, 128, 128, 3), (4, 128, 128, 8), (48, 128, 128, 6), (17, 128, 128, 5), (3, 128, 128, 1), (33, 128, 128, 2),
             (3, 128, 64, 1), (64, 128, 64, 8), (60, 128, 64, 8), (32, 64, 128, 4), (7, 64, 128, 2), (12, 64, 64, 3), (3, 64, 64, 1), (33, 128, 64, 5)]


@pytest.mark.parametrize("S,h1,h2,A", S3_SHAPES)
@pytest.mark.parametrize("B", [200, 1024, 1])
def test_ppo_step_split_arith(ops, dev, S, h1, h2, A, B):
    """K6 on the bf16 matrix pipe (every operand split into three bf16 parts, six partial products, fp32 accumulation:
    csrc/ppo_step_s3_impl.h) against the fp64 restatement, next to the fp32-MFMA kernel on the same inputs: the split kernel
    must be as close to fp64 as the fp32 one (its error may not exceed twice the fp32 kernel's, floor 1e-6 of the gradient's
    scale: the fp32 kernel's own errors on this set reach 8.5e-7; one bf16 rounding would be 4e-3) -- "fp32-equivalent" is
    asserted, not assumed."""
    rng = np.random.default_rng(7 * S + B)
    n_slabs, stride = ops.ppo_num_slabs(B), ops.ppo_slab_stride(S, h1, h2, A)
    H, N = 9, 50
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = buf_ids[:6], buf_ids[6]
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    Pa, Pc = ops.MlpSpec(S, h1, h2, A, True).count, ops.MlpSpec(S, h1, h2, 1, False).count
    ga, gc, objs = oracle_flat_grads(buf, ids, actor, critic, 0.25, 0.001, np.float64)
    errs = {}
    prev = ops.ppo_set_arith("f32")
    try:
        for arith in ("f32", "split"):
            ops.ppo_set_arith(arith)
            assert ops.ppo_arith_in_use(S, h1, h2, A) == arith
            slabs = th.full((n_slabs, stride), float("nan"), device=dev)
            flat = th.zeros(stride, device=dev)
            ops.ppo_step(cu(flat_params(actor), dev), cu(flat_params(critic), dev), cu(actor.state_avg, dev), cu(actor.state_std, dev),
                         cu(critic.state_avg, dev), cu(critic.state_std, dev), S, h1, h2, A, *[cu(x, dev) for x in buf], cu(ids, dev),
                         0.25, 0.001, 1.0 / B, slabs, n_slabs)
            ops.grad_reduce(slabs, n_slabs, stride, flat)
            got = flat.cpu().numpy().astype(np.float64)
            assert np.isfinite(got).all(), f"{arith}: a slab slot was left unwritten"
            assert not np.any(got[Pa + Pc + 4:])
            errs[arith] = [np.abs(got[:Pa] - ga).max() / max(1e-30, np.abs(ga).max()), np.abs(got[Pa:Pa + Pc] - gc).max() / max(1e-30, np.abs(gc).max()),
                           np.abs(got[Pa + Pc:Pa + Pc + 3] - objs).max() / max(1e-30, np.abs(objs).max())]
    finally:
        ops.ppo_set_arith(prev)
    print(f"S={S} A={A} B={B}: max error / scale vs fp64 (actor grad, critic grad, objectives): f32 MFMA {errs['f32']}, split bf16 {errs['split']}")
    for name, e32, es in zip(("actor grad", "critic grad", "objectives"), errs["f32"], errs["split"]):
        assert es <= max(2.0 * e32, 1e-6), f"{name}: split arithmetic error {es:.3e} against the fp32 kernel's {e32:.3e}"


@pytest.mark.parametrize("B", [1, 200, 128 * 4, 128 * 5 + 17, 128 * 7, 128 * 11 + 1, 128 * 16, 128 * 35 + 3, 16384])
def test_ppo_step_workgroup_maps_agree(ops, dev, B, monkeypatch):
    """the split-arithmetic minibatch kernel under both workgroup maps (csrc/ppo_step.h k6_wg_map: network = blockIdx.y against one
    network per XCD, with the halves fallback for a slab count that is no multiple of 4): which workgroup computes a slab must not show in
    the slab -- every slab bit for bit the same, none left unwritten."""
    S, h1, h2, A, H, N = 64, 128, 128, 8, 9, 2000
    rng = np.random.default_rng(B)
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = buf_ids[:6], buf_ids[6]
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    n_slabs, stride = ops.ppo_num_slabs(B), ops.ppo_slab_stride(S, h1, h2, A)
    got = {}
    prev = ops.ppo_set_arith("split")
    try:
        for wg_map in ("0", "1", "2"):
            monkeypatch.setenv("ERL_K6_WG_MAP", wg_map)           # read per launch
            slabs = th.full((n_slabs, stride), float("nan"), device=dev)
            ops.ppo_step(cu(flat_params(actor), dev), cu(flat_params(critic), dev), cu(actor.state_avg, dev), cu(actor.state_std, dev),
                         cu(critic.state_avg, dev), cu(critic.state_std, dev), S, h1, h2, A, *[cu(x, dev) for x in buf], cu(ids, dev),
                         0.25, 0.001, 1.0 / B, slabs, n_slabs)
            got[wg_map] = slabs.cpu().numpy()
            assert np.isfinite(got[wg_map]).all(), f"map {wg_map}: a slab slot was left unwritten"
            assert _hip.ppo_wg_map_info()["forced"] == int(wg_map) and _hip.ppo_wg_map_info()["map"] == int(wg_map)
    finally:
        ops.ppo_set_arith(prev)
    assert np.array_equal(got["0"].view(np.uint32), got["1"].view(np.uint32)) and np.array_equal(got["0"].view(np.uint32), got["2"].view(np.uint32))


def test_ppo_step_workgroup_map_is_measured_once(ops, dev, monkeypatch):
    """no override: the first full-chip launch on a device measures both maps and the device keeps one (include/erl_hip.h,
    erl_ppo_wg_map_info); the launch that did the measuring leaves the same slabs as a forced-map launch"""
    monkeypatch.delenv("ERL_K6_WG_MAP", raising=False)
    S, h1, h2, A, H, N, B = 64, 128, 128, 8, 9, 2000, 16384
    rng = np.random.default_rng(5)
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = buf_ids[:6], buf_ids[6]
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    n_slabs, stride = ops.ppo_num_slabs(B), ops.ppo_slab_stride(S, h1, h2, A)
    args = lambda slabs: (cu(flat_params(actor), dev), cu(flat_params(critic), dev), cu(actor.state_avg, dev), cu(actor.state_std, dev),
                          cu(critic.state_avg, dev), cu(critic.state_std, dev), S, h1, h2, A, *[cu(x, dev) for x in buf], cu(ids, dev),
                          0.25, 0.001, 1.0 / B, slabs, n_slabs)
    prev = ops.ppo_set_arith("split")
    try:
        a = th.full((n_slabs, stride), float("nan"), device=dev)
        ops.ppo_step(*args(a))
        th.cuda.synchronize()
        info = _hip.ppo_wg_map_info()
        print("workgroup map on this box:", info)
        assert info["forced"] is None and info["map"] in (0, 2)
        assert info["us_map0"] and info["us_map2"] and 10.0 < info["us_map0"] < 500.0 and 10.0 < info["us_map2"] < 500.0
        if abs(info["us_map2"] - 0.97 * info["us_map0"]) > 0.02:            # (the reported times are rounded to 0.01 us)
            assert info["map"] == (2 if info["us_map2"] < 0.97 * info["us_map0"] else 0)
        monkeypatch.setenv("ERL_K6_WG_MAP", "0")
        b = th.full((n_slabs, stride), float("nan"), device=dev)
        ops.ppo_step(*args(b))
        assert th.equal(a.view(th.int32), b.view(th.int32))
        monkeypatch.delenv("ERL_K6_WG_MAP")
        assert _hip.ppo_wg_map_info() == info                      # decided once
    finally:
        ops.ppo_set_arith(prev)


def test_update_loop_code_touch_changes_nothing(ops, dev, monkeypatch):
    """the update loop's first minibatch-kernel launch may start by pulling the kernel's own code into the XCDs' L2s with data loads
    (csrc/ppo_step.h k6_code_touch: devices with a slow instruction-cache miss path; ERL_K6_CODE_TOUCH=1 forces it): a full-chip loop with
    the touch forced on leaves the same weights, moments and gradient rows as one with it forced off, under every workgroup map"""
    S, h1, h2, A, H, N, B, T = 64, 128, 128, 8, 9, 4000, 16384, 3
    rng = np.random.default_rng(99)
    buf = ppo_case(rng, H, N, S, A, B)[:6]
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)
    ids = cu(rng.integers(0, H * N, (T, B)), dev)
    P0 = cu(np.concatenate([flat_params(actor), flat_params(critic)]), dev)
    norm = [cu(actor.state_avg, dev), cu(actor.state_std, dev), cu(critic.state_avg, dev), cu(critic.state_std, dev)]
    tb = [cu(x, dev) for x in buf]
    out = {}
    prev = ops.ppo_set_arith("split")
    try:
        for touch, wg_map in (("0", "0"), ("1", "0"), ("1", "2"), ("1", "1")):
            monkeypatch.setenv("ERL_K6_CODE_TOUCH", touch)
            monkeypatch.setenv("ERL_K6_WG_MAP", wg_map)
            P, M1, M2 = P0.clone(), th.zeros_like(P0), th.zeros_like(P0)
            slabs, rows = th.full((n_slabs, stride), float("nan"), device=dev), th.full((T, stride), float("nan"), device=dev)
            for rep in range(2):                                  # two loops: the second one's first launch touches with the range known
                ops.ppo_update(P, M1, M2, *norm, S, h1, h2, A, *tb, ids, 0.25, 0.001, slabs, rows, 1 + T * rep, 1e-3, 3.0)
            th.cuda.synchronize()
            _hip.check_async_faults()
            out[(touch, wg_map)] = [x.cpu().numpy().view(np.uint32) for x in (P, M1, M2, rows)]
            assert np.isfinite(out[(touch, wg_map)][0].view(np.float32)).all()
    finally:
        ops.ppo_set_arith(prev)
    ref = out[("0", "0")]
    for key, got in out.items():
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), key


@pytest.mark.parametrize("S,A,B,h1,h2,N", [(64, 8, 16384, 128, 128, 4000), (64, 8, 1000, 128, 128, 300), (17, 5, 300, 128, 128, 77), (3, 1, 4096, 128, 64, 500),
                                            (32, 3, 260, 64, 128, 100), (20, 4, 129, 64, 64, 50)])
def test_update_loop_two_chains_equal_one_chain(ops, dev, monkeypatch, S, A, B, h1, h2, N):
    """csrc/comm.cpp, round 6: the actor's and the critic's minibatches share nothing (own gradient, own clip norm, own Adam step:
    elegantrl/agents/AgentPPO.py:196-204), so the update loop runs them as TWO chains of half-chip launches on two streams
    (ERL_PPO_CHAINS=2, the default where it applies; one-network minibatch kernel, one-group slab reduction, one-group clip + Adam).
    Same kernels, same associations: weights, both moments and every gradient row (the logged sums included) BIT FOR BIT against the
    one-chain loop, over two loops of three minibatches, full chip and ragged shapes, the actor | critic seam inside a 64-element chunk."""
    H, T = 9, 3
    rng = np.random.default_rng(S * 1000 + B)
    buf = ppo_case(rng, H, N, S, A, B)[:6]
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)
    ids = cu(rng.integers(0, H * N, (T, B)), dev)
    P0 = cu(np.concatenate([flat_params(actor), flat_params(critic)]), dev)
    norm = [cu(actor.state_avg, dev), cu(actor.state_std, dev), cu(critic.state_avg, dev), cu(critic.state_std, dev)]
    tb = [cu(x, dev) for x in buf]
    out, took = {}, {}
    prev = ops.ppo_set_arith("split")
    monkeypatch.setenv("ERL_K6_WG_MAP", "0")           # (a device that keeps map 2 stays on one chain: decided here, not measured)
    try:
        for chains in ("1", "2"):
            monkeypatch.setenv("ERL_PPO_CHAINS", chains)
            P, M1, M2 = P0.clone(), th.zeros_like(P0), th.zeros_like(P0)
            slabs, rows = th.full((n_slabs, stride), float("nan"), device=dev), th.full((2 * T, stride), float("nan"), device=dev)
            for rep in range(2):
                ops.ppo_update(P, M1, M2, *norm, S, h1, h2, A, *tb, ids, 0.25, 0.001, slabs, rows[T * rep:], 1 + T * rep, 1e-3, 3.0)
            th.cuda.synchronize()
            _hip.check_async_faults()
            took[chains] = _hip.ppo_update_chains()
            out[chains] = [x.cpu().numpy().view(np.uint32) for x in (P, M1, M2, rows)]
            assert np.isfinite(out[chains][0].view(np.float32)).all() and np.isfinite(out[chains][3].view(np.float32)).all()
    finally:
        ops.ppo_set_arith(prev)
    assert took == {"1": 1, "2": 2}, took
    for name, a, b in zip(("weights", "exp_avg", "exp_avg_sq", "gradient rows"), out["1"], out["2"]):
        assert np.array_equal(a, b), name
    assert np.abs(out["2"][0].view(np.float32) - P0.cpu().numpy()).max() > 1e-4


def test_ppo_step_split_arith_at_benchmark_size(ops, dev):
    """BASELINE configs[3] at full size (4096 envs x 32 steps, minibatch 16384, obs 64, act 8, net [128,128]): the split-arithmetic
    kernel and the fp32-MFMA kernel on the same minibatch, both against the fp64 restatement -- 128 gradient slabs per network
    summed in a fixed order; the two kernels may differ from fp64 (and from each other) by fp32 rounding only."""
    S, h1, h2, A, H, N, B = 64, 128, 128, 8, 32, 4096, 16384
    rng = np.random.default_rng(2024)
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = buf_ids[:6], buf_ids[6]
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    Pa, Pc = ops.MlpSpec(S, h1, h2, A, True).count, ops.MlpSpec(S, h1, h2, 1, False).count
    n_slabs, stride = ops.ppo_num_slabs(B), ops.ppo_slab_stride(S, h1, h2, A)
    assert n_slabs == 128
    ga, gc, objs = oracle_flat_grads(buf, ids, actor, critic, 0.25, 0.001, np.float64)
    ref = np.concatenate([ga, gc])
    got = {}
    prev = ops.ppo_set_arith("f32")
    try:
        for arith in ("f32", "split"):
            ops.ppo_set_arith(arith)
            slabs = th.full((n_slabs, stride), float("nan"), device=dev)
            flat = th.zeros(stride, device=dev)
            ops.ppo_step(cu(flat_params(actor), dev), cu(flat_params(critic), dev), cu(actor.state_avg, dev), cu(actor.state_std, dev),
                         cu(critic.state_avg, dev), cu(critic.state_std, dev), S, h1, h2, A, *[cu(x, dev) for x in buf], cu(ids, dev),
                         0.25, 0.001, 1.0 / B, slabs, n_slabs)
            ops.grad_reduce(slabs, n_slabs, stride, flat)
            got[arith] = flat.cpu().numpy().astype(np.float64)
            assert np.isfinite(got[arith]).all()
    finally:
        ops.ppo_set_arith(prev)
    scale_a, scale_c = np.abs(ga).max(), np.abs(gc).max()
    for arith, g in got.items():
        ea, ec = np.abs(g[:Pa] - ga).max() / scale_a, np.abs(g[Pa:Pa + Pc] - gc).max() / scale_c
        eo = np.abs(g[Pa + Pc:Pa + Pc + 3] - objs).max() / np.abs(objs).max()
        print(f"{arith}: actor grad {ea:.2e}, critic grad {ec:.2e}, objectives {eo:.2e} of the scale against fp64")
        assert ea < 1e-6 and ec < 1e-6 and eo < 1e-6, (arith, ea, ec, eo)
    d = np.abs(got["split"][:Pa + Pc] - got["f32"][:Pa + Pc])
    assert d[:Pa].max() / scale_a < 1e-6 and d[Pa:].max() / scale_c < 1e-6


@pytest.mark.parametrize("S,A,B,h1,h2", [(64, 8, 512, 128, 128), (32, 3, 200, 128, 128), (17, 5, 300, 128, 128), (3, 1, 128, 128, 128),
                                          (3, 1, 256, 128, 64), (64, 8, 300, 64, 128), (20, 4, 200, 64, 64)])
def test_update_loop_split_arith_images(ops, dev, S, A, B, h1, h2):
    """the C update loop under split arithmetic hands the minibatch kernel pre-split W2 images (built once per loop, refreshed by
    clip + Adam element by element, copied into LDS by DMA) where the stand-alone erl_ppo_step_f32 splits W2 itself: both must
    leave the same bits -- weights, moments, gradient rows.  Next to it the fp32-MFMA loop: after six Adam steps the weights agree
    to a few 1e-6 except where a gradient sits at rounding level (Adam's update is ~lr sign(g) there)."""
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
    out = {}
    prev = ops.ppo_set_arith("split")
    try:
        # (a) python loop, stand-alone minibatch kernel
        P, M1, M2 = P0.clone(), th.zeros_like(P0), th.zeros_like(P0)
        slabs, rows = th.zeros((n_slabs, stride), device=dev), th.zeros((T, stride), device=dev)
        for k in range(T):
            ops.ppo_step(P[:Pa], P[Pa:], *norm, S, h1, h2, A, *tb, ids[k], 0.25, 0.001, 1.0 / B, slabs, n_slabs)
            ops.grad_reduce_partials(slabs, n_slabs, stride, rows[k], groups)
            ops.clip_adam_partials(P, rows[k], M1, M2, stride, groups, k + 1, 1e-3, 3.0)
        out["python"] = (P, M1, M2, rows)
        # (b) the C loop with images, (c) the C loop on the fp32 MFMA
        for name, arith in (("c", "split"), ("f32", "f32")):
            ops.ppo_set_arith(arith)
            P, M1, M2 = P0.clone(), th.zeros_like(P0), th.zeros_like(P0)
            slabs, rows = th.zeros((n_slabs, stride), device=dev), th.zeros((T, stride), device=dev)
            ops.ppo_update(P, M1, M2, *norm, S, h1, h2, A, *tb, ids, 0.25, 0.001, slabs, rows, 1, 1e-3, 3.0)
            out[name] = (P, M1, M2, rows)
    finally:
        ops.ppo_set_arith(prev)
    for x, y, what in zip(out["python"], out["c"], ("weights", "exp_avg", "exp_avg_sq", "gradient rows")):
        assert th.equal(x, y), f"{what}: the loop with W2 images differs from the stand-alone kernel (max {float((x - y).abs().max()):.3e})"
    dp = (out["c"][0] - out["f32"][0]).abs().cpu().numpy()
    moved = float((out["c"][0] - P0).abs().max())
    assert moved > 1e-3
    assert np.mean(dp > 2e-5) < 0.01 and dp.max() <= 2.1 * T * 1e-3, f"split vs fp32 weights: {np.mean(dp > 2e-5):.4f} of the elements differ by > 2e-5, max {dp.max():.3e}"
    gr = (out["c"][3] - out["f32"][3]).abs().max().item() / out["f32"][3].abs().max().item()
    assert gr < 1e-4, f"gradient rows: split vs fp32 {gr:.3e} of the scale"


OBJECTIVES = {"canonical": 1, "a2c": 2}


@pytest.mark.parametrize("S,h1,h2,A", [(64, 128, 128, 8), (3, 128, 64, 1), (128, 96, 128, 5), (6, 64, 32, 2)],
                         ids=["one-wave-per-SIMD kernel", "one-wave-per-SIMD kernel, Pendulum shape", "8-wave kernel", "8-wave kernel, small net"])
@pytest.mark.parametrize("objective", list(OBJECTIVES))
def test_ppo_step_objective_forms(ops, dev, S, h1, h2, A, objective):
    """the textbook clipped surrogate (SURVEY App. A1 / helloworld_PPO_single_file.py:337-339) and AgentA2C's un-clipped
    objective (AgentPPO.py:296-303) through both K6 kernels, against the fp64 restatement (itself checked against torch
    autograd of the quoted expressions and, for A2C, against the reference's own run: tests/test_oracle_golden.py)."""
    rng = np.random.default_rng(S + len(objective))
    B, H, N = 300, 9, 50
    buf_ids = ppo_case(rng, H, N, S, A, B)
    buf, ids = list(buf_ids[:6]), buf_ids[6]
    buf[3] = (buf[3] + 0.5 * rng.standard_normal(buf[3].shape)).astype(np.float32)     # ratios on both sides of the clip
    actor, critic = random_net(rng, S, h1, h2, A, True), random_net(rng, S, h1, h2, 1, False)
    stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)
    Pa, Pc = ops.MlpSpec(S, h1, h2, A, True).count, ops.MlpSpec(S, h1, h2, 1, False).count
    slabs = th.full((n_slabs, stride), float("nan"), device=dev)
    flat = th.zeros(stride, device=dev)
    ops.ppo_step(cu(flat_params(actor), dev), cu(flat_params(critic), dev), cu(actor.state_avg, dev), cu(actor.state_std, dev),
                 cu(critic.state_avg, dev), cu(critic.state_std, dev), S, h1, h2, A, *[cu(x, dev) for x in buf], cu(ids, dev),
                 0.25, 0.001, 1.0 / B, slabs, n_slabs, objective=OBJECTIVES[objective])
    ops.grad_reduce(slabs, n_slabs, stride, flat)
    got = flat.cpu().numpy().astype(np.float64)
    assert not np.any(got[Pa + Pc + 4:])                      # the row's pad to whole 128-byte lines is written (as zeros)
    ga, gc, objs = oracle_flat_grads(buf, ids, actor, critic, 0.25, 0.001, np.float64, objective)
    for name, g, ref in (("actor", got[:Pa], ga), ("critic", got[Pa:Pa + Pc], gc)):
        scale = np.abs(ref).max()
        assert np.abs(g - ref).max() <= 1e-4 * scale + 1e-7, f"{name} grad err {np.abs(g - ref).max():.3e} (scale {scale:.3e})"
    np.testing.assert_allclose(got[Pa + Pc:Pa + Pc + 3], objs, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("n_a,n_c", [(25872, 24961), (279556, 70662), (70000, 5)])
def test_clip_adam_matches_oracle(ops, dev, n_a, n_c):
    """(25872, 24961): one element per thread; groups beyond 64 Ki elements take the partial-norm + grid-wait kernel (SAC's
    critic ensemble and actor sizes), also next to a short group in the same launch."""
    rng = np.random.default_rng(9)
    p = rng.standard_normal(n_a + n_c).astype(np.float32)
    P, M1, M2 = cu(p, dev), th.zeros(n_a + n_c, device=dev), th.zeros(n_a + n_c, device=dev)
    pa, pc = [p[:n_a].copy()], [p[n_a:].copy()]
    sa, sc = O.AdamState(), O.AdamState()
    for step in range(1, 6):
        scale = 10.0 if step % 2 else 0.01                 # exercise both clipped and unclipped
        g = (scale * rng.standard_normal(n_a + n_c) / 100).astype(np.float32)
        ops.clip_adam(P, cu(g, dev), M1, M2, [(0, n_a), (n_a, n_c)], step, 1e-3, 3.0)
        O.optimizer_backward(pa, [g[:n_a]], sa, 1e-3, 3.0)
        O.optimizer_backward(pc, [g[n_a:]], sc, 1e-3, 3.0)
        np.testing.assert_allclose(P.cpu().numpy(), np.concatenate([pa[0], pc[0]]), rtol=0, atol=2e-6)
    th.cuda.synchronize()
    _hip.check_async_faults()


@pytest.mark.parametrize("grid_wait", [False, True])
@pytest.mark.parametrize("n_slabs,Pa,Pc,max_norm", [(128, 25872, 24961, 3.0), (128, 25872, 24961, 1e9), (7, 1000, 37, 0.5), (1, 300, 200, 3.0)])
def test_reduce_clip_adam_equals_the_two_kernel_tail(ops, dev, n_slabs, Pa, Pc, max_norm, grid_wait):
    """erl_reduce_clip_adam_f32 (one launch, last-arriver) and erl_reduce_clip_adam_grid_f32 (one launch, every workgroup waits
    for the norm) against erl_grad_reduce_f32 + erl_clip_adam_f32: the reduced gradient
    bit for bit (same association), parameters and moments to fp32 round-off of the norm (fp64 partial sums in another order),
    over several back-to-back steps (the arrival counter is monotonic across launches)."""
    g = th.Generator(device=dev).manual_seed(n_slabs + Pa)
    stride = Pa + Pc + 4
    groups = [(0, Pa), (Pa, Pc)]
    p0 = th.randn(Pa + Pc, device=dev, generator=g)
    pa, ma, va = p0.clone(), th.zeros_like(p0), th.zeros_like(p0)
    pb, mb, vb = p0.clone(), th.zeros_like(p0), th.zeros_like(p0)
    fa, fb = th.empty(stride, device=dev), th.empty(stride, device=dev)
    if grid_wait:
        assert ops.reduce_clip_adam_grid_ok(stride)
    for step in range(1, 6):
        slabs = th.randn((n_slabs, stride), device=dev, generator=g) * (0.1 if step % 2 else 10.0)
        ops.grad_reduce(slabs, n_slabs, stride, fa)
        ops.clip_adam(pa, fa, ma, va, groups, step, 1e-3, max_norm)
        ops.reduce_clip_adam(slabs, n_slabs, stride, fb, pb, mb, vb, groups, step, 1e-3, max_norm, grid_wait=grid_wait)
        assert th.equal(fa, fb)
        for x, y in ((pa, pb), (ma, mb), (va, vb)):
            np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=2e-6, atol=1e-9)
    assert float((pa - p0).abs().max()) > 1e-4
    th.cuda.synchronize()
    _hip.check_async_faults()


@pytest.mark.parametrize("n_slabs,Pa,Pc,max_norm,scale", [(128, 25872, 24961, 3.0, 1.0), (128, 25872, 24961, 1e9, 0.125), (7, 1000, 37, 0.5, 0.5),
                                                          (1, 300, 200, 3.0, 1.0), (37, 70000, 5, 3.0, 1.0)])
def test_partials_tail_equals_the_two_kernel_tail(ops, dev, n_slabs, Pa, Pc, max_norm, scale):
    """csrc/grad_tail.hip, the default optimiser tail since round 3: erl_grad_reduce_partials_f32 (slab sum + every 256-element
    workgroup's fp64 share of the group norms) + erl_clip_adam_partials_f32, against erl_grad_reduce_f32 + erl_clip_adam_f32:
    the reduced gradient bit for bit (same association), parameters and moments to fp32 round-off of the norm (fp64 partial
    sums in another order); and erl_grad_sq_partials_f32 on the already-summed row (the route behind a foreign all-reduce)
    gives exactly the fused launch's parameters."""
    g = th.Generator(device=dev).manual_seed(n_slabs + Pa)
    stride = (Pa + Pc + 4 + 31) // 32 * 32
    groups = [(0, Pa), (Pa, Pc)]
    p0 = th.randn(Pa + Pc, device=dev, generator=g)
    pa, ma, va = p0.clone(), th.zeros_like(p0), th.zeros_like(p0)
    pb, mb, vb = p0.clone(), th.zeros_like(p0), th.zeros_like(p0)
    pc, mc, vc = p0.clone(), th.zeros_like(p0), th.zeros_like(p0)
    fa, fb, fc = th.empty(stride, device=dev), th.empty(stride, device=dev), th.empty(stride, device=dev)
    for step in range(1, 6):
        slabs = th.randn((n_slabs, stride), device=dev, generator=g) * (0.1 if step % 2 else 10.0)
        ops.grad_reduce(slabs, n_slabs, stride, fa)
        ops.clip_adam(pa, fa, ma, va, groups, step, 1e-3, max_norm, grad_scale=scale)
        ops.grad_reduce_partials(slabs, n_slabs, stride, fb, groups, grad_scale=scale)
        ops.clip_adam_partials(pb, fb, mb, vb, stride, groups, step, 1e-3, max_norm, grad_scale=scale)
        ops.grad_reduce(slabs, n_slabs, stride, fc)
        ops.grad_sq_partials(fc, stride, groups, grad_scale=scale)
        ops.clip_adam_partials(pc, fc, mc, vc, stride, groups, step, 1e-3, max_norm, grad_scale=scale)
        assert th.equal(fa, fb) and th.equal(fa, fc)
        for x, y in ((pa, pb), (ma, mb), (va, vb)):
            np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=2e-6, atol=1e-9)
        assert th.equal(pb, pc) and th.equal(mb, mc) and th.equal(vb, vc)
    assert float((pa - p0).abs().max()) > 1e-4


@pytest.mark.parametrize("n_slabs,Pa,Pc,max_norm,scale", [(128, 25872, 24961, 3.0, 1.0), (128, 25872, 24961, 1e9, 0.125), (7, 1000, 37, 0.5, 0.5),
                                                          (1, 300, 200, 3.0, 1.0), (37, 70000, 5, 3.0, 1.0), (256, 47648, 46337, 3.0, 1.0)])
def test_single_launch_tail_is_bit_identical_to_the_two_launch_tail(ops, dev, n_slabs, Pa, Pc, max_norm, scale):
    """csrc/grad_tail.hip tail_fused_kernel (round 5): the default tail -- erl_grad_reduce_partials_f32 + erl_clip_adam_partials_f32 -- as ONE
    launch: the reduced gradient, parameters and both moments BIT FOR BIT over six steps (gradients alternating in scale so that the clip
    is and is not active), i.e. the same partial norms summed in the same order; no wait of the kernel timed out."""
    from elegantrl_amd import _hip
    g = th.Generator(device=dev).manual_seed(n_slabs + Pa)
    stride = (Pa + Pc + 4 + 31) // 32 * 32
    assert ops.tail_fused_ok(stride)
    groups = [(0, Pa), (Pa, Pc)]
    p0 = th.randn(Pa + Pc, device=dev, generator=g)
    pa, ma, va = p0.clone(), th.zeros_like(p0), th.zeros_like(p0)
    pb, mb, vb = p0.clone(), th.zeros_like(p0), th.zeros_like(p0)
    fa, fb = th.empty(stride, device=dev), th.empty(stride, device=dev)
    for step in range(1, 7):
        slabs = th.randn((n_slabs, stride), device=dev, generator=g) * (0.1 if step % 2 else 10.0)
        ops.grad_reduce_partials(slabs, n_slabs, stride, fa, groups, grad_scale=scale)
        ops.clip_adam_partials(pa, fa, ma, va, stride, groups, step, 1e-3, max_norm, grad_scale=scale)
        ops.reduce_clip_adam_fused(slabs, n_slabs, stride, fb, pb, mb, vb, groups, step, 1e-3, max_norm, grad_scale=scale)
        assert th.equal(fa, fb), f"gradient, step {step}"
        assert th.equal(pa, pb) and th.equal(ma, mb) and th.equal(va, vb), f"parameters / moments, step {step}"
    assert float((pa - p0).abs().max()) > 1e-4
    th.cuda.synchronize()
    _hip.check_async_faults()


def test_single_launch_tail_in_the_update_loop_and_its_timeout(ops, dev, monkeypatch):
    """(i) erl_ppo_update_f32 under ERL_FUSED_TAIL=3 leaves exactly the weights, moments, gradient rows and -- for the split-arithmetic
    minibatch kernel -- weight IMAGES (the next minibatch reads them) of the default two-launch loop: six minibatches on the reference
    golden's shape; (ii) a row too long for the kernel is refused by erl_tail_fused_ok (the loop keeps the two launches)."""
    g = load("ppo_c4shape.npz")
    from tests.test_agent_gpu import make_agent
    res = {}
    for mode in ("0", "3"):
        monkeypatch.setenv("ERL_FUSED_TAIL", mode)
        agent, _ = make_agent(g)
        agent.last_state = th.from_numpy(g["last_state"]).to(dev)
        buf = [th.from_numpy(g[k]).to(dev) for k in ("states", "actions", "logprobs", "rewards", "undones", "unmasks")]
        ids = th.from_numpy(np.concatenate([g["ids"]] * 3)).to(dev)
        agent.repeat_times = ids.shape[0] * agent.batch_size / buf[0].shape[0]
        objs = agent.update_net(buf, ids=ids)
        res[mode] = (agent._flat.clone(), agent._exp_avg.clone(), agent._exp_avg_sq.clone(), agent._grads[:ids.shape[0]].clone(), objs)
    monkeypatch.delenv("ERL_FUSED_TAIL")
    for a, b in zip(res["0"][:4], res["3"][:4]):
        assert th.equal(a, b)
    assert res["0"][4] == res["3"][4]
    assert not ops.tail_fused_ok(64 * 2049) and ops.tail_fused_ok(64 * 2048)


def test_clip_adam_grad_scale_equals_prescaled(ops, dev):
    rng = np.random.default_rng(10)
    n = 5000
    p = rng.standard_normal(n).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    P1, P2 = cu(p, dev), cu(p, dev)
    z = lambda: th.zeros(n, device=dev)
    ops.clip_adam(P1, cu(g, dev), z(), z(), [(0, n)], 1, 1e-3, 3.0, grad_scale=0.125)
    ops.clip_adam(P2, cu(g * np.float32(0.125), dev), z(), z(), [(0, n)], 1, 1e-3, 3.0)
    np.testing.assert_array_equal(P1.cpu().numpy(), P2.cpu().numpy())


@pytest.mark.parametrize("name", PPO_GOLDENS)
def test_update_loop_on_reference_golden(ops, dev, name):
    """gather + fwd/bwd + reduce + clip + Adam for every recorded minibatch of the reference run."""
    g = load(name)
    hp, d = hyper(g), dims(g)
    S, A, h1, h2, B = d["S"], d["A"], d["h1"], d["h2"], d["B"]
    a0, c0 = mlp_from(g, "act0"), mlp_from(g, "cri0")
    Pa, Pc = ops.MlpSpec(S, h1, h2, A, True).count, ops.MlpSpec(S, h1, h2, 1, False).count
    P = cu(np.concatenate([flat_params(a0), flat_params(c0)]), dev)
    M1, M2 = th.zeros_like(P), th.zeros_like(P)
    stride = ops.ppo_slab_stride(S, h1, h2, A)
    n_slabs = ops.ppo_num_slabs(B)
    slabs, flat = th.zeros((n_slabs, stride), device=dev), th.zeros(stride, device=dev)
    buf = [cu(g[k], dev) for k in ("states", "actions", "unmasks", "logprobs", "advantages_norm", "reward_sums")]
    logs = []
    for step, ids in enumerate(g["ids"], start=1):
        ops.ppo_step(P[:Pa], P[Pa:], cu(a0.state_avg, dev), cu(a0.state_std, dev), cu(c0.state_avg, dev), cu(c0.state_std, dev),
                     S, h1, h2, A, *buf, cu(ids, dev), hp["ratio_clip"], hp["lambda_entropy"], 1.0 / B, slabs, n_slabs)
        ops.grad_reduce(slabs, n_slabs, stride, flat)
        logs.append(flat[Pa + Pc:Pa + Pc + 3].cpu().numpy().astype(np.float64))
        ops.clip_adam(P, flat, M1, M2, [(0, Pa), (Pa, Pc)], step, hp["lr"], hp["max_norm"])
    np.testing.assert_allclose(np.mean(logs, axis=0), g["objs"], rtol=2e-4, atol=2e-6)
    ref = np.concatenate([flat_params(mlp_from(g, "act1")), flat_params(mlp_from(g, "cri1"))])
    np.testing.assert_allclose(P.cpu().numpy(), ref, rtol=0, atol=2e-5)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,S,A", [(1000, 64, 8), (4096, 64, 8), (777, 60, 8), (50, 11, 3), (300, 128, 16), (200, 64, 20)])
def test_synenv_step_matches_formula(ops, dev, N, S, A):
    """A <= 16: 16-env MFMA tiles (synenv_tile_kernel); A = 20: the one-wave-per-env form."""
    rng = np.random.default_rng(4)
    s = rng.standard_normal((N, S), dtype=np.float32)
    s[:5] *= 30.0                                             # force terminals
    a = np.tanh(rng.standard_normal((N, A), dtype=np.float32))
    Ws = (0.9 * np.eye(S) + 0.05 * rng.standard_normal((S, S))).astype(np.float32)
    Wa = (0.1 * rng.standard_normal((A, S))).astype(np.float32)
    sc = np.zeros(N, np.int32)
    sc[10:20] = 99                                            # will truncate at max_step = 100
    ts, tsc, tep = cu(s, dev), cu(sc, dev), th.zeros(N, dtype=th.int32, device=dev)
    rew, term, trunc = th.zeros(N, device=dev), th.zeros(N, dtype=th.bool, device=dev), th.zeros(N, dtype=th.bool, device=dev)
    ops.synenv_step(ts, cu(a, dev), cu(Ws, dev), cu(Wa, dev), tsc, tep, rew, term, trunc, 100, 7)
    s2 = s.astype(np.float64) @ Ws + a.astype(np.float64) @ Wa
    np.testing.assert_allclose(rew.cpu().numpy(), -(s2 ** 2).mean(1) - 0.01 * (a.astype(np.float64) ** 2).mean(1), rtol=1e-4, atol=1e-5)
    t_ref = np.abs(s2).max(1) > 10
    np.testing.assert_array_equal(term.cpu().numpy(), t_ref)
    tr_ref = (sc + 1 >= 100) & ~t_ref
    np.testing.assert_array_equal(trunc.cpu().numpy(), tr_ref)
    done = t_ref | tr_ref
    assert done[:5].all() and done[10:20].all()
    out = ts.cpu().numpy()
    np.testing.assert_allclose(out[~done], s2[~done], rtol=1e-4, atol=1e-5)
    assert abs(out[done].mean()) < 0.2 and 0.7 < out[done].std() < 1.3     # fresh N(0,1) rows
    np.testing.assert_array_equal(tsc.cpu().numpy(), np.where(done, 0, sc + 1))
    np.testing.assert_array_equal(tep.cpu().numpy(), done.astype(np.int32))


def test_pendulum_step_matches_gym_formula(ops, dev):
    rng = np.random.default_rng(6)
    N = 512
    phys = np.stack([rng.uniform(-np.pi, np.pi, N), rng.uniform(-8, 8, N)], 1).astype(np.float32)
    act = rng.uniform(-1.5, 1.5, (N, 1)).astype(np.float32)
    tp, obs = cu(phys, dev), th.zeros((N, 3), device=dev)
    sc, ep = th.zeros(N, dtype=th.int32, device=dev), th.zeros(N, dtype=th.int32, device=dev)
    rew, term, trunc = th.zeros(N, device=dev), th.zeros(N, dtype=th.bool, device=dev), th.zeros(N, dtype=th.bool, device=dev)
    ops.pendulum_step(tp, obs, cu(act, dev), sc, ep, rew, term, trunc, 200, 0)
    th_, thd = phys[:, 0].astype(np.float64), phys[:, 1].astype(np.float64)
    u = np.clip(2.0 * act[:, 0].astype(np.float64), -2, 2)
    ang = ((th_ + np.pi) % (2 * np.pi)) - np.pi
    cost = ang ** 2 + 0.1 * thd ** 2 + 0.001 * u ** 2
    nthd = np.clip(thd + (3 * 10.0 / 2 * np.sin(th_) + 3.0 * u) * 0.05, -8, 8)
    nth = th_ + nthd * 0.05
    np.testing.assert_allclose(rew.cpu().numpy(), -0.5 * cost, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(obs.cpu().numpy(), np.stack([np.cos(nth), np.sin(nth), nthd], 1), rtol=1e-4, atol=1e-4)
    assert not term.any().item() and not trunc.any().item()
