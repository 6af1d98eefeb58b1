"""Row f2: prioritised replay.  The reference's SumTree SAMPLING does not run (its own assert fires), so the target there is the corrected
restatement oracle/per_numpy.py: its properties are checked on the CPU, the device trees (csrc/per.hip) against it bit for bit.  The
UPDATE half does run in the reference (SumTree.update_ids, replay_buffer.py:249-258): tests/golden/per_update.npz holds the reference's own
tree tensors (oracle/make_golden.py::make_per_update) and both the restatement and the device trees are compared with them bitwise on
every level the reference recomputes (round 6: f2 "update half pinned")."""
import numpy as np
import pytest
import torch as th

from oracle.per_numpy import PerTrees


def test_reference_sumtree_asserts_which_is_why_parity_is_unpinned():
    """documents the claim in oracle/per_numpy.py: run the reference's SumTree when it is mounted (authoring container only)."""
    import os
    import sys
    ref = os.environ.get("ERL_REFERENCE", "/root/reference")
    if not os.path.isdir(ref):
        pytest.skip("reference not mounted")
    sys.path.insert(0, ref)
    try:
        from elegantrl.train.replay_buffer import SumTree
    finally:
        sys.path.remove(ref)
    for buf_len in (8, 1000, 1024):
        tree = SumTree(buf_len=buf_len)
        tree.update_ids(th.arange(buf_len), prob=th.rand(buf_len) + 0.1)
        with pytest.raises(AssertionError):
            tree.important_sampling(batch_size=16, beg=-buf_len, end=-1, per_beta=0.4)


def test_reference_per_append_raises_on_wrap():
    """why tests/golden/per_update.npz has no wrapping append: the reference's PER branch of ReplayBuffer.update builds
    th.arange(self.p, p) AFTER p was reduced modulo max_size (replay_buffer.py:92,109) and raises (authoring container only)."""
    import os
    import sys
    ref = os.environ.get("ERL_REFERENCE", "/root/reference")
    if not os.path.isdir(ref):
        pytest.skip("reference not mounted")
    sys.path.insert(0, ref)
    try:
        from elegantrl.train.replay_buffer import ReplayBuffer
    finally:
        sys.path.remove(ref)
    buf = ReplayBuffer(max_size=8, state_dim=3, action_dim=2, gpu_id=-1, num_seqs=2, if_use_per=True)
    mk = lambda add: (th.randn(add, 2, 3), th.randn(add, 2, 2), th.randn(add, 2), th.rand(add, 2) > 0.1, th.rand(add, 2) > 0.1)   # noqa: E731
    buf.update(mk(6))
    with pytest.raises(RuntimeError):
        buf.update(mk(5))


def _golden_per_steps():
    """(max_size, Q, depth_levels_updated, [(p0, add, ids0 (Q, n), td (Q, n), prob (Q, n), tree_after_append (Q, 2 m - 1), tree_after_td)])"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "per_update.npz"))
    for ci, (max_size, Q, n_steps) in enumerate(g["cases"]):
        steps = [(int(g[f"c{ci}_s{k}_append"][0]), int(g[f"c{ci}_s{k}_append"][1]), g[f"c{ci}_s{k}_ids0"], g[f"c{ci}_s{k}_td"],
                  g[f"c{ci}_s{k}_prob"], g[f"c{ci}_s{k}_tree_after_append"], g[f"c{ci}_s{k}_tree_after_td"]) for k in range(int(n_steps))]
        yield int(max_size), int(Q), int(g[f"c{ci}_depth"][1]), steps


def _assert_levels_equal(ours_sum, ref_tree, max_size, levels_updated):
    """ours: (Q, 2 L) heap with root 1, leaves at L + row; the reference's: (Q, 2 m - 1) heap with root 0, leaves at m - 1 + row, m = L a
    power of two: node i of theirs is node i + 1 of ours.  The reference recomputes `levels_updated` levels above the leaves (never its
    root): those, and the leaves, must agree bit for bit."""
    L = max_size
    assert ours_sum.shape[1] == 2 * L and ref_tree.shape[1] == 2 * L - 1
    leaf_level = L.bit_length() - 1
    np.testing.assert_array_equal(ours_sum[:, L:], ref_tree[:, L - 1:])
    for level in range(leaf_level - 1, leaf_level - 1 - levels_updated, -1):
        assert level >= 1
        np.testing.assert_array_equal(ours_sum[:, 1 << level:1 << (level + 1)], ref_tree[:, (1 << level) - 1:(1 << (level + 1)) - 1])
    assert (ref_tree[:, 0] == 0).all()           # the reference never reaches its root (depth - 2 iterations): what its sampler then trips over


def test_oracle_tree_update_matches_the_reference_sumtree_bitwise():
    """oracle/per_numpy.py against the reference's own SumTree tensors: appends at priority 10, then td-error updates (distinct rows)."""
    for max_size, Q, levels, steps in _golden_per_steps():
        t = PerTrees(max_size, Q)
        assert t.L == max_size
        for p0, add, ids0, td, prob, tree_append, tree_td in steps:
            t.add_rows(p0, add)
            _assert_levels_equal(t.sum, tree_append, max_size, levels)
            # the priority formula (:168): numpy's powf against torch's -- one ulp at most; the tree arithmetic continues from the reference's
            ours = np.power(np.clip(td, np.float32(1e-8), np.float32(10.0)), np.float32(0.6)).astype(np.float32)
            np.testing.assert_allclose(ours, prob, rtol=5e-7)           # (two libms' powf: a few ulp at 1e-8^0.6)
            n = ids0.shape[1]
            t.set(ids0.reshape(-1), np.repeat(np.arange(Q), n), prob.reshape(-1))
            _assert_levels_equal(t.sum, tree_td, max_size, levels)


@pytest.mark.gpu
def test_device_tree_update_matches_the_reference_sumtree_bitwise():
    """csrc/per.hip (erl_per_add_rows_f32 / erl_per_update_f32) against the reference's own SumTree tensors, level by level, bitwise.  The
    device computes clamp(td)^alpha itself (powf: within a few ulp of torch's, asserted); where a leaf differs by that ulp the comparison of
    the upper levels would fail for that reason alone, so a second pair of trees is updated with alpha = 1 and td = the reference's prob
    (exact: clamp(prob) = prob, x^1 = x), which isolates the tree arithmetic: bitwise on every level the reference recomputes."""
    from elegantrl_amd import ops
    dev = th.device("cuda:0")
    for max_size, Q, levels, steps in _golden_per_steps():
        t, t1 = ops.PerTrees(max_size, Q, dev), ops.PerTrees(max_size, Q, dev)
        for p0, add, ids0, td, prob, tree_append, tree_td in steps:
            n = ids0.shape[1]
            i0 = th.from_numpy(ids0.reshape(-1).astype(np.int64)).to(dev)
            i1 = th.from_numpy(np.repeat(np.arange(Q), n).astype(np.int64)).to(dev)
            for trees in (t, t1):
                trees.add_rows(p0, add)
                _assert_levels_equal(trees.sum.view(Q, -1).cpu().numpy(), tree_append, max_size, levels)
            t.update(i0, i1, th.from_numpy(td.reshape(-1)).to(dev), 0.6)
            got = t.sum.view(Q, -1).cpu().numpy()
            np.testing.assert_allclose(got[:, max_size:], tree_td[:, max_size - 1:], rtol=5e-7)          # leaves: the device's own powf
            t1.update(i0, i1, th.from_numpy(prob.reshape(-1)).to(dev), 1.0)                              # the reference's priorities as they are
            _assert_levels_equal(t1.sum.view(Q, -1).cpu().numpy(), tree_td, max_size, levels)
            if np.array_equal(got[:, max_size:], tree_td[:, max_size - 1:]):
                _assert_levels_equal(got, tree_td, max_size, levels)
            # keep the two device trees on the reference's leaves for the next step
            t.update(i0, i1, th.from_numpy(prob.reshape(-1)).to(dev), 1.0)


def test_oracle_trees_are_consistent_and_sampling_is_proportional():
    rng = np.random.default_rng(0)
    max_size, Q = 37, 3
    t = PerTrees(max_size, Q)
    t.add_rows(0, 30)
    t.td_error_update(rng.integers(0, 30, 40), rng.integers(0, Q, 40), rng.random(40).astype(np.float32) * 12)
    L = t.L
    for q in range(Q):                          # every parent is the sum / min of its children; unwritten leaves 0 / inf
        for node in range(1, L):
            assert t.sum[q, node] == np.float32(t.sum[q, 2 * node] + t.sum[q, 2 * node + 1])
            assert t.min[q, node] == min(t.min[q, 2 * node], t.min[q, 2 * node + 1])
        assert (t.sum[q, L + 30:] == 0).all() and np.isinf(t.min[q, L + 30:]).all()
    # proportional draws: row frequencies follow the priorities (the last filled row's share moves to its neighbour, D4)
    n = 20000
    ids0, ids1, w = t.sample(rng.random((Q, n)).astype(np.float32), cur_size=30)
    assert ids0.min() >= 0 and ids0.max() <= 28 and (ids1 == np.repeat(np.arange(Q), n)).all()
    for q in range(Q):
        pri = t.sum[q, L:L + 30].astype(np.float64)
        expect = pri.copy()
        expect[28] += expect[29]
        expect = expect[:29] / pri.sum()
        freq = np.bincount(ids0[q * n:(q + 1) * n], minlength=29) / n
        assert np.abs(freq - expect).max() < 0.01
    np.testing.assert_allclose(w, (t.sum[ids1, L + ids0] / t.min[ids1, 1]) ** -0.4, rtol=1e-6)
    assert w.max() <= 1.0 + 1e-6                # the least likely transition has weight 1, all others less


@pytest.mark.gpu
@pytest.mark.parametrize("max_size,Q", [(37, 3), (1000, 1), (4096, 8)])
def test_device_trees_match_the_oracle_bitwise(max_size, Q):
    from elegantrl_amd import ops
    dev = th.device("cuda:0")
    rng = np.random.default_rng(max_size)
    ref, t = PerTrees(max_size, Q), ops.PerTrees(max_size, Q, dev)
    cur, p, full = 0, 0, False
    for add in (max_size // 3, max_size // 2, max_size // 4 + 1, 5):       # wraps; the second one takes the bulk (per-level) path
        ref.add_rows(p, add)
        t.add_rows(p, add)
        full = full or p + add > max_size
        p = (p + add) % max_size
        cur = min(max_size, cur + add)
        n = min(64, cur * Q)
        # WITH repeats: a transition drawn twice gets the priority of its LAST td error (oracle D7), deterministically
        flat = rng.choice(cur * Q, size=n, replace=True)
        flat[1::7] = flat[0]
        ids0, ids1 = flat % cur, flat // cur
        td = (rng.random(n) * 12).astype(np.float32)
        ref.td_error_update(ids0, ids1, td)
        # plus two pairs outside the trees, which the device skips (the host cannot validate device-resident ids)
        d0 = np.concatenate([ids0, [max_size + 3, 0]]).astype(np.int64)
        d1 = np.concatenate([ids1, [0, Q]]).astype(np.int64)
        dtd = np.concatenate([td, [5.0, 5.0]]).astype(np.float32)
        t.update(th.from_numpy(d0).to(dev), th.from_numpy(d1).to(dev), th.from_numpy(dtd).to(dev), 0.6)
        got_sum, got_min = t.sum.view(Q, -1).cpu().numpy(), t.min.view(Q, -1).cpu().numpy()
        np.testing.assert_allclose(got_sum[:, ref.L:], ref.sum[:, ref.L:], rtol=2e-7)        # powf: one ulp
        leaves_equal = np.array_equal(got_sum[:, ref.L:], ref.sum[:, ref.L:])
        if not leaves_equal:                     # continue from the device's leaves so that the tree arithmetic is compared exactly
            ref.sum[:, ref.L:], ref.min[:, ref.L:] = got_sum[:, ref.L:], got_min[:, ref.L:]
            for node in range(ref.L - 1, 0, -1):
                ref.sum[:, node] = ref.sum[:, 2 * node] + ref.sum[:, 2 * node + 1]
                ref.min[:, node] = np.minimum(ref.min[:, 2 * node], ref.min[:, 2 * node + 1])
        np.testing.assert_array_equal(got_sum[:, 1:], ref.sum[:, 1:])
        np.testing.assert_array_equal(got_min[:, 1:], ref.min[:, 1:])
        u = rng.random((Q, 48)).astype(np.float32)
        cursor = p if full else -1
        idx, w = t.sample(th.from_numpy(u).to(dev), cur, 0.4, cursor=cursor)
        r0, r1, rw = ref.sample(u, cur, cursor=cursor)
        np.testing.assert_array_equal(idx.cpu().numpy(), r1 * cur + r0)
        np.testing.assert_allclose(w.cpu().numpy(), rw, rtol=2e-6)
        if full:                                 # D6: the newest row (its successor slot holds the oldest data) is never drawn
            assert ((p - 1) % max_size) not in set(r0.tolist())
    assert full


@pytest.mark.gpu
def test_large_update_list_with_duplicates_takes_the_bulk_path():
    """more than 8192 td errors in one list (the per-level launches): duplicates resolved like the single-workgroup kernel"""
    from elegantrl_amd import ops
    dev = th.device("cuda:0")
    rng = np.random.default_rng(5)
    max_size, Q, n = 3000, 4, 9000
    ref, t = PerTrees(max_size, Q), ops.PerTrees(max_size, Q, dev)
    ref.add_rows(0, max_size)
    t.add_rows(0, max_size)
    ids0, ids1 = rng.integers(0, max_size, n), rng.integers(0, Q, n)
    td = (rng.random(n) * 12).astype(np.float32)
    ref.td_error_update(ids0, ids1, td)
    t.update(th.from_numpy(ids0).to(dev), th.from_numpy(ids1).to(dev), th.from_numpy(td).to(dev), 0.6)
    got = t.sum.view(Q, -1).cpu().numpy()
    np.testing.assert_allclose(got[:, ref.L:], ref.sum[:, ref.L:], rtol=2e-7)
    assert len(set(zip(ids0.tolist(), ids1.tolist()))) < n                    # the list did contain repeats


@pytest.mark.gpu
def test_resumed_per_buffer_does_not_sample_from_empty_trees(tmp_path):
    """save_or_load_history(if_save=False) on a prioritised buffer: the priorities are not in the reference's file set, so the
    loaded transitions re-enter at the maximum priority (empty trees gave total = 0, weights = inf)."""
    from elegantrl_amd.train import Config, ReplayBuffer
    dev = th.device("cuda:0")
    args = Config()
    args.per_alpha, args.per_beta = 0.6, 0.4
    mk = lambda: ReplayBuffer(max_size=40, state_dim=3, action_dim=2, gpu_id=0, num_seqs=2, if_use_per=True, args=args)   # noqa: E731
    a = mk()
    g = th.Generator(device=dev).manual_seed(0)
    a.update((th.randn((25, 2, 3), device=dev, generator=g), th.randn((25, 2, 2), device=dev, generator=g),
              th.randn((25, 2), device=dev, generator=g), th.rand((25, 2), device=dev, generator=g) < 0.9,
              th.rand((25, 2), device=dev, generator=g) < 0.9))
    a.save_or_load_history(str(tmp_path), if_save=True)
    b = mk()
    b.save_or_load_history(str(tmp_path), if_save=False)
    assert b.cur_size == 25
    out = b.sample_for_per(16)
    w, idx = out[6], out[7]
    assert bool(th.isfinite(w).all()) and th.allclose(w, th.ones_like(w))
    assert len(set(th.fmod(idx, b.cur_size).tolist())) > 4                   # draws spread over the rows, not all on row 0


@pytest.mark.gpu
def test_replay_buffer_per_end_to_end():
    from elegantrl_amd.train import Config, ReplayBuffer
    dev = th.device("cuda:0")
    args = Config()
    args.per_alpha, args.per_beta = 0.6, 0.4
    max_size, S, A, Q = 50, 5, 2, 4
    buf = ReplayBuffer(max_size=max_size, state_dim=S, action_dim=A, gpu_id=0, num_seqs=Q, if_use_per=True, args=args)
    g = th.Generator(device=dev).manual_seed(0)
    for add in (20, 20, 20):                                                              # wraps once
        buf.update((th.randn((add, Q, S), device=dev, generator=g), th.randn((add, Q, A), device=dev, generator=g),
                    th.randn((add, Q), device=dev, generator=g), th.rand((add, Q), device=dev, generator=g) < 0.9,
                    th.rand((add, Q), device=dev, generator=g) < 0.9))
    assert buf.if_full and buf.cur_size == max_size
    out = buf.sample_for_per(32)
    assert len(out) == 8
    state, action, reward, undone, unmask, next_state, w, idx = out
    ids0, ids1 = th.fmod(idx, buf.cur_size), th.div(idx, buf.cur_size, rounding_mode="floor")
    assert th.equal(ids0, buf.ids0) and th.equal(ids1, buf.ids1) and int(ids0.max()) <= max_size - 2
    assert th.equal(ids1, th.arange(Q, device=dev).repeat_interleave(8))                   # sequence-major, batch_size // num_seqs each
    assert th.equal(state, buf.states[ids0, ids1]) and th.equal(next_state, buf.states[ids0 + 1, ids1])
    assert th.equal(action, buf.actions[ids0, ids1]) and th.equal(reward, buf.rewards[ids0, ids1])
    assert th.allclose(w, th.ones_like(w))                                                 # all priorities equal (10) so far
    buf.td_error_update_for_per(idx, th.full((32,), 1e-3, device=dev))                     # sampled transitions become unlikely
    again = th.fmod(buf.sample_for_per(32)[7], buf.cur_size)
    leaf = buf.sum_trees.sum.view(Q, -1)[ids1, buf.sum_trees.leaves + ids0]
    np.testing.assert_allclose(leaf.cpu().numpy(), np.float32(1e-3) ** np.float32(0.6), rtol=1e-6)
    w2 = buf.sample_for_per(32)[6]
    assert float(w2.min()) < 1.0 and again.shape == (32,)
