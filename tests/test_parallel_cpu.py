"""World-size-2 `gloo` tests of the data-parallel path (elegantrl_amd/parallel.py) on CPU.

What the N > 1 path adds over the single-GPU path is exactly: identical initial weights by broadcast, one
all-reduce of the 5 advantage sums per iteration, one all-reduce (SUM) of the flat gradient per minibatch with
1/world folded into the optimiser, rank-local shards.  The kernels themselves need a GPU, so here the
per-rank gradients come from the CPU oracle and the test checks the *distributed algebra*: two ranks with half
of the global minibatch each end up with the bit-identical weights that one rank with the whole minibatch gets
(up to fp64 summation order), and stay identical to each other.
"""
import os
import socket

import numpy as np
import pytest
import torch as th
import torch.multiprocessing as mp

from oracle import ppo_numpy as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(seed=0, S=6, A=2, h=32, H=8, N=16):
    rng = np.random.default_rng(seed)
    f = np.float64

    def mlp(out, with_std):
        ws = [rng.standard_normal((h, S)) * 0.3, rng.standard_normal((h, h)) * 0.2, rng.standard_normal((out, h)) * 0.2]
        bs = [rng.standard_normal(h) * 0.1, rng.standard_normal(h) * 0.1, rng.standard_normal(out) * 0.1]
        return O.Mlp([w.astype(f) for w in ws], [b.astype(f) for b in bs], np.zeros(S, f), np.ones(S, f),
                     (rng.standard_normal(out) * 0.1).astype(f) if with_std else None)

    actor, critic = mlp(A, True), mlp(1, False)
    buf = (rng.standard_normal((H, N, S)), rng.standard_normal((H, N, A)), rng.random((H, N)) > 0.1,
           rng.standard_normal((H, N)) - 3.0, rng.standard_normal((H, N)), rng.standard_normal((H, N)))
    return actor, critic, buf


def _flat_grads(buf, ids, actor, critic, inv_batch):
    """flat [actor grads | critic grads] of the SUM objective scaled by inv_batch (what one rank's K6 + slab
    reduce produces: every rank divides by the GLOBAL batch, the all-reduce SUM then yields the global mean)."""
    states, actions, unmasks, logprobs, advantages, rsums = buf
    H = states.shape[0]
    i0, i1 = O.split_ids(ids, H)
    s, a = states[i0, i1], actions[i0, i1]
    um, lp, adv, rs = unmasks[i0, i1], logprobs[i0, i1], advantages[i0, i1], rsums[i0, i1]
    n_local = len(ids)
    _, gw, gb = O.critic_objective(s, rs, um, critic)
    gc = [x for pair in zip(gw, gb) for x in pair]
    _, _, gw, gb, gsl = O.actor_objective(s, a, lp, adv, um, actor, 0.25, 0.001)
    ga = [x for pair in zip(gw, gb) for x in pair] + [gsl]
    # the oracle differentiates the LOCAL mean (1/n_local); rescale to inv_batch
    scale = n_local * inv_batch
    return np.concatenate([g.reshape(-1) * scale for g in ga + gc])


def _apply(actor, critic, flat, st_a, st_c, lr=1e-3, max_norm=3.0):
    pa, pc = actor.trainable(), critic.trainable()
    off = 0
    ga, gc = [], []
    for p in pa:
        ga.append(flat[off:off + p.size].reshape(p.shape)); off += p.size
    for p in pc:
        gc.append(flat[off:off + p.size].reshape(p.shape)); off += p.size
    O.optimizer_backward(pa, ga, st_a, lr, max_norm)
    O.optimizer_backward(pc, gc, st_c, lr, max_norm)


def _worker(rank, world, port, ids_all, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from elegantrl_amd import parallel
    r, w, lr_ = parallel.init_from_env(backend="gloo")
    assert (r, w, lr_) == (rank, world, rank) and parallel.is_distributed()

    # broadcast: ranks start from different weights, rank 0's win
    actor, critic, buf = _problem(seed=0)
    flat_params = th.from_numpy(np.concatenate([p.reshape(-1) for p in actor.trainable() + critic.trainable()]))
    if rank != 0:
        flat_params += 1.0
    parallel.broadcast_(flat_params)
    ref = np.concatenate([p.reshape(-1) for p in actor.trainable() + critic.trainable()])
    np.testing.assert_array_equal(flat_params.numpy(), ref)

    # advantage statistics: raw sums add up across shards
    stats = th.tensor([1.0 + rank, 10.0, 2.0 * (rank + 1), 3.0, 4.0], dtype=th.float64)
    parallel.all_reduce_sum(stats)
    np.testing.assert_array_equal(stats.numpy(), [3.0, 20.0, 6.0, 6.0, 8.0])

    # two minibatches: each rank differentiates its half, all-reduce SUM, identical clip + Adam everywhere
    st_a, st_c = O.AdamState(), O.AdamState()
    B = ids_all.shape[1]
    shard = parallel.shard_range(B, rank, world)
    for ids in ids_all:
        g = th.from_numpy(_flat_grads(buf, ids[shard.start:shard.stop], actor, critic, inv_batch=1.0 / B))
        parallel.all_reduce_sum(g)
        _apply(actor, critic, g.numpy(), st_a, st_c)
    final = np.concatenate([p.reshape(-1) for p in actor.trainable() + critic.trainable()])
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), final)
    assert parallel.all_reduce_max_float(float(rank)) == world - 1
    parallel.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from elegantrl_amd.parallel import shard_range
    for total in (0, 1, 7, 4096):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard_range(total, r, world)]
            assert got == list(range(total))
            sizes = [len(shard_range(total, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_data_parallel_equals_single_rank(tmp_path):
    world, B = 2, 64
    rng = np.random.default_rng(1)
    ids_all = rng.integers(0, 8 * 16, size=(2, B))
    mp.spawn(_worker, args=(world, _free_port(), ids_all, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    np.testing.assert_array_equal(r0, r1)          # ranks stay bit-identical without any weight broadcast

    actor, critic, buf = _problem(seed=0)          # single rank, whole minibatch
    st_a, st_c = O.AdamState(), O.AdamState()
    for ids in ids_all:
        _apply(actor, critic, _flat_grads(buf, ids, actor, critic, inv_batch=1.0 / B), st_a, st_c)
    single = np.concatenate([p.reshape(-1) for p in actor.trainable() + critic.trainable()])
    np.testing.assert_allclose(r0, single, rtol=1e-10, atol=1e-12)


def test_gradient_comm_is_none_without_a_gpu_and_shutdown_is_idempotent(monkeypatch):
    """no HIP device -> the library's RCCL communicator is never attempted (the exchange stays in torch.distributed), and
    parallel.shutdown() is a no-op outside a process group, callable twice."""
    from elegantrl_amd import parallel
    monkeypatch.setattr(parallel, "_grad_comm", None)
    monkeypatch.setattr(parallel, "_grad_comm_tried", False)
    if not th.cuda.is_available():
        assert parallel.gradient_comm() is None
        monkeypatch.setenv("ERL_FORCE_DP", "1")
        monkeypatch.setattr(parallel, "_grad_comm_tried", False)
        assert parallel.gradient_comm() is None
    parallel.shutdown()
    parallel.shutdown()
    assert parallel._grad_comm is None and not parallel._grad_comm_tried


def test_unique_id_comes_from_rccl_without_a_gpu():
    """erl_comm_unique_id only needs librccl (bound with dlopen): 128 opaque bytes, different on every call."""
    import ctypes
    from elegantrl_amd import _hip
    L = _hip.lib()
    a, b = (ctypes.c_uint8 * _hip.COMM_ID_BYTES)(), (ctypes.c_uint8 * _hip.COMM_ID_BYTES)()
    rc = L.erl_comm_unique_id(a)
    if rc != 0:
        pytest.skip("librccl not loadable here: " + L.erl_last_error_string().decode())
    assert L.erl_comm_unique_id(b) == 0
    assert bytes(a) != bytes(b) and any(bytes(a))
    assert L.erl_comm_world_size(None) == 1
    assert L.erl_comm_allreduce_sum_f32(None, None, 4, None) != 0 and b"bad argument" in L.erl_last_error_string()
