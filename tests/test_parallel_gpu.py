"""Data-parallel AgentPPO end to end on the real kernels: two ranks share the one GPU of the test box (RCCL refuses
two ranks per device, so the collective backend here is gloo on CUDA tensors; the code path in AgentPPO /
parallel.py is the one the 8-GPU run takes with backend "nccl").  Checks: identical initial weights after the
broadcast, the job-wide advantage statistics, one flat-gradient all-reduce per minibatch, and that both ranks hold
bit-identical weights after two iterations although they rolled out different env shards."""
import os
import socket

import numpy as np
import pytest
import torch as th
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, backend="gloo", one_gpu_per_rank=False, collective="torch", shape=None):
    gpu = rank if one_gpu_per_rank else 0
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(gpu), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      ERL_DP_COLLECTIVE=collective)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from elegantrl_amd import parallel
    from elegantrl_amd.agents import AgentPPO
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config
    parallel.init_from_env(backend=backend)
    th.cuda.set_device(gpu)
    N, S, A, H, B = 256, 64, 8, 16, 1024
    net_dims = None
    if shape is not None:                                       # (N, S, A, H, B, net_dims): small rows for many ranks on ONE GPU
        N, S, A, H, B, net_dims = shape
    args = Config(AgentPPO, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 50, "state_dim": S, "action_dim": A,
                                        "if_discrete": False})
    if net_dims is not None:
        args.net_dims = list(net_dims)
    args.horizon_len, args.batch_size, args.repeat_times = H, B, 3 * B / H
    args.learning_rate, args.random_seed = 1e-3, 5
    args.world_size, args.rank, args.gpu_id = world, rank, gpu
    th.manual_seed(100 + rank)                                  # ranks start from DIFFERENT weights on purpose
    agent = AgentPPO(args.net_dims, S, A, gpu_id=gpu, args=args)
    parallel.broadcast_(agent._flat)
    w0 = agent._flat.clone()
    env = SynVecEnv(N, S, A, max_step=50, gpu_id=gpu, seed=7919 * rank)
    agent.last_state = env.reset()[0]
    logs = []
    for _ in range(2):
        items = agent.explore_env(env, H)
        logs.append(agent.update_net(list(items)))
    np.save(os.path.join(out_dir, f"w0_{rank}.npy"), w0.cpu().numpy())
    np.save(os.path.join(out_dir, f"w_{rank}.npy"), agent._flat.cpu().numpy())
    np.save(os.path.join(out_dir, f"stats_{rank}.npy"), agent._stats.cpu().numpy())
    np.save(os.path.join(out_dir, f"rew_{rank}.npy"), items[3].cpu().numpy())
    np.save(os.path.join(out_dir, f"logs_{rank}.npy"), np.array(logs))
    comm = parallel.gradient_comm()
    np.save(os.path.join(out_dir, f"comm_{rank}.npy"), np.array([comm is not None, comm.world if comm is not None else 0,
                                                                 {"rccl": 1, "p2p": 2}[comm.kind] if comm is not None else 0]))
    import json
    with open(os.path.join(out_dir, f"route_{rank}.json"), "w") as f:
        json.dump(parallel.route_report(), f)
    parallel.barrier()
    parallel.shutdown()


def _p2p_worker(rank, world, port, out_dir):
    """the one-shot peer-to-peer exchange on its own: two ranks, ONE GPU, the peers' stages mapped through HIP IPC"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from elegantrl_amd import _hip, ops, parallel
    parallel.init_from_env(backend="gloo")
    th.cuda.set_device(0)
    comm = parallel.P2PComm.create(max_count=70000)
    result = {"created": comm is not None}
    if comm is not None:
        g = th.Generator(device="cuda").manual_seed(1234 + rank)
        outs, refs = [], []
        for it, n in enumerate([50848, 1, 70000, 257, 50848, 50848, 4096, 50848]):     # both stage halves, several sizes, reuse
            x = th.randn(n, device="cuda", generator=g) * (1 + it)
            ref = x.cpu().clone()
            parallel.dist.all_reduce(ref)                       # gloo on the host copy: the expected SUM (order-free for 2 ranks)
            comm.all_reduce_sum(x)
            d = th.randn(5 + it, device="cuda", generator=g, dtype=th.float64)          # the advantage sums' route (fp64)
            dref = d.cpu().clone()
            parallel.dist.all_reduce(dref)
            comm.all_reduce_sum(d)
            th.cuda.synchronize()
            assert th.equal(d.cpu(), dref), "fp64 peer-to-peer sum differs from gloo's"
            outs.append(x.cpu().numpy())
            refs.append(ref.numpy())
        # the fused optimiser tail: [slab reduction + exchange + partial norms] -> [clip + Adam], against the same tail with
        # gloo in the middle (reduce -> gloo all-reduce -> partial norms -> clip + Adam)
        n_slabs, stride, groups = 37, 50848, [(0, 25872), (25872, 24961)]
        slabs = th.randn((n_slabs, stride), device="cuda", generator=g)
        slabs[:, 50833 + 4:] = 0
        w = []
        for route in ("p2p", "gloo"):
            gsum = th.empty(stride, device="cuda")
            params = th.linspace(-1, 1, 50833, device="cuda")
            m1, m2 = th.zeros_like(params), th.zeros_like(params)
            for step in (1, 2):
                if route == "p2p":
                    ops.grad_reduce_partials(slabs, n_slabs, stride, gsum, groups, grad_scale=0.5, comm=comm)
                else:
                    ops.grad_reduce(slabs, n_slabs, stride, gsum)
                    h = gsum.cpu()
                    parallel.dist.all_reduce(h)
                    gsum.copy_(h)
                    ops.grad_sq_partials(gsum, stride, groups, grad_scale=0.5)
                ops.clip_adam_partials(params, gsum, m1, m2, stride, groups, step, 1e-3, 0.5, grad_scale=0.5)
            th.cuda.synchronize()
            w.append(params.cpu().numpy())
            np.save(os.path.join(out_dir, f"tail_{route}_{rank}.npy"), w[-1])
            np.save(os.path.join(out_dir, f"tailg_{route}_{rank}.npy"), gsum.cpu().numpy())
        _hip.check_async_faults()
        for k, (o, r) in enumerate(zip(outs, refs)):
            np.save(os.path.join(out_dir, f"p2p_out{k}_{rank}.npy"), o)
            np.save(os.path.join(out_dir, f"p2p_ref{k}_{rank}.npy"), r)
        result["n"] = len(outs)
        parallel.barrier()
        comm.close()
    np.save(os.path.join(out_dir, f"p2p_{rank}.npy"), np.array([int(result["created"]), result.get("n", 0)]))
    parallel.barrier()
    parallel.dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_p2p_allreduce_two_ranks_on_one_gpu(tmp_path):
    """csrc/p2p.hip + csrc/grad_tail.hip (SURVEY 8e's one-shot exchange): handle exchange, per-workgroup flag protocol, stage
    reuse over 16 back-to-back exchanges of different sizes and dtypes; sums equal gloo's and are bit-identical on both
    ranks; the FUSED tail (exchange inside the slab reduction) leaves the weights of the same tail with gloo in the middle.
    Skips when the box cannot share device memory between two processes (IPC)."""
    world = 2
    mp.spawn(_p2p_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    meta = [np.load(tmp_path / f"p2p_{r}.npy") for r in range(world)]
    if not all(m[0] for m in meta):
        pytest.skip("HIP IPC between two processes on this device is unavailable")
    for k in range(int(meta[0][1])):
        o = [np.load(tmp_path / f"p2p_out{k}_{r}.npy") for r in range(world)]
        ref = np.load(tmp_path / f"p2p_ref{k}_0.npy")
        np.testing.assert_array_equal(o[0], o[1])
        np.testing.assert_array_equal(o[0], ref)          # two addends: every summation order gives the same fp32 result
    t = {(route, r): np.load(tmp_path / f"tail_{route}_{r}.npy") for route in ("p2p", "gloo") for r in range(world)}
    tg = {(route, r): np.load(tmp_path / f"tailg_{route}_{r}.npy") for route in ("p2p", "gloo") for r in range(world)}
    np.testing.assert_array_equal(tg[("p2p", 0)], tg[("p2p", 1)])
    np.testing.assert_array_equal(tg[("p2p", 0)], tg[("gloo", 0)])
    np.testing.assert_array_equal(t[("p2p", 0)], t[("p2p", 1)])
    np.testing.assert_array_equal(t[("p2p", 0)], t[("gloo", 0)])
    assert np.isfinite(t[("p2p", 0)]).all() and not np.array_equal(t[("p2p", 0)], np.linspace(-1, 1, 50833, dtype=np.float32))


def _load(tmp_path, sub, n, world=2):
    return [np.load(tmp_path / sub / f"{n}_{r}.npy") for r in range(world)]


@pytest.mark.timeout(900)
def test_two_rank_agent_in_lockstep_on_every_route(tmp_path):
    """Two ranks share the one GPU (gloo process group) and run two data-parallel PPO iterations three times: through
    torch.distributed (`torch`), through the peer-to-peer communicator forced (`p2p`), and with the route chosen by the
    start-up self-test (`auto`: RCCL cannot span two ranks on one device, so the self-tested peer-to-peer route must win).
    Every run: identical weights after the broadcast, job-wide advantage sums, ranks bit-identical after training.  Across
    runs: the fused tail (exchange inside the slab-reduction launch) leaves exactly the weights of the gloo route."""
    import json
    world = 2
    for mode in ("torch", "p2p", "auto"):
        (tmp_path / mode).mkdir()
        mp.spawn(_worker, args=(world, _free_port(), str(tmp_path / mode), "gloo", False, mode), nprocs=world, join=True)
    w0, w, stats, rew, logs = (_load(tmp_path, "torch", n) for n in ("w0", "w", "stats", "rew", "logs"))
    np.testing.assert_array_equal(w0[0], w0[1])                       # broadcast
    assert not np.array_equal(rew[0], rew[1])                         # different env shards were rolled out
    np.testing.assert_array_equal(stats[0], stats[1])                 # job-wide advantage sums
    assert stats[0][1] == 2 * 16 * 256                                # counts add up over both shards
    assert not np.array_equal(w[0], w0[0])                            # training moved the weights ...
    np.testing.assert_array_equal(w[0], w[1])                         # ... identically on both ranks
    np.testing.assert_allclose(logs[0], logs[1], rtol=1e-6)           # logged objectives are global means
    assert np.isfinite(w[0]).all() and np.isfinite(logs[0]).all()
    assert all(c[0] == 0 for c in _load(tmp_path, "torch", "comm"))
    p2p_comm = _load(tmp_path, "p2p", "comm")
    if not all(c[0] for c in p2p_comm):
        pytest.skip("HIP IPC between two processes on this device is unavailable")
    for mode in ("p2p", "auto"):
        comm = _load(tmp_path, mode, "comm")
        assert all(c[0] == 1 and c[1] == 2 and c[2] == 2 for c in comm), f"{mode}: the peer-to-peer route was not selected"
        wm = _load(tmp_path, mode, "w")
        np.testing.assert_array_equal(wm[0], wm[1])
        np.testing.assert_array_equal(wm[0], w[0])                    # == the torch.distributed / gloo route, bit for bit
        np.testing.assert_array_equal(_load(tmp_path, mode, "stats")[0], stats[0])
        np.testing.assert_array_equal(_load(tmp_path, mode, "logs")[0], logs[0])
    rep = json.load(open(tmp_path / "auto" / "route_0.json"))
    assert rep["p2p_probe"] == "ok", rep                               # first contact happened in a throw-away child process per rank
    assert rep["p2p_selftest"] == "ok" and rep["p2p_us"] > 0 and "peer-to-peer" in rep["selected"], rep


@pytest.mark.timeout(1200)
@pytest.mark.skipif(th.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_rank_one_gpu_per_rank_every_route(tmp_path):
    """The multi-GPU path as the 8-GPU run takes it: backend "nccl" (= RCCL over xGMI), one GPU per rank.  `rccl`: the
    library's own communicator (erl_comm_init with world > 1) issuing the gradient all-reduce from inside
    erl_ppo_update_dp_f32; `p2p`: the exchange across xGMI inside the slab-reduction launch; `auto`: both self-tested, the
    faster kept.  Ranks end bit-identical to each other AND to the gloo route on the same two GPUs (a two-term sum is
    order-free, so any all-reduce algorithm yields the same bits)."""
    import json
    world = 2
    for mode, backend in (("rccl", "nccl"), ("p2p", "nccl"), ("auto", "nccl"), ("torch", "gloo")):
        (tmp_path / mode).mkdir()
        mp.spawn(_worker, args=(world, _free_port(), str(tmp_path / mode), backend, True, mode), nprocs=world, join=True)
    wg = _load(tmp_path, "torch", "w")
    comm = _load(tmp_path, "rccl", "comm")
    assert all(c[0] == 1 and c[1] == world and c[2] == 1 for c in comm), "the library RCCL communicator did not come up with world = 2"
    assert all(c[0] == 0 for c in _load(tmp_path, "torch", "comm"))
    rep = json.load(open(tmp_path / "auto" / "route_0.json"))
    assert rep["rccl_selftest"] == "ok", rep
    for mode in ("rccl", "p2p", "auto"):
        if mode == "p2p" and not all(c[0] for c in _load(tmp_path, "p2p", "comm")):
            continue                                                  # no peer mapping on this box: auto must have fallen back
        w = _load(tmp_path, mode, "w")
        np.testing.assert_array_equal(w[0], w[1])                     # ranks in lockstep
        np.testing.assert_array_equal(w[0], wg[0])                    # == torch.distributed/gloo route
        np.testing.assert_array_equal(_load(tmp_path, mode, "stats")[0], _load(tmp_path, "torch", "stats")[0])
        assert not np.array_equal(w[0], _load(tmp_path, mode, "w0")[0]) and np.isfinite(w[0]).all()
    assert all(c[0] == 1 for c in _load(tmp_path, "auto", "comm")), rep


# ---- the exchange at the node's real world sizes (4, 8 ranks), on the one GPU of the test box -----------------------
# Ranks share the device, so their launches share its 256 CUs: rows are kept short (<= 8192 floats = 32 workgroups per rank)
# so that every rank's exchange workgroups are resident together -- a workgroup spins for its peers' SAME slice, which only
# arrives if those launches get CUs (on a node every rank has its own GPU).
def _rank_ordered_sum(t_host):
    """what the exchange kernel computes: x[0] + x[1] + ... in RANK order, sequential adds in the tensor's dtype (gloo's
    all-reduce associates differently for world > 2, so it is not the bit-exact reference any more)"""
    from elegantrl_amd import parallel
    world = parallel.dist.get_world_size()
    parts = [th.empty_like(t_host) for _ in range(world)]
    parallel.dist.all_gather(parts, t_host)
    ref = parts[0].clone()
    for r in range(1, world):
        ref += parts[r]
    return ref


def _pn_worker(rank, world, port, out_dir, withhold):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from elegantrl_amd import _hip, ops, parallel
    parallel.init_from_env(backend="gloo")
    th.cuda.set_device(0)
    comm = parallel.P2PComm.create(max_count=8192)
    res = {"created": comm is not None, "exact": True, "tail": True, "fault": None}
    if comm is not None:
        g = th.Generator(device="cuda").manual_seed(99 + rank)
        for it, n in enumerate([4096, 1, 8192, 257, 5000, 5000, 8192, 300]):          # both stage halves, reuse, odd sizes
            x = th.randn(n, device="cuda", generator=g) * (1 + it)
            d = th.randn(3 + it, device="cuda", generator=g, dtype=th.float64)
            ref, dref = _rank_ordered_sum(x.cpu()), _rank_ordered_sum(d.cpu())
            comm.all_reduce_sum(x)
            comm.all_reduce_sum(d)
            th.cuda.synchronize()
            res["exact"] = res["exact"] and bool(th.equal(x.cpu(), ref)) and bool(th.equal(d.cpu(), dref))
        # the fused tail (exchange inside the slab reduction) against reduce -> rank-ordered host sum -> partial norms -> Adam
        n_slabs, stride, groups = 19, 8192, [(0, 4100), (4100, 4060)]
        slabs = th.randn((n_slabs, stride), device="cuda", generator=g)
        slabs[:, 8160 + 4:] = 0
        w = {}
        for route in ("p2p", "host"):
            gsum = th.empty(stride, device="cuda")
            params = th.linspace(-1, 1, 8160, device="cuda")
            m1, m2 = th.zeros_like(params), th.zeros_like(params)
            for step in (1, 2):
                if route == "p2p":
                    ops.grad_reduce_partials(slabs, n_slabs, stride, gsum, groups, grad_scale=1.0 / world, comm=comm)
                else:
                    ops.grad_reduce(slabs, n_slabs, stride, gsum)
                    gsum.copy_(_rank_ordered_sum(gsum.cpu()))
                    ops.grad_sq_partials(gsum, stride, groups, grad_scale=1.0 / world)
                ops.clip_adam_partials(params, gsum, m1, m2, stride, groups, step, 1e-3, 0.5, grad_scale=1.0 / world,
                                       comm=comm if route == "p2p" else None)
            th.cuda.synchronize()
            w[route] = params.cpu()
        res["tail"] = bool(th.equal(w["p2p"], w["host"])) and bool(th.isfinite(w["p2p"]).all())
        np.save(os.path.join(out_dir, f"tailw_{rank}.npy"), w["p2p"].numpy())
        _hip.check_async_faults()
        if withhold:
            # ---- fault path: the LAST rank never launches this exchange.  Every other rank's wait is bounded, reports through
            # erl_async_fault_count (source "peer-to-peer ... exchange") and POISONS the communicator: clip + Adam of this update
            # loop touch nothing.  Reading the fault clears the poison.
            parallel.barrier()
            params = th.linspace(-1, 1, 8160, device="cuda")
            m1, m2 = th.zeros_like(params), th.zeros_like(params)
            before = params.clone()
            if rank != world - 1:
                comm.set_spin(1 << 13)
                gsum = th.empty(stride, device="cuda")
                ops.grad_reduce_partials(slabs, n_slabs, stride, gsum, groups, grad_scale=1.0 / world, comm=comm)
                ops.clip_adam_partials(params, gsum, m1, m2, stride, groups, 1, 1e-3, 0.5, grad_scale=1.0 / world, comm=comm)
                ops.clip_adam_partials(params, gsum, m1, m2, stride, groups, 2, 1e-3, 0.5, grad_scale=1.0 / world, comm=comm)   # sticky
                th.cuda.synchronize()
                parallel.barrier()                                                 # (the withholding rank launches now: see below)
                untouched = bool(th.equal(params, before)) and not bool(m1.any()) and not bool(m2.any())
                n_faults = _hip.lib().erl_async_fault_count(1)                     # report + reset: clears the poison
                msg = _hip.lib().erl_last_error_string().decode()
                # un-poisoned: the same launch (partial norms from a local reduction) now updates
                ops.grad_reduce_partials(slabs, n_slabs, stride, gsum, groups, grad_scale=1.0 / world)
                ops.clip_adam_partials(params, gsum, m1, m2, stride, groups, 1, 1e-3, 0.5, grad_scale=1.0 / world, comm=comm)
                th.cuda.synchronize()
                res["fault"] = [untouched, int(n_faults), "peer-to-peer" in msg and "SKIPPED" in msg, not bool(th.equal(params, before))]
            else:
                # round 5: the ranks that gave up raised their word in EVERY rank's table.  The late rank's launch of the same exchange finds
                # its peers' slices (they published before they waited) -- and their poison: its own optimiser steps are skipped too and its
                # host raises as well, instead of one replica stepping on alone
                parallel.barrier()
                gsum = th.empty(stride, device="cuda")
                ops.grad_reduce_partials(slabs, n_slabs, stride, gsum, groups, grad_scale=1.0 / world, comm=comm)
                ops.clip_adam_partials(params, gsum, m1, m2, stride, groups, 1, 1e-3, 0.5, grad_scale=1.0 / world, comm=comm)
                th.cuda.synchronize()
                res["fault"] = "withheld"
                res["late_rank"] = [bool(th.equal(params, before)), int(_hip.lib().erl_async_fault_count(1))]
        parallel.barrier()
        comm.close()
    import json
    with open(os.path.join(out_dir, f"pn_{rank}.json"), "w") as f:
        json.dump(res, f)
    parallel.barrier()
    parallel.dist.destroy_process_group()


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("world", [4, 8])
def test_p2p_exchange_at_node_world_sizes_on_one_gpu(tmp_path, world):
    """`world` ranks share the GPU through HIP IPC (ERL_P2P_MAX_WORLD = 8: the 8-stage rank-ordered sum and the
    [sender][workgroup] flag table of csrc/grad_tail.hip at their real sizes): stand-alone fp32 / fp64 all-reduces equal the
    rank-ordered sum BIT FOR BIT on every rank, the fused tail leaves the weights of the tail with a host-side sum in the
    middle on every rank (identical across ranks), and -- world 4 -- a rank that withholds its launch makes every other
    rank's bounded wait give up, skip its optimiser steps and name the exchange in erl_async_fault_count."""
    import json
    mp.spawn(_pn_worker, args=(world, _free_port(), str(tmp_path), world == 4), nprocs=world, join=True)
    res = [json.load(open(tmp_path / f"pn_{r}.json")) for r in range(world)]
    if not all(r["created"] for r in res):
        pytest.skip("HIP IPC between processes on this device is unavailable")
    assert all(r["exact"] for r in res), "an all-reduce differs from the rank-ordered sum"
    assert all(r["tail"] for r in res), "fused tail != tail with the host-side rank-ordered sum"
    w = [np.load(tmp_path / f"tailw_{r}.npy") for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(w[0], w[r])
    if world == 4:
        assert res[world - 1]["fault"] == "withheld"
        late_untouched, late_faults = res[world - 1]["late_rank"]
        assert late_untouched and late_faults > 0, "the peers' poison did not reach the rank that launched late"
        for r in range(world - 1):
            untouched, n_faults, named, recovered = res[r]["fault"]
            assert untouched, f"rank {r}: a timed-out exchange was applied to the parameters"
            assert n_faults > 0 and named, f"rank {r}: the timeout was not reported as a peer-to-peer exchange fault"
            assert recovered, f"rank {r}: the communicator stayed poisoned after the fault was read"


_SMALL = (64, 8, 2, 8, 256, (64, 64))      # N, S, A, H, B, net: gradient row of 9.7k floats = 38 exchange workgroups per rank


@pytest.mark.timeout(2400)
@pytest.mark.parametrize("world", [4, 8])
def test_agent_in_lockstep_at_node_world_sizes_on_one_gpu(tmp_path, world):
    """One data-parallel PPO job of 4 / 8 ranks on the one GPU (gloo group; small nets so that all ranks' exchange workgroups
    are co-resident), two iterations per route: `torch` (torch.distributed), `p2p` (forced), `auto` (probe + self-test).
    Ranks end BIT-IDENTICAL on every route; `p2p` == `auto` bit for bit; against `torch` only to rounding (gloo's all-reduce
    associates the `world` addends differently from the kernel's rank order)."""
    import json
    for mode in ("torch", "p2p", "auto"):
        (tmp_path / mode).mkdir()
        mp.spawn(_worker, args=(world, _free_port(), str(tmp_path / mode), "gloo", False, mode, _SMALL), nprocs=world, join=True)
    N, S, A, H, B, _ = _SMALL
    wt, w0 = _load(tmp_path, "torch", "w", world), _load(tmp_path, "torch", "w0", world)
    stats = _load(tmp_path, "torch", "stats", world)
    for r in range(1, world):
        np.testing.assert_array_equal(w0[0], w0[r])
        np.testing.assert_array_equal(wt[0], wt[r])
        np.testing.assert_array_equal(stats[0], stats[r])
    assert stats[0][1] == world * H * N and not np.array_equal(wt[0], w0[0]) and np.isfinite(wt[0]).all()
    if not all(c[0] for c in _load(tmp_path, "p2p", "comm", world)):
        pytest.skip("HIP IPC between processes on this device is unavailable")
    wp = _load(tmp_path, "p2p", "w", world)
    for mode in ("p2p", "auto"):
        comm = _load(tmp_path, mode, "comm", world)
        assert all(c[0] == 1 and c[1] == world and c[2] == 2 for c in comm), f"{mode}: the peer-to-peer route was not selected"
        wm = _load(tmp_path, mode, "w", world)
        for r in range(1, world):
            np.testing.assert_array_equal(wm[0], wm[r])
        np.testing.assert_array_equal(wm[0], wp[0])
        np.testing.assert_allclose(wm[0], wt[0], rtol=0, atol=2e-5)
        np.testing.assert_allclose(_load(tmp_path, mode, "stats", world)[0], stats[0], rtol=1e-6)    # second rollout: weights differ by rounding
    rep = json.load(open(tmp_path / "auto" / "route_0.json"))
    assert rep["p2p_probe"] == "ok" and rep["p2p_selftest"] == "ok", rep


# ---- the library-owned RCCL communicator (erl_comm_*) on one rank --------------------------------------------------
_DP_SCRIPT = r"""
import os, sys
import numpy as np
import torch as th
from elegantrl_amd import parallel
from elegantrl_amd.agents import AgentPPO
from elegantrl_amd.envs import SynVecEnv
from elegantrl_amd.train import Config
rank, world, _ = parallel.init_from_env()
N, S, A, H, B = 256, 64, 8, 16, 1024
args = Config(AgentPPO, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 50, "state_dim": S, "action_dim": A,
                                    "if_discrete": False})
args.horizon_len, args.batch_size, args.repeat_times = H, B, 3 * B / H
args.learning_rate, args.random_seed = 1e-3, 5
args.world_size, args.rank, args.gpu_id = world, rank, 0
th.manual_seed(3)
args.net_dims = [int(x) for x in os.environ["ERL_TEST_NET"].split(",")]
agent = AgentPPO(args.net_dims, S, A, gpu_id=0, args=args)
env = SynVecEnv(N, S, A, max_step=50, gpu_id=0, seed=11)
agent.last_state = env.reset()[0]
logs = [agent.update_net(list(agent.explore_env(env, H))) for _ in range(2)]
comm = parallel.gradient_comm() if parallel.force_dp() else None
if comm is not None:                      # the collective on its own: SUM over one rank is the identity, on torch's stream
    x = th.randn(50837, device="cuda")
    y = comm.all_reduce_sum(x.clone())
    assert th.equal(x, y)
    d = th.randn(8, device="cuda", dtype=th.float64)
    assert th.equal(d, comm.all_reduce_sum(d.clone()))
np.savez(sys.argv[1], w=agent._flat.cpu().numpy(), logs=np.array(logs), used_comm=comm is not None,
         pg=parallel.dist.is_initialized())
"""


def _run_dp_script(tmp_path, name, **env):
    import subprocess
    import sys
    out = tmp_path / f"{name}.npz"
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", **env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _DP_SCRIPT, str(out)], env=e, cwd=root, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return np.load(out)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("net", ["128,128", "96,64,32"], ids=["fused", "generic"])
def test_one_rank_comm_matches_single_process(tmp_path, net):
    """ERL_FORCE_DP=1 drives the data-parallel branch of update_net with one rank: (a) through the library's RCCL
    communicator inside erl_ppo_update_dp_f32 (unique id carried by the nccl process group), (b) through the peer-to-peer
    communicator (the exchange protocol inside the slab-reduction launch, with itself as the only peer), (c) with the
    self-tested choice, (d) through torch.distributed's all_reduce.  All must leave exactly the weights of the plain
    single-process loop."""
    plain = _run_dp_script(tmp_path, "plain", ERL_TEST_NET=net)
    assert not plain["used_comm"] and not plain["pg"]
    for mode in ("rccl", "p2p", "auto", "torch"):
        run = _run_dp_script(tmp_path, mode, ERL_TEST_NET=net, ERL_FORCE_DP="1", ERL_DP_COLLECTIVE=mode)
        assert run["pg"]
        assert bool(run["used_comm"]) == (mode != "torch"), f"{mode}: library communicator did not come up on the GPU box"
        np.testing.assert_array_equal(plain["w"], run["w"])
        np.testing.assert_array_equal(plain["logs"], run["logs"])


_WIDE = (128, 24, 4, 8, 256, (256, 128))      # N, S, A, H, B, net_dims: the reference demo's (256, 128) network (csrc/ppo_step_wd_impl.h)


def test_two_rank_agent_with_the_wide_minibatch_kernel(tmp_path):
    """two data-parallel ranks on one GPU with net_dims (256, 128): the fused wide minibatch kernel under the `torch` route (gradient rows
    of 80k floats all-reduced by torch.distributed); ranks end bit-identical.  (The in-launch peer-to-peer exchange cannot be exercised
    with this kernel on ONE shared GPU: a rank's spinning exchange workgroups occupy SIMDs, and the other rank's minibatch kernel needs
    whole CUs -- 512 registers per wave, 154 KB of LDS -- to start; with one GPU per rank the two never compete.)"""
    (tmp_path / "torch").mkdir()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path / "torch"), "gloo", False, "torch", _WIDE), nprocs=2, join=True)
    wt, w0 = _load(tmp_path, "torch", "w", 2), _load(tmp_path, "torch", "w0", 2)
    np.testing.assert_array_equal(w0[0], w0[1])
    np.testing.assert_array_equal(wt[0], wt[1])
    assert not np.array_equal(wt[0], w0[0]) and np.isfinite(wt[0]).all()
    stats = _load(tmp_path, "torch", "stats", 2)
    np.testing.assert_array_equal(stats[0], stats[1])
