"""Data-parallel plumbing: one process per GPU, env shards one-per-rank, gradients (and the advantage
statistics) summed with a single flat all-reduce per minibatch over torch.distributed -- backend "nccl" is
RCCL over xGMI on ROCm; "gloo" on CPU for the tests.

The reference has no collective on this path (its multi-GPU mode all-gathers rollout *data* through host
pipes, elegantrl/train/run.py:305-320); this is the replacement design of SURVEY.md section 8e:
203 KB of fp32 gradients per minibatch, latency-bound, one collective, 1/world scaling folded into K7.
"""
from __future__ import annotations

import os
from typing import Optional

import torch as th
import torch.distributed as dist


def force_dp() -> bool:
    """ERL_FORCE_DP=1: run the data-parallel code path even with one rank (measures its overhead on a 1-GPU box)."""
    return os.environ.get("ERL_FORCE_DP") == "1"


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_dp())


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun).
    Returns (rank, world_size, local_rank).  No-op for world_size 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or force_dp()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # "nccl" IS RCCL on ROCm; ERL_DIST_BACKEND=gloo lets several ranks share one GPU (tests)
            backend = os.environ.get("ERL_DIST_BACKEND") or ("nccl" if th.cuda.is_available() else "gloo")
        if th.cuda.is_available():
            local_rank = local_rank % th.cuda.device_count()
            th.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


class RcclComm:
    """RCCL communicator owned by liberl_hip.so (erl_comm_*): the gradient all-reduce is enqueued on the SAME stream as
    the kernels either side of it, from inside the C loop.  torch.distributed stays the control plane: it carries the
    128-byte unique id to the ranks and the all-ranks agreement that the communicator came up everywhere."""

    kind = "rccl"

    def __init__(self, handle: int, rank: int, world: int):
        self.handle, self.rank, self.world = handle, rank, world

    @classmethod
    def create(cls) -> Optional["RcclComm"]:
        """Collective over the default process group (or a 1-rank communicator when there is none).  Returns None --
        on EVERY rank -- unless every rank got its communicator, so the ranks can never disagree about the path."""
        import ctypes
        from . import _hip
        L = _hip.lib()
        distributed = is_distributed()
        rank = dist.get_rank() if distributed else 0
        world = dist.get_world_size() if distributed else 1
        ident = (ctypes.c_uint8 * _hip.COMM_ID_BYTES)()
        ok = 1
        if rank == 0 and L.erl_comm_unique_id(ident) != 0:
            ok = 0
        if distributed:
            box = [bytes(ident) if ok else None]
            dist.broadcast_object_list(box, src=0)
            if box[0] is None:
                return None
            ident = (ctypes.c_uint8 * _hip.COMM_ID_BYTES).from_buffer_copy(box[0])
        elif not ok:
            return None
        out = ctypes.c_void_p(None)
        ok = int(L.erl_comm_init(ident, rank, world, ctypes.byref(out)) == 0 and bool(out.value))
        if distributed:
            flag = th.tensor([ok], dtype=th.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            all_ok = int(flag.item())
        else:
            all_ok = ok
        if not all_ok:
            if ok:
                L.erl_comm_destroy(out)
            return None
        return cls(out.value, rank, world)

    def all_reduce_sum(self, t: th.Tensor) -> th.Tensor:
        """in-place SUM over ranks of a contiguous fp32 or fp64 device tensor, on torch's current stream"""
        from . import _hip
        L = _hip.lib()
        if t.dtype == th.float64:
            _hip.check(L.erl_comm_allreduce_sum_f64(self.handle, _hip.ptr(t, th.float64), t.numel(), _hip.stream_ptr()),
                       "erl_comm_allreduce_sum_f64")
        else:
            _hip.check(L.erl_comm_allreduce_sum_f32(self.handle, _hip.ptr(t, th.float32), t.numel(), _hip.stream_ptr()),
                       "erl_comm_allreduce_sum_f32")
        return t

    def close(self):
        if self.handle:
            from . import _hip
            th.cuda.synchronize()
            _hip.lib().erl_comm_destroy(self.handle)
            self.handle = None


class P2PComm(RcclComm):
    """One-shot peer-to-peer exchange (csrc/p2p.hip + csrc/grad_tail.hip): the same library handle type as the RCCL
    communicator, so `all_reduce_sum` and the C update loop take it unchanged -- and the update loop's exchange then happens
    INSIDE its slab-reduction launch.  torch.distributed carries every rank's 64-byte IPC handle to every rank and the
    all-ranks agreement that the peers' stages are mapped everywhere."""

    kind = "p2p"
    MAX_COUNT = 1 << 18            # floats per exchange the stage rows are sized for (1 MiB per row, 2 x world rows per rank)

    def set_spin(self, spins: int) -> None:
        """polls before a wait for a peer gives up (0 = default); the self-test lowers it so a dead route costs seconds"""
        from . import _hip
        _hip.check(_hip.lib().erl_comm_p2p_set_spin(self.handle, int(spins)), "erl_comm_p2p_set_spin")

    @classmethod
    def create(cls, max_count: Optional[int] = None) -> Optional["P2PComm"]:
        import ctypes
        from . import _hip
        L = _hip.lib()
        distributed = is_distributed()
        rank = dist.get_rank() if distributed else 0
        world = dist.get_world_size() if distributed else 1
        if world > 8:
            return None
        out = ctypes.c_void_p(None)
        handle = (ctypes.c_uint8 * _hip.P2P_HANDLE_BYTES)()
        ok = int(L.erl_comm_p2p_create(rank, world, int(max_count or cls.MAX_COUNT), ctypes.byref(out), handle) == 0 and bool(out.value))
        handles = [bytes(handle) if ok else None]
        if distributed:
            handles = [None] * world
            dist.all_gather_object(handles, bytes(handle) if ok else None)
        if any(h is None for h in handles):
            if ok:
                L.erl_comm_destroy(out)
            return None
        blob = (ctypes.c_uint8 * (_hip.P2P_HANDLE_BYTES * world)).from_buffer_copy(b"".join(handles))
        ok = int(L.erl_comm_p2p_connect(out, blob) == 0)
        if distributed:
            flag = th.tensor([ok], dtype=th.int32)
            if dist.get_backend() == "nccl":
                flag = flag.cuda()
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if not ok:
            L.erl_comm_destroy(out)
            return None
        return cls(out.value, rank, world)


_grad_comm: Optional[RcclComm] = None
_grad_comm_tried = False
_route_report: dict = {}


def route_report() -> dict:
    """what `gradient_comm` measured and decided (bench.py prints it): per-route self-test verdict and microseconds per
    all-reduce of the gradient row, the selected route, the ranks RCCL saw."""
    return dict(_route_report)


def _agree(ok: bool) -> bool:
    """all-ranks AND over the process group (identity without one)"""
    if not (dist.is_available() and dist.is_initialized()):
        return bool(ok)
    flag = th.tensor([int(bool(ok))], dtype=th.int32)
    if dist.get_backend() == "nccl":
        flag = flag.cuda()
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


def _reference_sum(x: th.Tensor) -> th.Tensor:
    """SUM over ranks through torch.distributed (the route that needs no validation); gloo groups reduce a host copy"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x.clone()
    if dist.get_backend() == "nccl":
        y = x.clone()
        dist.all_reduce(y)
        return y
    y = x.cpu()
    dist.all_reduce(y)
    return y.to(x.device)


def selftest(comm: RcclComm, count: int, rounds: int = 8, timed_calls: int = 100) -> dict:
    """Validate a library communicator on THIS machine before the update loop trusts it: `rounds` all-reduces of `count`
    floats (rank-distinct data, both stage halves of the peer-to-peer route, plus an 8-double one) must agree with
    torch.distributed's sums on every rank and leave no device-side fault; then `timed_calls` back-to-back calls are timed
    with HIP events (max over ranks).  Collective: every rank calls it with the same arguments.  Returns
    {"ok": bool, "us": float | None, "why": str}."""
    from . import _hip
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    world = comm.world
    is_p2p = getattr(comm, "kind", "rccl") == "p2p"
    ok, why = True, "ok"
    _hip.lib().erl_async_fault_count(1)
    if is_p2p:
        comm.set_spin(1 << 20)                # a peer that never shows up costs ~a second here, not the run-time bound
    g = th.Generator(device="cuda").manual_seed(4242 + rank)
    for it in range(rounds):
        n = count if it % 3 != 2 else max(1, count // 3 + it)
        x = th.randn(n, device="cuda", generator=g) * (1.0 + it)
        d = th.randn(8, device="cuda", generator=g, dtype=th.float64)
        ref, dref = _reference_sum(x), _reference_sum(d)          # the process group's collectives first, in lockstep ...
        good, err = False, ""
        try:                                                      # ... then the route under test, whose failure stays local
            y = comm.all_reduce_sum(x.clone())
            dy = comm.all_reduce_sum(d.clone())
            th.cuda.synchronize()
            faults = _hip.lib().erl_async_fault_count(1)
            good = (faults == 0 and bool(th.allclose(y, ref, rtol=1e-5, atol=1e-5 * (1.0 + it) * world))
                    and bool(th.allclose(dy, dref, rtol=1e-12, atol=1e-12 * world)))
            err = "device-side wait timed out" if faults else "sums disagree with torch.distributed"
        except Exception as e:                                    # noqa: BLE001 -- a route that raises is a route that is not selected
            err = f"raised {type(e).__name__}: {e}"
        if not _agree(good):                                      # ... until the all-ranks verdict of the round
            ok, why = False, f"round {it}: " + (err if not good else "failed on another rank")
            break
    us = None
    if ok:
        buf = th.zeros(count, dtype=th.float32, device="cuda")
        for _ in range(10):
            comm.all_reduce_sum(buf)
        th.cuda.synchronize()
        barrier()
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(timed_calls):
            comm.all_reduce_sum(buf)
        e1.record()
        th.cuda.synchronize()
        us = all_reduce_max_float(e0.elapsed_time(e1) * 1e3 / timed_calls, device="cuda")
        if not _agree(_hip.lib().erl_async_fault_count(1) == 0):
            ok, why, us = False, "device-side wait timed out in the timed calls", None
    if is_p2p:
        comm.set_spin(0)
    return {"ok": ok, "us": None if us is None else round(us, 2), "why": why}


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def probe_p2p_out_of_process(count: int, timeout_s: float = 180.0) -> str:
    """First contact with the peer-to-peer route in a throw-away child process per rank (elegantrl_amd/p2p_probe.py): a route
    whose peer mapping faults on this machine kills the CHILD, never the run.  Collective over the default process group.
    Returns "ok" or the reason the route is not to be used (agreed by all ranks)."""
    import subprocess
    import sys
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [_free_port() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"),
               MASTER_PORT=str(box[0]), LOCAL_RANK=str(th.cuda.current_device()), ERL_P2P_PROBE_COUNT=str(int(count)))
    for k in [k for k in env if k.startswith("TORCHELASTIC_") or k in ("ERL_FORCE_DP", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME")]:
        env.pop(k)                        # (under torchrun the child must host its own store on the new port, not look for the agent's)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    why = "ok"
    try:
        out = subprocess.run([sys.executable, "-m", "elegantrl_amd.p2p_probe"], env=env, cwd=root, capture_output=True, text=True,
                             timeout=timeout_s)
        if out.returncode != 0:
            why = f"probe process exited with {out.returncode}: {(out.stderr or out.stdout)[-200:].strip()}"
    except subprocess.TimeoutExpired:
        why = f"probe process timed out after {timeout_s:.0f} s"
    if not _agree(why == "ok"):
        return why if why != "ok" else "probe failed on another rank"
    return "ok"


# the peer-to-peer route folds the exchange into the slab-reduction launch; RCCL needs two more launches per minibatch
# (the collective + the partial norms): that much slower in the stand-alone timing still wins in the loop
_P2P_LAUNCH_CREDIT_US = 8.0


def gradient_comm(count: Optional[int] = None) -> Optional[RcclComm]:
    """The job's library communicator for the per-minibatch gradient exchange and the advantage sums; None means "use
    torch.distributed" (CPU/gloo runs without a usable library route, or ERL_DP_COLLECTIVE=torch).

    ERL_DP_COLLECTIVE = auto (default) | p2p | rccl | torch.  `auto` brings up BOTH library routes -- RCCL (only when the
    process group is "nccl": one rank per GPU) and the one-shot peer-to-peer exchange -- runs `selftest` on each (agreement
    with torch.distributed's sums on every rank, no device-side timeouts) and keeps the faster of the ones that passed; a
    route that fails its self-test anywhere is dropped on every rank, so an unvalidated path can never carry a run.
    `count` = floats of the gradient row (the agent's slab stride).  ERL_FORCE_DP=1 builds 1-rank communicators without
    world > 1 (measures the data-parallel loop on one GPU)."""
    global _grad_comm, _grad_comm_tried, _route_report
    if _grad_comm_tried:
        return _grad_comm
    _grad_comm_tried = True
    mode = os.environ.get("ERL_DP_COLLECTIVE", "auto")
    report = {"mode": mode, "selected": "torch.distributed", "rccl_us": None, "p2p_us": None, "rccl_selftest": "not tried",
              "p2p_selftest": "not tried", "p2p_probe": "not run", "ranks_seen_by_rccl": None}
    _route_report = report
    if mode == "torch" or not th.cuda.is_available() or not (is_distributed() or force_dp()):
        return None
    count = int(count or 50848)
    up = dist.is_available() and dist.is_initialized()
    want_rccl = mode in ("auto", "rccl") and (not up or dist.get_backend() == "nccl")     # gloo groups share GPUs: RCCL cannot
    want_p2p = mode in ("auto", "p2p")
    cands = {}
    if want_rccl:
        c = RcclComm.create()
        if c is None:
            report["rccl_selftest"] = "communicator did not come up on every rank"
        else:
            report["ranks_seen_by_rccl"] = c.world
            t = selftest(c, count)
            report["rccl_selftest"], report["rccl_us"] = t["why"], t["us"]
            if t["ok"]:
                cands["rccl"] = (c, t["us"])
            else:
                c.close()
    if want_p2p and up and dist.get_world_size() > 1 and os.environ.get("ERL_P2P_PROBE", "1") != "0":
        # first contact out of process: a faulting peer mapping must not take the run down with it
        report["p2p_probe"] = probe_p2p_out_of_process(count)
        if report["p2p_probe"] != "ok":
            report["p2p_selftest"] = "not run: " + report["p2p_probe"]
            want_p2p = False
    if want_p2p:
        c = P2PComm.create(max_count=max(count, P2PComm.MAX_COUNT))
        if c is None:
            report["p2p_selftest"] = "HIP IPC stages could not be created / mapped on every rank"
        else:
            t = selftest(c, count)
            report["p2p_selftest"], report["p2p_us"] = t["why"], t["us"]
            if t["ok"]:
                cands["p2p"] = (c, t["us"])
            else:
                c.close()
    if not cands:
        return None
    pick = "p2p" if "p2p" in cands and ("rccl" not in cands or cands["p2p"][1] <= cands["rccl"][1] + _P2P_LAUNCH_CREDIT_US) else "rccl"
    for name, (c, _) in cands.items():
        if name != pick:
            barrier()
            c.close()
    _grad_comm = cands[pick][0]
    report["selected"] = ("library one-shot peer-to-peer exchange inside the slab-reduction launch (csrc/grad_tail.hip)" if pick == "p2p"
                          else "library RCCL communicator on the kernels' stream")
    return _grad_comm


def shutdown() -> None:
    """Orderly end of a data-parallel job, called by every rank: drain the GPU, meet at a barrier, destroy the library's
    RCCL communicator (an intra-node collective in RCCL, hence the barrier first), then the process group."""
    global _grad_comm, _grad_comm_tried
    up = dist.is_available() and dist.is_initialized()
    if th.cuda.is_available() and (up or _grad_comm is not None):
        th.cuda.synchronize()
    if up:
        _barrier()
    if _grad_comm is not None:
        _grad_comm.close()
    _grad_comm, _grad_comm_tried = None, False
    if up:
        dist.destroy_process_group()


def all_reduce_sum(t: th.Tensor) -> th.Tensor:
    """in-place SUM over ranks (identity when not distributed)."""
    if is_distributed():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_max_float(x: float, device=None) -> float:
    if not is_distributed():
        return x
    t = th.tensor([x], dtype=th.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_(t: th.Tensor, src: int = 0) -> th.Tensor:
    if is_distributed():
        dist.broadcast(t, src=src)
    return t


def shard_range(total: int, rank: int, world: int) -> range:
    """contiguous shard of `total` units for `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def _barrier():
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[th.cuda.current_device()])
    else:
        dist.barrier()


def barrier():
    if is_distributed():
        _barrier()
