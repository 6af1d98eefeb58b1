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
        from . import _hip
        _hip.check(_hip.lib().erl_comm_allreduce_sum_f32(self.handle, _hip.ptr(t, th.float32), t.numel(), _hip.stream_ptr()),
                   "erl_comm_allreduce_sum_f32")
        return t

    def close(self):
        if self.handle:
            from . import _hip
            th.cuda.synchronize()
            _hip.lib().erl_comm_destroy(self.handle)
            self.handle = None


class P2PComm(RcclComm):
    """One-shot peer-to-peer all-reduce (csrc/p2p.hip; prototype, ERL_DP_COLLECTIVE=p2p): the same library handle type as the
    RCCL communicator, so `all_reduce_sum` and the C update loop take it unchanged.  torch.distributed carries every rank's
    64-byte IPC handle to every rank and the all-ranks agreement that the peers' stages are mapped everywhere."""

    kind = "p2p"
    MAX_COUNT = 1 << 22            # floats per all-reduce the stages are sized for (16 MiB per half)

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


def gradient_comm() -> Optional[RcclComm]:
    """The job's RCCL communicator for the per-minibatch gradient exchange; None means "use torch.distributed"
    (CPU/gloo runs, ERL_DP_COLLECTIVE=torch, or RCCL did not come up on every rank).  ERL_FORCE_DP=1 builds a 1-rank
    communicator without a process group (measures the data-parallel loop on one GPU)."""
    global _grad_comm, _grad_comm_tried
    if _grad_comm_tried:
        return _grad_comm
    _grad_comm_tried = True
    mode = os.environ.get("ERL_DP_COLLECTIVE", "rccl")
    if mode == "p2p" and th.cuda.is_available() and (is_distributed() or force_dp()):
        _grad_comm = P2PComm.create()        # None (on every rank) when IPC / peer mapping is unavailable: torch.distributed then
        return _grad_comm
    if mode != "rccl" or not th.cuda.is_available():
        return None
    if is_distributed():
        if dist.get_backend() != "nccl":        # several ranks on one GPU (gloo tests): RCCL cannot span duplicates
            return None
    elif not force_dp():
        return None
    _grad_comm = RcclComm.create()
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
