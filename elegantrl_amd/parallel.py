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


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun).
    Returns (rank, world_size, local_rank).  No-op for world_size 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # "nccl" IS RCCL on ROCm; ERL_DIST_BACKEND=gloo lets several ranks share one GPU (tests)
            backend = os.environ.get("ERL_DIST_BACKEND") or ("nccl" if th.cuda.is_available() else "gloo")
        if th.cuda.is_available():
            local_rank = local_rank % th.cuda.device_count()
            th.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


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


def barrier():
    if is_distributed():
        dist.barrier()
