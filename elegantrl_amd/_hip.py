"""ctypes binding of liberl_hip.so (the C ABI declared in include/erl_hip.h).

This is the only way the package reaches the GPU kernels.  There is NO CPU or PyTorch fallback:
if the shared library is missing, or a kernel reports an error, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import functools
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p
from typing import Optional

import torch as th

_HERE = os.path.dirname(os.path.abspath(__file__))
# ERL_HIP_LIB selects another build of the same library (profiling / A-B builds); it is still this HIP library or nothing
LIB_PATH = os.environ.get("ERL_HIP_LIB") or os.path.join(_HERE, "lib", "liberl_hip.so")

# flags mirrored from include/erl_hip.h
GAE_VTRACE, GAE_MUTATE, GAE_STATS = 0x1, 0x2, 0x4
GAE_ALGO_AUTO, GAE_ALGO_EXACT, GAE_ALGO_CHUNKED, GAE_ALGO_LOOKBACK = 0x00, 0x10, 0x20, 0x30
MAX_STATE_DIM, MAX_HIDDEN, MAX_ACTION_DIM = 128, 128, 16
MAX_LAYERS, MAXN_WIDTH = 6, 4096
ABI_VERSION = 19
PPO_OBJ_REFERENCE, PPO_OBJ_CANONICAL, PPO_OBJ_A2C = 0, 1, 2      # include/erl_hip.h ERL_PPO_OBJ_*
SAC_ACTOR_SAC, SAC_ACTOR_FIX = 0, 1                               # include/erl_hip.h ERL_SAC_ACTOR_*
COMM_ID_BYTES = 128
P2P_HANDLE_BYTES = 64

_P = c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "erl_abi_version": (c_int, []),
    "erl_last_error_string": (c_char_p, []),
    "erl_device_info": (c_int, [POINTER(c_int), POINTER(c_int)]),
    "erl_gae_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "erl_gae_scan_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_float, c_float, c_int, _P, _P, c_int64, _P]),
    "erl_async_fault_count": (c_int, [c_int]),
    "erl_cum_rewards_f32": (c_int, [_P, _P, _P, _P, c_int64, c_int64, c_float, _P]),
    "erl_adv_stats_f32": (c_int, [_P, c_int64, c_int64, _P, _P, c_int64, _P]),
    "erl_adv_normalize_f32": (c_int, [_P, _P, c_int64, c_int64, _P, _P]),
    "erl_adv_stats_fold_f32": (c_int, [_P, c_int, c_int64, c_int64, _P, _P]),
    "erl_rollout_gae_partials": (c_int, [c_int64]),
    "erl_split_ids_i64": (c_int, [_P, c_int64, c_int64, _P, _P, _P]),
    "erl_ppo_gather_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int, c_int, _P, c_int64,
                                   _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "erl_replay_write_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int64, c_int64, c_int, c_int,
                                     c_int64, c_int64, _P]),
    "erl_replay_sample_f32": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int64, c_int, c_int, _P, c_int64, c_int64,
                                      _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "erl_replay_row_floats": (c_int64, [c_int, c_int]),
    "erl_replay_write_rows_f32": (c_int, [_P, c_int64, c_int64, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int64, c_int64, _P]),
    "erl_replay_sample_rows_f32": (c_int, [_P, c_int64, c_int64, c_int, c_int, _P, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "erl_replay_write_discrete_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int64, c_int64, c_int, c_int64,
                                              c_int64, _P]),
    "erl_replay_sample_discrete_f32": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int64, c_int, _P, c_int64, c_int64,
                                               _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "erl_per_tree_floats": (c_int64, [c_int64, c_int64]),
    "erl_per_init_f32": (c_int, [_P, _P, c_int64, c_int64, _P]),
    "erl_per_add_rows_f32": (c_int, [_P, _P, c_int64, c_int64, c_int64, c_int64, c_float, _P]),
    "erl_per_update_f32": (c_int, [_P, _P, c_int64, c_int64, _P, _P, _P, c_int64, c_float, _P]),
    "erl_per_sample_f32": (c_int, [_P, _P, c_int64, c_int64, _P, c_int64, c_int64, c_int64, c_float, _P, _P, _P]),
    "erl_mlp_param_count": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "erl_value_forward_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int64, _P, _P]),
    "erl_rollout_step_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int64, _P, c_uint64, c_uint64,
                                     _P, _P, _P, _P, _P]),
    "erl_rollout_fused_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "erl_rollout_gae_workspace_bytes": (c_int64, [c_int64]),
    "erl_rollout_synenv_f32": (c_int, [_P] * 6 + [c_int] * 4 + [_P] * 5 + [c_int, c_uint64, c_int64, c_int64, _P, c_uint64, c_uint64,
                                       c_float] + [_P] * 8 + [_P] * 5 + [c_int64, c_float, c_float, c_int, _P]),
    "erl_rollout_pendulum_f32": (c_int, [_P] * 6 + [c_int] * 2 + [_P] * 4 + [c_int, c_uint64, c_int64, c_int64, _P, c_uint64,
                                         c_uint64, c_float] + [_P] * 8 + [_P] * 5 + [c_int64, c_float, c_float, c_int, _P]),
    "erl_ppo_slab_stride": (c_int64, [c_int, c_int, c_int, c_int]),
    "erl_ppo_num_slabs": (c_int, [c_int64]),
    "erl_ppo_set_arith": (c_int, [c_int]),
    "erl_ppo_arith_in_use": (c_int, [c_int, c_int, c_int, c_int]),
    "erl_k6_timing_spans": (c_int, [c_int, POINTER(c_int64), POINTER(c_double), c_int]),
    "erl_ppo_wg_map_info": (c_int, [c_int, POINTER(c_int), POINTER(c_double), POINTER(c_double)]),
    "erl_ppo_update_chains": (c_int, []),
    "erl_ppo_step_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P,
                                 c_int64, c_int64, _P, c_int64, c_float, c_float, c_float, c_int, _P, c_int, _P]),
    "erl_grad_reduce_f32": (c_int, [_P, c_int, c_int64, _P, _P]),
    "erl_clip_adam_f32": (c_int, [_P, _P, _P, _P, POINTER(c_int64), POINTER(c_int64), c_int, _P, c_int32, c_float,
                                  c_float, c_float, c_float, c_float, c_float, _P]),
    "erl_reduce_clip_adam_f32": (c_int, [_P, c_int, c_int64, _P, _P, _P, _P, POINTER(c_int64), POINTER(c_int64), c_int, c_int32, c_float,
                                         c_float, c_float, c_float, c_float, c_float, _P]),
    "erl_reduce_clip_adam_grid_f32": (c_int, [_P, c_int, c_int64, _P, _P, _P, _P, POINTER(c_int64), POINTER(c_int64), c_int, c_int32, c_float,
                                              c_float, c_float, c_float, c_float, c_float, _P]),
    "erl_reduce_clip_adam_grid_ok": (c_int, [c_int64]),
    "erl_tail_fused_ok": (c_int, [c_int64]),
    "erl_reduce_clip_adam_fused_f32": (c_int, [_P, c_int, c_int64, _P, _P, _P, _P, POINTER(c_int64), POINTER(c_int64), c_int, c_int32, c_float,
                                               c_float, c_float, c_float, c_float, c_float, _P]),
    "erl_ppo_update_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int64, c_int64,
                                   _P, c_int64, c_int, c_float, c_float, c_int, _P, _P, c_int32, c_float, c_float, c_float, c_float,
                                   c_float, _P]),
    "erl_comm_unique_id": (c_int, [_P]),
    "erl_comm_init": (c_int, [_P, c_int, c_int, POINTER(c_void_p)]),
    "erl_comm_destroy": (c_int, [_P]),
    "erl_comm_world_size": (c_int, [_P]),
    "erl_comm_allreduce_sum_f32": (c_int, [_P, _P, c_int64, _P]),
    "erl_comm_allreduce_sum_f64": (c_int, [_P, _P, c_int64, _P]),
    "erl_comm_kind": (c_int, [_P]),
    "erl_comm_reduce_exchange_f32": (c_int, [_P, _P, c_int, c_int64, _P, POINTER(c_int64), POINTER(c_int64), c_int, c_float, _P]),
    "erl_grad_reduce_partials_f32": (c_int, [_P, c_int, c_int64, _P, POINTER(c_int64), POINTER(c_int64), c_int, c_float, _P]),
    "erl_ppo_logs_mean_f32": (c_int, [_P, c_int64, c_int64, c_int, c_float, _P, _P]),
    "erl_ppo_finish_f32": (c_int, [_P, c_int64, c_int64, c_int, c_float, _P, _P, _P, _P, _P, c_int64, _P]),
    "erl_grad_sq_partials_f32": (c_int, [_P, c_int64, POINTER(c_int64), POINTER(c_int64), c_int, c_float, _P]),
    "erl_clip_adam_partials_f32": (c_int, [_P, _P, _P, _P, c_int64, POINTER(c_int64), POINTER(c_int64), c_int, c_int32, c_float,
                                           c_float, c_float, c_float, c_float, c_float, _P]),
    "erl_comm_clip_adam_partials_f32": (c_int, [_P, _P, _P, _P, _P, c_int64, POINTER(c_int64), POINTER(c_int64), c_int, c_int32, c_float,
                                                c_float, c_float, c_float, c_float, c_float, _P]),
    "erl_comm_p2p_create": (c_int, [c_int, c_int, c_int64, POINTER(c_void_p), _P]),
    "erl_comm_p2p_connect": (c_int, [_P, _P]),
    "erl_comm_p2p_set_spin": (c_int, [_P, ctypes.c_uint32]),
    "erl_ppo_update_dp_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int64, c_int64,
                                      _P, c_int64, c_int, c_float, c_float, c_int, _P, _P, c_int32, c_float, c_float, c_float,
                                      c_float, c_float, _P, _P, c_int, _P, _P]),
    "erl_mlpn_param_count": (c_int64, [POINTER(c_int), c_int, c_int]),
    "erl_mlpn_workspace_bytes": (c_int64, [POINTER(c_int), c_int, c_int64, c_int]),
    "erl_mlpn_value_forward_f32": (c_int, [_P, _P, _P, POINTER(c_int), c_int, _P, c_int64, _P, _P, c_int64, _P]),
    "erl_mlpn_rollout_step_f32": (c_int, [_P, _P, _P, POINTER(c_int), c_int, _P, c_int64, _P, c_uint64, c_uint64, _P, _P, _P, _P,
                                          _P, c_int64, _P]),
    "erl_mlpn_ppo_step_f32": (c_int, [_P, _P, _P, _P, _P, _P, POINTER(c_int), c_int, _P, _P, _P, _P, _P, _P, c_int64, c_int64, _P,
                                      c_int64, c_float, c_float, c_float, c_int, _P, _P, c_int64, _P]),
    "erl_mlpn_rollout_step_discrete_f32": (c_int, [_P, _P, _P, POINTER(c_int), c_int, _P, c_int64, _P, c_uint64, c_uint64, _P, _P,
                                                   _P, _P, _P, c_int64, _P]),
    "erl_mlpn_ppo_step_discrete_f32": (c_int, [_P, _P, _P, _P, _P, _P, POINTER(c_int), c_int, _P, _P, _P, _P, _P, _P, c_int64,
                                               c_int64, _P, c_int64, c_float, c_float, c_float, _P, _P, c_int64, _P]),
    "erl_sac_param_counts": (c_int, [c_int, c_int, POINTER(c_int), c_int, c_int, POINTER(c_int64), POINTER(c_int64)]),
    "erl_sac_workspace_bytes": (c_int64, [c_int, c_int, POINTER(c_int), c_int, c_int, c_int64]),
    "erl_sac_update_f32": (c_int, [_P] * 10 + [c_int, c_int, POINTER(c_int), c_int, c_int] + [_P] * 9 + [c_float, c_int64, _P, _P, c_uint64,
                                   c_uint64] + [c_float] * 8 + [c_int32, _P, _P, c_int64, _P]),
    "erl_sac_update_opt_f32": (c_int, [_P] * 10 + [c_int, c_int, POINTER(c_int), c_int, c_int] + [_P] * 9 + [c_float, c_int64, _P, _P, c_uint64,
                                       c_uint64] + [c_float] * 8 + [c_int32, _P, _P, c_int64, _P, _P]),
    "erl_sac_update_ring_f32": (c_int, [_P] * 10 + [c_int, c_int, POINTER(c_int), c_int, c_int, _P] + [_P] * 6 + [c_int64, _P, _P, c_uint64,
                                        c_uint64] + [c_float] * 8 + [c_int32, _P, _P, c_int64, _P]),
    "erl_sac_update_ring_loop_f32": (c_int, [_P] * 10 + [c_int, c_int, POINTER(c_int), c_int, c_int, _P, _P, c_int64] + [_P] * 6 + [c_int64, c_uint64,
                                             c_uint64] + [c_float] * 8 + [c_int32, _P, _P, c_int64, _P]),
    "erl_sac_explore_action_f32": (c_int, [_P, c_int, c_int, POINTER(c_int), c_int, _P, c_int64, _P, c_uint64, c_uint64, _P, _P, _P,
                                           c_int64, _P]),
    "erl_sac_explore_action_opt_f32": (c_int, [_P, c_int, c_int, POINTER(c_int), c_int, _P, c_int64, _P, c_uint64, c_uint64, _P, _P, _P,
                                               c_int64, c_int, _P]),
    "erl_sac_rollout_synenv_supported": (c_int, [c_int, c_int, POINTER(c_int), c_int, c_int64]),
    "erl_sac_rollout_synenv_f32": (c_int, [_P, c_int, c_int, POINTER(c_int), c_int, _P, _P, _P, _P, _P, c_int, c_uint64, c_int64, c_int64, _P,
                                           c_uint64, c_uint64, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "erl_sac_rollout_pendulum_f32": (c_int, [_P, POINTER(c_int), c_int, _P, _P, _P, _P, c_int, c_uint64, c_int64, c_int64, _P, c_uint64, c_uint64,
                                             c_float, _P, _P, _P, _P, _P, _P, _P]),
    "erl_k6_timing_enable": (None, [c_int]),
    "erl_k6_timing_read": (c_int, [POINTER(ctypes.c_double), POINTER(c_int)]),
    "erl_k6_timing_read2": (c_int, [POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int)]),
    "erl_k6_timing_null_bracket_us": (c_int, [_P, c_int, POINTER(ctypes.c_double)]),
    "erl_k6_timing_last_records": (c_int, [c_int, POINTER(ctypes.c_uint64), c_int]),
    "erl_kernel_span_enable": (None, [c_int]),
    "erl_kernel_span_read": (c_int, [c_int, POINTER(ctypes.c_double), POINTER(c_int)]),
    "erl_k6_timing_clocks": (c_int, [c_int, POINTER(ctypes.c_double), POINTER(c_int), POINTER(ctypes.c_double), POINTER(ctypes.c_double),
                                     POINTER(ctypes.c_double), c_int, POINTER(c_int), POINTER(c_int)]),
    "erl_synenv_step_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_uint64, _P]),
    "erl_pendulum_step_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_uint64, _P]),
    "erl_selftest_mfma": (c_int, [POINTER(c_float)]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib: Optional[ctypes.CDLL] = None


class HipExtensionError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load liberl_hip.so once; fail loudly if it is absent (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionError(
                f"{LIB_PATH} not found: build it with `make -C elegantrl_amd/csrc` (or __graft_entry__.build()). "
                "elegantrl_amd has no CPU/PyTorch fallback for its kernels.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if handle.erl_abi_version() != ABI_VERSION:
            raise HipExtensionError(f"liberl_hip.so ABI {handle.erl_abi_version()} != binding ABI {ABI_VERSION}")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().erl_last_error_string()
        raise HipExtensionError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def stream_ptr() -> int:
    """hipStream_t of torch's current stream ON THE CURRENT DEVICE (kernels are ordered with torch ops on that stream).
    The C ABI launches on the calling thread's current HIP device, so every tensor handed to it must live there: `ptr`
    enforces it, and the agents / buffers / envs enter their own device first (`on_device`)."""
    try:                                  # (the raw handle without building a torch.cuda.Stream object: ~0.3 us instead of ~8)
        return th._C._cuda_getCurrentRawStream(th._C._cuda_getDevice())
    except AttributeError:                # pragma: no cover  (a torch build without the private accessors)
        return th.cuda.current_stream().cuda_stream


def on_device(method):
    """run a method of an object with a `.device` attribute with that device current (`args.gpu_id` != 0 in a single
    process: the library's launches, its stream lookup and its library-owned buffers all follow the current device)."""
    @functools.wraps(method)
    def inner(self, *a, **k):
        dev = getattr(self, "device", None)
        if dev is None or dev.type != "cuda" or dev.index is None or dev.index == th.cuda.current_device():
            return method(self, *a, **k)
        with th.cuda.device(dev):
            return method(self, *a, **k)
    return inner


try:
    _current_device = th._C._cuda_getDevice      # torch.cuda.current_device() without its lazy-init bookkeeping (called per tensor argument)
except AttributeError:                           # pragma: no cover
    _current_device = th.cuda.current_device


def ptr(t: Optional[th.Tensor], dtype: Optional[th.dtype] = None) -> Optional[int]:
    """Raw device pointer of a contiguous CUDA(HIP) tensor on the current device; None passes NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipExtensionError("elegantrl_amd kernels need tensors on a HIP device (no CPU path); got a CPU tensor")
    if t.device.index != _current_device():
        raise HipExtensionError(f"tensor lives on {t.device} but the current HIP device is cuda:{th.cuda.current_device()}: "
                                "kernels launch on the current device (enter `th.cuda.device(...)` / use the agent's methods)")
    if not t.is_contiguous():
        raise HipExtensionError("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise HipExtensionError(f"expected dtype {dtype}, got {t.dtype}")
    return t.data_ptr()


def flag_ptr(t: th.Tensor) -> int:
    """bool/uint8 flags are passed as bytes."""
    if t.dtype not in (th.bool, th.uint8):
        raise HipExtensionError(f"flag tensor must be bool/uint8, got {t.dtype}")
    return ptr(t)


def check_async_faults() -> None:
    """raise if a kernel recorded a device-side fault since the last check (call after a stream synchronisation; reads a
    pinned host block, no GPU work): the look-back GAE scan's bounded wait (csrc/gae_lookback.hip), the peer-to-peer gradient
    exchange's wait for a peer (csrc/grad_tail.hip), the clip + Adam grid wait (csrc/optim.hip) -- the message says which."""
    n = lib().erl_async_fault_count(1)
    if n:
        msg = lib().erl_last_error_string()
        raise HipExtensionError(f"device-side fault ({n}): {msg.decode() if msg else '?'}")


def device_info():
    cu, lds = c_int(0), c_int(0)
    check(lib().erl_device_info(ctypes.byref(cu), ctypes.byref(lds)), "erl_device_info")
    return cu.value, lds.value


def k6_timing_enable(every_nth: int) -> None:
    """bracket every n-th K6 launch with HIP events on its stream (0 / False = off, 1 / True = every launch)."""
    lib().erl_k6_timing_enable(int(every_nth))


def k6_timing_read():
    """(seconds summed over the K6 launches recorded since the last read, number of launches)."""
    ms, n = ctypes.c_double(0), c_int(0)
    check(lib().erl_k6_timing_read(ctypes.byref(ms), ctypes.byref(n)), "erl_k6_timing_read")
    return ms.value * 1e-3, n.value


def k6_timing_read2():
    """(event-bracket seconds, in-kernel span seconds, number of launches) summed over the K6 launches sampled since the last read:
    the HIP-event bracket contains the bracket's own dispatch / completion overhead, the span is first workgroup in to last
    workgroup out on the device's constant-rate clock."""
    ev, sp, n = ctypes.c_double(0), ctypes.c_double(0), c_int(0)
    check(lib().erl_k6_timing_read2(ctypes.byref(ev), ctypes.byref(sp), ctypes.byref(n)), "erl_k6_timing_read2")
    return ev.value * 1e-3, sp.value * 1e-3, n.value


SPAN_GAE, SPAN_REPLAY_SAMPLE, SPAN_SAC_CRITIC_TRAIN, SPAN_SLAB_REDUCE, SPAN_CLIP_ADAM, SPAN_ROLLOUT = range(6)      # include/erl_hip.h ERL_SPAN_*


def kernel_span_enable(every_nth) -> None:
    """every n-th launch of a tagged kernel leaves its own first-workgroup-in to last-workgroup-out span on the device clock (True = every
    launch, 0 / False = off); enabling clears earlier records"""
    lib().erl_kernel_span_enable(int(every_nth))


def kernel_span_read(tag: int):
    """(mean span in microseconds or None, number of launches) of the tagged kernel's sampled launches since the hook was ENABLED (the
    records are not cleared by a read: enable again to start a new window; launches on another device than the enabling one leave none)"""
    us, n = ctypes.c_double(0), c_int(0)
    check(lib().erl_kernel_span_read(int(tag), ctypes.byref(us), ctypes.byref(n)), "erl_kernel_span_read")
    return (us.value / n.value if n.value else None), n.value


K6_PHASES = ("prologue", "layer1_forward", "layer2_forward", "output_objective_backward", "stage_dW1", "stage_dW3_stage", "dW2_logs_drain")


def k6_timing_clocks(bracketed: bool = False):
    """what the launches drained by the last k6_timing_read2() say about the box, for the launches WITHOUT an event bracket around them
    (default: the kernel as the loop runs it) or the bracketed ones: {"launches": n, "span_us": mean first-workgroup-in to
    last-workgroup-out span, "shader_mhz": the clock they ran at, "workgroup_us": a workgroup's mean duration, "phase_cycles": {phase:
    mean shader cycles of an actor workgroup's first wave} (empty for kernels that stamp no phases), "phase_workgroups": n}"""
    mhz, wg, sp = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
    ph = (ctypes.c_double * 8)()
    n, nw, nl = c_int(0), c_int(0), c_int(0)
    check(lib().erl_k6_timing_clocks(int(bool(bracketed)), ctypes.byref(sp), ctypes.byref(nl), ctypes.byref(mhz), ctypes.byref(wg), ph, 8,
                                     ctypes.byref(n), ctypes.byref(nw)), "erl_k6_timing_clocks")
    return {"launches": nl.value, "span_us": sp.value * 1e3 / nl.value if nl.value else None, "shader_mhz": mhz.value, "workgroup_us": wg.value,
            "phase_cycles": {K6_PHASES[k]: ph[k] for k in range(min(n.value, len(K6_PHASES)))}, "phase_workgroups": nw.value}


def k6_timing_last_records(bracketed: bool = False):
    """diagnostics: per-workgroup records of the group's last sampled K6 launch as a list of dicts (start offset and duration in us on
    the constant-rate clock, shader cycles, where it ran: xcc / se / sh / cu ids from HW_REG_HW_ID / HW_REG_XCC_ID); actor workgroups first"""
    buf = (ctypes.c_uint64 * (8 * 1024))()
    n = lib().erl_k6_timing_last_records(int(bool(bracketed)), buf, 1024)
    recs = []
    t0 = min((buf[8 * i] for i in range(n) if buf[8 * i + 1] > buf[8 * i]), default=0)
    for i in range(n):
        w0, w1, m0, m1, hw = buf[8 * i], buf[8 * i + 1], buf[8 * i + 2], buf[8 * i + 3], buf[8 * i + 7]
        if w1 <= w0:
            continue
        hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
        recs.append({"wg": i, "start_us": (w0 - t0) / 100.0, "dur_us": (w1 - w0) / 100.0, "cycles": m1 - m0, "xcc": xcc,
                     "se": (hwid >> 13) & 0x7, "sh": (hwid >> 12) & 1, "cu": (hwid >> 8) & 0xf, "simd": (hwid >> 4) & 3})
    return recs


def k6_timing_spans(bracketed: bool = False, max_launches: int = 65536):
    """[(launch number since k6_timing_enable, span in microseconds)] of every sampled launch the last k6_timing_read2 drained"""
    idx, us = (c_int64 * max_launches)(), (c_double * max_launches)()
    n = lib().erl_k6_timing_spans(1 if bracketed else 0, idx, us, max_launches)
    return [(int(idx[i]), float(us[i])) for i in range(n)]


def k6_wg_summary(recs):
    """compact reading of k6_timing_last_records(): did every workgroup get a compute unit of its own at once?  (a launch of 256
    workgroups that each need a whole CU -- 155 KB of LDS, 512 registers per lane -- runs in ONE round only if 256 CUs take them)"""
    if not recs:
        return None
    import statistics
    cus = {}
    for r in recs:
        cus.setdefault((r["xcc"], r["se"], r["sh"], r["cu"]), []).append(r)
    shared = {k: v for k, v in cus.items() if len(v) > 1}
    durs, starts = sorted(r["dur_us"] for r in recs), sorted(r["start_us"] for r in recs)
    per_xcc = {}
    for r in recs:
        per_xcc.setdefault(r["xcc"], []).append(r["dur_us"])
    return {"workgroups": len(recs), "distinct_cus": len(cus), "cus_hosting_more_than_one_workgroup": len(shared),
            "workgroups_in_a_second_round": sum(len(v) - 1 for v in shared.values()),
            "start_us": {"median": round(statistics.median(starts), 2), "p90": round(starts[int(0.9 * (len(starts) - 1))], 2), "max": round(starts[-1], 2)},
            "dur_us": {"min": round(durs[0], 2), "median": round(statistics.median(durs), 2), "max": round(durs[-1], 2)},
            "end_us_max": round(max(r["start_us"] + r["dur_us"] for r in recs), 2),
            "late_starters": sorted(({"wg": r["wg"], "start_us": round(r["start_us"], 1), "dur_us": round(r["dur_us"], 1), "xcc": r["xcc"], "se": r["se"],
                                      "cu": r["cu"]} for r in recs if r["start_us"] > 5.0), key=lambda x: -x["start_us"])[:12],
            "dur_us_mean_by_xcc": {str(k): round(sum(v) / len(v), 2) for k, v in sorted(per_xcc.items())},
            "workgroups_by_xcc": {str(k): len(v) for k, v in sorted(per_xcc.items())},
            # every workgroup: [index (actor's first), xcc, se, sh, cu, duration in 0.1 us] -- CU pairs share an instruction cache
            "table": [[r["wg"], r["xcc"], r["se"], r["sh"], r["cu"], int(round(r["dur_us"] * 10))] for r in recs]}


def ppo_wg_map_info(device: int = None, wide: bool = False) -> dict:
    """the minibatch kernel's workgroup map on `device` (default: the current one; wide: of the (256, h2[, h3]) kernels): {'map': 0 | 1 | None (not decided yet), 'forced': the
    ERL_K6_WG_MAP override or None, 'us_map0', 'us_map2': the per-launch times the device's one-off measurement saw (None: never measured)}"""
    import torch as th
    dev = th.cuda.current_device() if device is None else int(device)
    m, u0, u1 = c_int(-1), c_double(0), c_double(0)
    check(lib().erl_ppo_wg_map_info(dev | (0x100 if wide else 0), ctypes.byref(m), ctypes.byref(u0), ctypes.byref(u1)), "erl_ppo_wg_map_info")
    env = os.environ.get("ERL_K6_WG_MAP", "")
    return {"map": None if m.value < 0 else m.value, "forced": int(env) if env in ("0", "1", "2") else None,
            "us_map0": round(u0.value, 2) if u0.value else None, "us_map2": round(u1.value, 2) if u1.value else None}


def ppo_update_chains() -> int:
    """the form the last PPO update loop of this process took: 1 = one chain of launches, 2 = one chain per network on two streams
    (include/erl_hip.h erl_ppo_update_chains), 0 = none yet"""
    return int(lib().erl_ppo_update_chains())


def k6_null_bracket_us(reps: int = 200) -> float:
    """median HIP-event bracket around an EMPTY launch on the current stream (microseconds): what a bracket adds to its content."""
    us = ctypes.c_double(0)
    check(lib().erl_k6_timing_null_bracket_us(stream_ptr(), int(reps), ctypes.byref(us)), "erl_k6_timing_null_bracket_us")
    return us.value


def selftest_mfma() -> float:
    err = c_float(0)
    check(lib().erl_selftest_mfma(ctypes.byref(err)), "erl_selftest_mfma")
    return float(err.value)
