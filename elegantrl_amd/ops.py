"""Tensor-level wrappers of the C ABI (include/erl_hip.h).  One function per kernel entry point.

All tensors must live on a HIP device; outputs are allocated with torch (the library itself never
allocates).  Every wrapper raises `HipExtensionError` on failure -- there is no fallback path.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch as th

from . import _hip
from ._hip import check, flag_ptr, lib, ptr, stream_ptr

TEN = th.Tensor
_ALGO = {"auto": _hip.GAE_ALGO_AUTO, "exact": _hip.GAE_ALGO_EXACT, "chunked": _hip.GAE_ALGO_CHUNKED,
         "lookback": _hip.GAE_ALGO_LOOKBACK}

_workspaces = {}


def _workspace(device: th.device, nbytes: int) -> TEN:
    """scratch block for a launch on the CURRENT stream of `device`: keyed by (device, stream), so that two agents driving two
    streams of one device never share scratch (launches that share a block are ordered by their stream)"""
    key = (device.type, device.index, stream_ptr())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = th.empty(max(nbytes, 1 << 22), dtype=th.uint8, device=device)
        _workspaces[key] = ws
    return ws


# ------------------------------------------------------------------------------------------------
# K3 / K4
# ------------------------------------------------------------------------------------------------
def gae_scan(rewards: TEN, undones: TEN, unmasks: TEN, values: TEN, next_value: TEN, gamma: float, lam: float, *,
             use_v_trace: bool = True, mutate: bool = True, algo: str = "auto", with_ret: bool = True,
             stats: Optional[TEN] = None, adv: Optional[TEN] = None, ret: Optional[TEN] = None):
    """AgentPPO.get_advantages (+ reward_sums).  Returns (advantages, reward_sums or None).
    With mutate=True `rewards`/`undones` receive the truncation fix-up in place, like the reference."""
    H, N = rewards.shape
    assert undones.shape == unmasks.shape == values.shape == (H, N) and next_value.shape == (N,)
    adv = th.empty_like(values) if adv is None else adv
    if with_ret and ret is None:
        ret = th.empty_like(values)
    flags = (_hip.GAE_VTRACE if use_v_trace else 0) | (_hip.GAE_MUTATE if mutate else 0) | _ALGO[algo]
    if stats is not None:
        assert stats.dtype == th.float64 and stats.numel() >= 5
        flags |= _hip.GAE_STATS
    nbytes = lib().erl_gae_workspace_bytes(H, N)
    ws = _workspace(rewards.device, nbytes)
    check(lib().erl_gae_scan_f32(ptr(rewards, th.float32), flag_ptr(undones), flag_ptr(unmasks), ptr(values, th.float32),
                                 ptr(next_value, th.float32), ptr(adv, th.float32), ptr(ret if with_ret else None),
                                 H, N, gamma, lam, flags, ptr(stats), ptr(ws), ws.numel(), stream_ptr()),
          "erl_gae_scan_f32")
    return adv, (ret if with_ret else None)


def adv_stats_fold(partials: TEN, n_partials: int, H: int, N: int, stats: TEN) -> TEN:
    """n_partials x 3 fp64 partial sums (the persistent rollout's epilogue) -> the 8-double block of raw sums (AgentPPO.py:149)."""
    check(lib().erl_adv_stats_fold_f32(ptr(partials, th.float64), int(n_partials), H, N, ptr(stats, th.float64), stream_ptr()),
          "erl_adv_stats_fold_f32")
    return stats


def cum_rewards(rewards: TEN, undones: TEN, next_value: TEN, gamma: float, out: Optional[TEN] = None) -> TEN:
    """AgentBase.get_cumulative_rewards' backward scan (AgentBase.py:226-237): rewards / undones (H, N) f32, next_value (N,)."""
    H, N = rewards.shape
    assert undones.shape == (H, N) and next_value.numel() == N
    out = th.empty_like(rewards) if out is None else out
    check(lib().erl_cum_rewards_f32(ptr(rewards, th.float32), ptr(undones, th.float32), ptr(next_value, th.float32),
                                    ptr(out, th.float32), H, N, gamma, stream_ptr()),
          "erl_cum_rewards_f32")
    return out


def adv_stats(adv: TEN, stats: Optional[TEN] = None) -> TEN:
    """Raw sums for AgentPPO.py:149 -> float64[5] = (sum, H*N, sum_sub, sumsq_sub, count_sub)."""
    H, N = adv.shape
    stats = th.empty(8, dtype=th.float64, device=adv.device) if stats is None else stats
    ws = _workspace(adv.device, lib().erl_gae_workspace_bytes(H, N))
    check(lib().erl_adv_stats_f32(ptr(adv, th.float32), H, N, ptr(stats, th.float64), ptr(ws), ws.numel(), stream_ptr()),
          "erl_adv_stats_f32")
    return stats


def adv_normalize(adv: TEN, stats: TEN, out: Optional[TEN] = None) -> TEN:
    H, N = adv.shape
    out = th.empty_like(adv) if out is None else out
    check(lib().erl_adv_normalize_f32(ptr(adv, th.float32), ptr(out, th.float32), H, N, ptr(stats, th.float64), stream_ptr()),
          "erl_adv_normalize_f32")
    return out


# ------------------------------------------------------------------------------------------------
# K5
# ------------------------------------------------------------------------------------------------
def split_ids(ids: TEN, sample_len: int) -> Tuple[TEN, TEN]:
    ids0, ids1 = th.empty_like(ids), th.empty_like(ids)
    check(lib().erl_split_ids_i64(ptr(ids, th.int64), ids.numel(), sample_len, ptr(ids0), ptr(ids1), stream_ptr()),
          "erl_split_ids_i64")
    return ids0, ids1


def ppo_gather(states: TEN, actions: TEN, unmasks: TEN, logprobs: TEN, advantages: TEN, reward_sums: TEN, ids: TEN):
    """x[ids % H, ids // H] for the six PPO buffers (AgentPPO.py:178-187) + the index pair."""
    H, N, S = states.shape
    A = actions.shape[2]
    B = ids.numel()
    dev = states.device
    o_s = th.empty((B, S), dtype=th.float32, device=dev)
    o_a = th.empty((B, A), dtype=th.float32, device=dev)
    o_u = th.empty((B,), dtype=th.bool, device=dev)
    o_l, o_ad, o_r = (th.empty((B,), dtype=th.float32, device=dev) for _ in range(3))
    i0, i1 = th.empty_like(ids), th.empty_like(ids)
    check(lib().erl_ppo_gather_f32(ptr(states, th.float32), ptr(actions, th.float32), flag_ptr(unmasks), ptr(logprobs, th.float32),
                                   ptr(advantages, th.float32), ptr(reward_sums, th.float32), H, N, S, A, ptr(ids, th.int64), B,
                                   ptr(o_s), ptr(o_a), ptr(o_u), ptr(o_l), ptr(o_ad), ptr(o_r), ptr(i0), ptr(i1), stream_ptr()),
          "erl_ppo_gather_f32")
    return (o_s, o_a, o_u, o_l, o_ad, o_r), (i0, i1)


# ------------------------------------------------------------------------------------------------
# K8 / K9
# ------------------------------------------------------------------------------------------------
def replay_write(buf_states: TEN, buf_actions: TEN, buf_rewards: TEN, buf_undones: TEN, buf_unmasks: TEN,
                 items: Sequence[TEN], p: int) -> None:
    """ring append (ReplayBuffer.update).  A uint8 `buf_actions` (max_size, num_seqs) is a discrete-action ring: `actions`
    arrive as (add, num_seqs) int32 (AgentBase.py:146) and their low byte is stored (replay_buffer.py:53-54)."""
    states, actions, rewards, undones, unmasks = items
    max_size, num_seqs, S = buf_states.shape
    add = rewards.shape[0]
    is_f32 = undones.dtype == th.float32
    assert unmasks.dtype == undones.dtype and (is_f32 or undones.dtype in (th.bool, th.uint8))
    if buf_actions.dtype == th.uint8:
        assert states.shape == (add, num_seqs, S) and actions.shape == (add, num_seqs) and buf_actions.shape == (max_size, num_seqs)
        check(lib().erl_replay_write_discrete_f32(ptr(buf_states, th.float32), ptr(buf_actions, th.uint8), ptr(buf_rewards, th.float32),
                                                  ptr(buf_undones, th.float32), ptr(buf_unmasks, th.float32), ptr(states, th.float32),
                                                  ptr(actions, th.int32), ptr(rewards, th.float32), ptr(undones), ptr(unmasks),
                                                  int(is_f32), max_size, num_seqs, S, p, add, stream_ptr()),
              "erl_replay_write_discrete_f32")
        return
    A = buf_actions.shape[2]
    assert states.shape == (add, num_seqs, S) and actions.shape == (add, num_seqs, A)
    check(lib().erl_replay_write_f32(ptr(buf_states, th.float32), ptr(buf_actions, th.float32), ptr(buf_rewards, th.float32),
                                     ptr(buf_undones, th.float32), ptr(buf_unmasks, th.float32), ptr(states, th.float32),
                                     ptr(actions, th.float32), ptr(rewards, th.float32), ptr(undones), ptr(unmasks), int(is_f32),
                                     max_size, num_seqs, S, A, p, add, stream_ptr()),
          "erl_replay_write_f32")


class ReplayRing:
    """the INTERLEAVED replay ring (ABI 18): one fp32 block [num_seqs][max_size][row_floats], row = [state | action | reward | undone |
    unmask | pad to 4 floats], sequence-major so that a transition and its next state are consecutive in memory (one or two 128-byte
    lines per sample instead of six to seven).  `states`, `actions`, `rewards`, `undones`, `unmasks` are the reference's attributes
    (elegantrl/train/replay_buffer.py:40-58: shapes (max_size, num_seqs[, .]), fp32) as strided VIEWS of the block: indexing, slicing and
    assignment work on them as on the reference's tensors; they are not contiguous."""

    def __init__(self, max_size: int, num_seqs: int, S: int, A: int, device):
        self.max_size, self.num_seqs, self.S, self.A = int(max_size), int(num_seqs), int(S), int(A)
        self.row_floats = int(lib().erl_replay_row_floats(self.S, self.A))
        if self.row_floats <= 0:
            raise HipExtensionError(f"erl_replay_row_floats({S}, {A}) failed")
        self.block = th.zeros((self.num_seqs, self.max_size, self.row_floats), dtype=th.float32, device=device)
        t = self.block.permute(1, 0, 2)                                        # (max_size, num_seqs, row_floats)
        self.states, self.actions = t[:, :, :S], t[:, :, S:S + A]
        self.rewards, self.undones, self.unmasks = t[:, :, S + A], t[:, :, S + A + 1], t[:, :, S + A + 2]

    def write(self, items: Sequence[TEN], p: int) -> None:
        """ring append (ReplayBuffer.update, replay_buffer.py:86-105) of (add, num_seqs, .) items at time row p, wrapping at max_size"""
        states, actions, rewards, undones, unmasks = items
        add = rewards.shape[0]
        is_f32 = undones.dtype == th.float32
        assert unmasks.dtype == undones.dtype and (is_f32 or undones.dtype in (th.bool, th.uint8))
        assert states.shape == (add, self.num_seqs, self.S) and actions.shape == (add, self.num_seqs, self.A)
        check(lib().erl_replay_write_rows_f32(ptr(self.block, th.float32), self.max_size, self.num_seqs, self.S, self.A, ptr(states, th.float32),
                                              ptr(actions, th.float32), ptr(rewards, th.float32), ptr(undones), ptr(unmasks), int(is_f32),
                                              int(p), add, stream_ptr()), "erl_replay_write_rows_f32")

    def sample(self, ids: TEN, sample_len: int, stage: Optional["ReplayStage"] = None):
        """ReplayBuffer.sample given the drawn ids: ((state, action, reward, undone, unmask, next_state), (ids0, ids1))"""
        B = ids.numel()
        st = stage if stage is not None else ReplayStage(B, self.S, self.A, False, self.block.device)
        assert st.B == B and not st.discrete
        check(lib().erl_replay_sample_rows_f32(ptr(self.block, th.float32), self.max_size, self.num_seqs, self.S, self.A, ptr(ids, th.int64), B,
                                               int(sample_len), st.p_state, st.p_action, st.p_reward, st.p_undone, st.p_unmask, st.p_next,
                                               st.p_ids0, st.p_ids1, stream_ptr()), "erl_replay_sample_rows_f32")
        return st.out, st.ids


class ReplayStage:
    """the output block of ReplayBuffer.sample for one batch size -- ONE fp32 allocation viewed as state / next_state / action /
    reward / undone / unmask, one int64 (2, B) for ids0 / ids1 and, for a discrete ring, a (B,) uint8 action vector -- with
    its raw addresses computed once.  A buffer that reuses its stage (ReplayBuffer.sample(..., reuse=True), the off-policy
    update loop) samples without touching the allocator or building views: at batch 256 that was most of the call."""

    def __init__(self, B: int, S: int, A: int, discrete: bool, device):
        self.B, self.discrete = B, discrete
        A_f = 0 if discrete else A
        self.flat = th.empty(B * (2 * S + A_f + 3), dtype=th.float32, device=device)
        o_s, o_n, o_a, o_r, o_ud, o_um = th.split(self.flat, [B * S, B * S, B * A_f, B, B, B])
        self.act_u8 = th.empty(B, dtype=th.uint8, device=device) if discrete else None
        self.i01 = th.empty((2, B), dtype=th.int64, device=device)
        self.out = (o_s.view(B, S), self.act_u8 if discrete else o_a.view(B, A), o_r, o_ud, o_um, o_n.view(B, S))
        self.ids = (self.i01[0], self.i01[1])
        base, ib = self.flat.data_ptr(), self.i01.data_ptr()
        self.p_state, self.p_next = base, base + 4 * B * S
        self.p_action = self.act_u8.data_ptr() if discrete else base + 8 * B * S
        self.p_reward = base + 4 * B * (2 * S + A_f)
        self.p_undone, self.p_unmask = self.p_reward + 4 * B, self.p_reward + 8 * B
        self.p_ids0, self.p_ids1 = ib, ib + 8 * B


def replay_sample(buf_states: TEN, buf_actions: TEN, buf_rewards: TEN, buf_undones: TEN, buf_unmasks: TEN, ids: TEN,
                  sample_len: int, stage: Optional[ReplayStage] = None):
    """ReplayBuffer.sample given the drawn ids: ((state, action, reward, undone, unmask, next_state), (ids0, ids1)).
    The six outputs are views of ONE allocation (and the two index vectors of another); pass a ReplayStage to reuse it."""
    max_size, num_seqs, S = buf_states.shape
    discrete = buf_actions.dtype == th.uint8
    A = 1 if discrete else buf_actions.shape[2]
    B = ids.numel()
    st = stage if stage is not None else ReplayStage(B, S, A, discrete, buf_states.device)
    assert st.B == B and st.discrete == discrete
    if discrete:
        check(lib().erl_replay_sample_discrete_f32(ptr(buf_states, th.float32), ptr(buf_actions, th.uint8), ptr(buf_rewards, th.float32),
                                                   ptr(buf_undones, th.float32), ptr(buf_unmasks, th.float32), max_size, num_seqs, S,
                                                   ptr(ids, th.int64), B, sample_len, st.p_state, st.p_action, st.p_reward,
                                                   st.p_undone, st.p_unmask, st.p_next, st.p_ids0, st.p_ids1, stream_ptr()),
              "erl_replay_sample_discrete_f32")
    else:
        check(lib().erl_replay_sample_f32(ptr(buf_states, th.float32), ptr(buf_actions, th.float32), ptr(buf_rewards, th.float32),
                                          ptr(buf_undones, th.float32), ptr(buf_unmasks, th.float32), max_size, num_seqs, S, A,
                                          ptr(ids, th.int64), B, sample_len, st.p_state, st.p_action, st.p_reward, st.p_undone,
                                          st.p_unmask, st.p_next, st.p_ids0, st.p_ids1, stream_ptr()),
              "erl_replay_sample_f32")
    return st.out, st.ids


# ------------------------------------------------------------------------------------------------
# prioritised replay (row f2): device-resident sum / min trees, csrc/per.hip
# ------------------------------------------------------------------------------------------------
class PerTrees:
    """one sum tree + one min tree per sequence as implicit heaps (num_seqs, 2 L) on the device."""

    def __init__(self, max_size: int, num_seqs: int, device):
        n = lib().erl_per_tree_floats(max_size, num_seqs)
        if n <= 0:
            raise HipExtensionError(f"erl_per_tree_floats({max_size}, {num_seqs}) failed")
        self.max_size, self.num_seqs = int(max_size), int(num_seqs)
        self.sum = th.empty(n, dtype=th.float32, device=device)
        self.min = th.empty(n, dtype=th.float32, device=device)
        check(lib().erl_per_init_f32(ptr(self.sum), ptr(self.min), self.max_size, self.num_seqs, stream_ptr()), "erl_per_init_f32")
        self.leaves = n // (2 * self.num_seqs)

    def add_rows(self, start: int, add: int, prob: float = 10.0) -> None:
        check(lib().erl_per_add_rows_f32(ptr(self.sum), ptr(self.min), self.max_size, self.num_seqs, int(start), int(add), float(prob),
                                         stream_ptr()), "erl_per_add_rows_f32")

    def update(self, ids0: TEN, ids1: TEN, td_error: TEN, per_alpha: float) -> None:
        check(lib().erl_per_update_f32(ptr(self.sum), ptr(self.min), self.max_size, self.num_seqs, ptr(ids0, th.int64),
                                       ptr(ids1, th.int64), ptr(td_error, th.float32), ids0.numel(), float(per_alpha), stream_ptr()),
              "erl_per_update_f32")

    def sample(self, uniform: TEN, cur_size: int, per_beta: float, cursor: int = -1):
        """uniform (num_seqs, n) in [0, 1) -> (is_indices (num_seqs * n,) int64 = ids1 * cur_size + ids0, is_weights float32);
        `cursor`: the ring's write position when the ring is full (the newest row then has no valid successor), else -1"""
        assert uniform.shape[0] == self.num_seqs and uniform.dtype == th.float32
        n = uniform.shape[1]
        idx = th.empty(self.num_seqs * n, dtype=th.int64, device=uniform.device)
        w = th.empty(self.num_seqs * n, dtype=th.float32, device=uniform.device)
        check(lib().erl_per_sample_f32(ptr(self.sum), ptr(self.min), self.max_size, self.num_seqs, ptr(uniform.contiguous(), th.float32),
                                       n, int(cur_size), int(cursor), float(per_beta), ptr(idx), ptr(w), stream_ptr()), "erl_per_sample_f32")
        return idx, w


# ------------------------------------------------------------------------------------------------
# MLP kernels (K1, K2, K6, K7)
# ------------------------------------------------------------------------------------------------
@dataclass
class MlpSpec:
    """Shape of a 2-hidden-layer build_mlp([S, h1, h2, out]) network in one flat fp32 buffer
    (W1,b1,W2,b2,W3,b3[,action_std_log]); see include/erl_hip.h."""
    S: int
    h1: int
    h2: int
    out: int
    with_std_log: bool

    @property
    def count(self) -> int:
        n = lib().erl_mlp_param_count(self.S, self.h1, self.h2, self.out, int(self.with_std_log))
        if n < 0:
            raise _hip.HipExtensionError(
                f"unsupported network for the fused HIP kernels: state_dim={self.S} net_dims=[{self.h1},{self.h2}] "
                f"out={self.out} (need 2 hidden layers, multiples of 32 and <= {_hip.MAX_HIDDEN}; state_dim <= "
                f"{_hip.MAX_STATE_DIM}; action_dim <= {_hip.MAX_ACTION_DIM})")
        return n

    def slices(self):
        """(name, offset, shape) in flat order."""
        out, o = [], 0
        for name, shape in (("net.0.weight", (self.h1, self.S)), ("net.0.bias", (self.h1,)),
                            ("net.2.weight", (self.h2, self.h1)), ("net.2.bias", (self.h2,)),
                            ("net.4.weight", (self.out, self.h2)), ("net.4.bias", (self.out,))):
            n = 1
            for s in shape:
                n *= s
            out.append((name, o, shape))
            o += n
        if self.with_std_log:
            out.append(("action_std_log", o, (1, self.out)))
        return out


def value_forward(params: TEN, spec: MlpSpec, state_avg: TEN, state_std: TEN, states: TEN, out: Optional[TEN] = None) -> TEN:
    """CriticPPO(states).squeeze(-1) for states (..., S)."""
    rows = states.numel() // spec.S
    out = th.empty(states.shape[:-1], dtype=th.float32, device=states.device) if out is None else out
    check(lib().erl_value_forward_f32(ptr(params, th.float32), ptr(state_avg, th.float32), ptr(state_std, th.float32), spec.S,
                                      spec.h1, spec.h2, ptr(states, th.float32), rows, ptr(out, th.float32), stream_ptr()),
          "erl_value_forward_f32")
    return out


def rollout_step(params: TEN, spec: MlpSpec, state_avg: TEN, state_std: TEN, state: TEN, *, noise: Optional[TEN] = None,
                 seed: int = 0, counter: int = 0, out_state: Optional[TEN] = None, out_action: Optional[TEN] = None,
                 out_logprob: Optional[TEN] = None, out_env_action: Optional[TEN] = None) -> None:
    N = state.shape[0]
    check(lib().erl_rollout_step_f32(ptr(params, th.float32), ptr(state_avg, th.float32), ptr(state_std, th.float32), spec.S,
                                     spec.h1, spec.h2, spec.out, ptr(state, th.float32), N, ptr(noise), seed & (2 ** 64 - 1),
                                     counter & (2 ** 64 - 1), ptr(out_state), ptr(out_action), ptr(out_logprob),
                                     ptr(out_env_action), stream_ptr()),
          "erl_rollout_step_f32")


def ppo_slab_stride(S: int, h1: int, h2: int, A: int) -> int:
    return lib().erl_ppo_slab_stride(S, h1, h2, A)


def ppo_num_slabs(batch_size: int) -> int:
    """number of per-workgroup gradient slabs erl_ppo_step_f32 writes for a minibatch of `batch_size` samples."""
    return lib().erl_ppo_num_slabs(batch_size)


PPO_ARITH = {"auto": 0, "f32": 1, "split": 2}


def ppo_set_arith(arith: str) -> str:
    """arithmetic of K6's large products: "f32" (fp32 MFMA), "split" (three-way bf16 operand split on the bf16 matrix pipe,
    fp32-equivalent) or "auto" (library default / ERL_K6_ARITH).  Returns the previous setting."""
    prev = lib().erl_ppo_set_arith(PPO_ARITH[arith])
    return {v: k for k, v in PPO_ARITH.items()}[prev]


def ppo_arith_in_use(S: int, h1: int, h2: int, A: int) -> str:
    """which arithmetic erl_ppo_step_f32 uses for this shape under the current setting ("f32" or "split")."""
    return {1: "f32", 2: "split"}[lib().erl_ppo_arith_in_use(S, h1, h2, A)]


def ppo_step(actor_params: TEN, critic_params: TEN, act_avg: TEN, act_std: TEN, cri_avg: TEN, cri_std: TEN, S: int, h1: int,
             h2: int, A: int, states: TEN, actions: TEN, unmasks: TEN, logprobs: TEN, advantages: TEN, reward_sums: TEN,
             ids: TEN, ratio_clip: float, lambda_entropy: float, inv_batch: float, slabs: TEN, n_slabs: int,
             objective: int = 0) -> None:
    """`objective`: _hip.PPO_OBJ_REFERENCE (AgentPPO.py:199) / PPO_OBJ_CANONICAL (textbook clip) / PPO_OBJ_A2C (AgentPPO.py:296-303)"""
    H, N = states.shape[0], states.shape[1]
    need = n_slabs * lib().erl_ppo_slab_stride(S, h1, h2, A)
    if slabs.numel() < need:
        raise ValueError(f"slabs holds {slabs.numel()} floats, erl_ppo_step_f32 writes n_slabs x erl_ppo_slab_stride = {need}")
    check(lib().erl_ppo_step_f32(ptr(actor_params, th.float32), ptr(critic_params, th.float32), ptr(act_avg), ptr(act_std),
                                 ptr(cri_avg), ptr(cri_std), S, h1, h2, A, ptr(states, th.float32), ptr(actions, th.float32),
                                 flag_ptr(unmasks), ptr(logprobs, th.float32), ptr(advantages, th.float32),
                                 ptr(reward_sums, th.float32), H, N, ptr(ids, th.int64), ids.numel(), ratio_clip,
                                 lambda_entropy, inv_batch, int(objective), ptr(slabs, th.float32), n_slabs, stream_ptr()),
          "erl_ppo_step_f32")


def grad_reduce(slabs: TEN, n_slabs: int, stride: int, flat_grad: TEN) -> None:
    check(lib().erl_grad_reduce_f32(ptr(slabs, th.float32), n_slabs, stride, ptr(flat_grad, th.float32), stream_ptr()),
          "erl_grad_reduce_f32")


def clip_adam(params: TEN, grads: TEN, exp_avg: TEN, exp_avg_sq: TEN, groups: Sequence[Tuple[int, int]], step: int, lr: float,
              max_norm: float, grad_scale: float = 1.0, betas=(0.9, 0.999), eps: float = 1e-8,
              step_base: Optional[TEN] = None) -> None:
    n = len(groups)
    off = (ctypes.c_int64 * n)(*[g[0] for g in groups])
    ln = (ctypes.c_int64 * n)(*[g[1] for g in groups])
    check(lib().erl_clip_adam_f32(ptr(params, th.float32), ptr(grads, th.float32), ptr(exp_avg, th.float32),
                                  ptr(exp_avg_sq, th.float32), off, ln, n, ptr(step_base), step, lr, betas[0], betas[1], eps,
                                  max_norm, grad_scale, stream_ptr()),
          "erl_clip_adam_f32")


def _groups_c(groups):
    n = len(groups)
    return n, (ctypes.c_int64 * n)(*[g[0] for g in groups]), (ctypes.c_int64 * n)(*[g[1] for g in groups])


def grad_reduce_partials(slabs: TEN, n_slabs: int, stride: int, flat_grad: TEN, groups: Sequence[Tuple[int, int]],
                         grad_scale: float = 1.0, comm=None) -> None:
    """launch 1 of the two-launch optimiser tail: slab reduction [+ the data-parallel exchange through `comm`] + the partial
    norms that `clip_adam_partials` consumes (library-owned table, same stream)."""
    n, off, ln = _groups_c(groups)
    check(lib().erl_comm_reduce_exchange_f32(None if comm is None else comm.handle, ptr(slabs, th.float32), n_slabs, stride,
                                             ptr(flat_grad, th.float32), off, ln, n, grad_scale, stream_ptr()),
          "erl_comm_reduce_exchange_f32")


def grad_sq_partials(grads: TEN, stride: int, groups: Sequence[Tuple[int, int]], grad_scale: float = 1.0) -> None:
    """partial norms of an already-summed gradient row (after a torch.distributed / RCCL all-reduce)."""
    n, off, ln = _groups_c(groups)
    check(lib().erl_grad_sq_partials_f32(ptr(grads, th.float32), stride, off, ln, n, grad_scale, stream_ptr()), "erl_grad_sq_partials_f32")


def clip_adam_partials(params: TEN, grads: TEN, exp_avg: TEN, exp_avg_sq: TEN, stride: int, groups: Sequence[Tuple[int, int]],
                       step: int, lr: float, max_norm: float, grad_scale: float = 1.0, betas=(0.9, 0.999), eps: float = 1e-8,
                       comm=None) -> None:
    """launch 2: clip_grad_norm_ from the partial norms left by launch 1 + Adam (AgentBase.py:246-248).  With the peer-to-peer
    `comm` of launch 1: skipped (nothing touched) when an exchange of this update loop timed out (include/erl_hip.h)."""
    n, off, ln = _groups_c(groups)
    check(lib().erl_comm_clip_adam_partials_f32(None if comm is None else comm.handle, ptr(params, th.float32), ptr(grads, th.float32),
                                                ptr(exp_avg, th.float32), ptr(exp_avg_sq, th.float32), stride, off, ln, n, step, lr,
                                                betas[0], betas[1], eps, max_norm, grad_scale, stream_ptr()),
          "erl_comm_clip_adam_partials_f32")


def reduce_clip_adam(slabs: TEN, n_slabs: int, stride: int, flat_grad: TEN, params: TEN, exp_avg: TEN, exp_avg_sq: TEN,
                     groups: Sequence[Tuple[int, int]], step: int, lr: float, max_norm: float, grad_scale: float = 1.0,
                     betas=(0.9, 0.999), eps: float = 1e-8, grid_wait: bool = False) -> None:
    """grad_reduce + clip_adam in one launch (the single-process minibatch loop's tail).  grid_wait: every workgroup waits for
    the norm and updates its own elements (erl_reduce_clip_adam_grid_f32; needs `reduce_clip_adam_grid_ok(stride)`), instead
    of the last-arriving workgroup applying Adam alone."""
    n = len(groups)
    off = (ctypes.c_int64 * n)(*[g[0] for g in groups])
    ln = (ctypes.c_int64 * n)(*[g[1] for g in groups])
    fn = lib().erl_reduce_clip_adam_grid_f32 if grid_wait else lib().erl_reduce_clip_adam_f32
    check(fn(ptr(slabs, th.float32), n_slabs, stride, ptr(flat_grad, th.float32), ptr(params, th.float32), ptr(exp_avg, th.float32),
             ptr(exp_avg_sq, th.float32), off, ln, n, step, lr, betas[0], betas[1], eps, max_norm, grad_scale, stream_ptr()),
          "erl_reduce_clip_adam_grid_f32" if grid_wait else "erl_reduce_clip_adam_f32")


def reduce_clip_adam_fused(slabs: TEN, n_slabs: int, stride: int, flat_grad: TEN, params: TEN, exp_avg: TEN, exp_avg_sq: TEN,
                           groups: Sequence[Tuple[int, int]], step: int, lr: float, max_norm: float, grad_scale: float = 1.0,
                           betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """grad_reduce_partials + clip_adam_partials (the default two-launch tail) as ONE launch with the same bits
    (erl_reduce_clip_adam_fused_f32: the partial norms are their own flags; needs `tail_fused_ok(stride)`)"""
    n = len(groups)
    off = (ctypes.c_int64 * n)(*[g[0] for g in groups])
    ln = (ctypes.c_int64 * n)(*[g[1] for g in groups])
    check(lib().erl_reduce_clip_adam_fused_f32(ptr(slabs, th.float32), n_slabs, stride, ptr(flat_grad, th.float32), ptr(params, th.float32),
                                               ptr(exp_avg, th.float32), ptr(exp_avg_sq, th.float32), off, ln, n, step, lr, betas[0], betas[1], eps,
                                               max_norm, grad_scale, stream_ptr()), "erl_reduce_clip_adam_fused_f32")


def tail_fused_ok(stride: int) -> bool:
    return bool(lib().erl_tail_fused_ok(stride))


def reduce_clip_adam_grid_ok(stride: int) -> bool:
    return bool(lib().erl_reduce_clip_adam_grid_ok(stride))


def ppo_update(flat_params: TEN, exp_avg: TEN, exp_avg_sq: TEN, act_avg: TEN, act_std: TEN, cri_avg: TEN, cri_std: TEN, S: int,
               h1: int, h2: int, A: int, states: TEN, actions: TEN, unmasks: TEN, logprobs: TEN, advantages: TEN, reward_sums: TEN,
               ids: TEN, ratio_clip: float, lambda_entropy: float, slabs: TEN, grads: TEN, first_step: int, lr: float,
               max_norm: float, betas=(0.9, 0.999), eps: float = 1e-8, comm=None, objective: int = 0, adv_stats: Optional[TEN] = None,
               adv_partials: Optional[TEN] = None, n_partials: int = 0) -> None:
    """the whole minibatch loop of AgentPPO.update_net in one C call; ids: (update_times, B).  `comm` (a
    parallel.RcclComm) puts the gradient all-reduce inside the loop, on the same stream (data-parallel ranks).
    `adv_stats` (8 float64: the raw sums left by gae_scan(stats=...) or the rollout's epilogue): `advantages` are then RAW
    and every minibatch kernel normalises them at its row load (AgentPPO.py:149) instead of a separate launch.  `adv_partials`
    (n_partials x 3 float64, the rollout epilogue's workspace): the sums are folded into `adv_stats` by the loop's first launch."""
    H, N = states.shape[0], states.shape[1]
    update_times, B = ids.shape
    assert grads.shape[0] >= update_times and slabs.shape[0] == ppo_num_slabs(B)
    check(lib().erl_ppo_update_dp_f32(ptr(flat_params, th.float32), ptr(exp_avg, th.float32), ptr(exp_avg_sq, th.float32),
                                      ptr(act_avg), ptr(act_std), ptr(cri_avg), ptr(cri_std), S, h1, h2, A, ptr(states, th.float32),
                                      ptr(actions, th.float32), flag_ptr(unmasks), ptr(logprobs, th.float32),
                                      ptr(advantages, th.float32), ptr(reward_sums, th.float32), H, N, ptr(ids, th.int64), B,
                                      update_times, ratio_clip, lambda_entropy, int(objective), ptr(slabs, th.float32),
                                      ptr(grads, th.float32),
                                      first_step, lr, betas[0], betas[1], eps, max_norm,
                                      None if adv_stats is None else ptr(adv_stats, th.float64),
                                      None if adv_partials is None else ptr(adv_partials, th.float64), int(n_partials),
                                      None if comm is None else comm.handle, stream_ptr()),
          "erl_ppo_update_dp_f32")


# ------------------------------------------------------------------------------------------------
# generic-shape MLP path (any depth / width): erl_mlpn_*
# ------------------------------------------------------------------------------------------------
class MlpSpecN:
    """build_mlp([S, d1, ..., dL, out]) of any depth in one flat fp32 buffer (W1 b1 ... Wout bout [action_std_log])."""

    def __init__(self, dims: Sequence[int], with_std_log: bool):
        self.dims = [int(d) for d in dims]
        self.with_std_log = bool(with_std_log)
        self.S, self.out = self.dims[0], self.dims[-1]
        self._c = (ctypes.c_int * len(self.dims))(*self.dims)

    @property
    def cdims(self):
        return self._c, len(self.dims)

    @property
    def count(self) -> int:
        n = lib().erl_mlpn_param_count(self._c, len(self.dims), int(self.with_std_log))
        if n < 0:
            raise _hip.HipExtensionError(f"unsupported network dims {self.dims}: at most {_hip.MAX_LAYERS} hidden layers of width "
                                         f"<= {_hip.MAXN_WIDTH}")
        return n

    def slices(self):
        out, o = [], 0
        for i, (d_in, d_out) in enumerate(zip(self.dims[:-1], self.dims[1:])):
            out.append((f"net.{2 * i}.weight", o, (d_out, d_in)))
            o += d_out * d_in
            out.append((f"net.{2 * i}.bias", o, (d_out,)))
            o += d_out
        if self.with_std_log:
            out.append(("action_std_log", o, (1, self.out)))
        return out

    def workspace_bytes(self, rows: int, training: bool) -> int:
        return lib().erl_mlpn_workspace_bytes(self._c, len(self.dims), rows, int(training))


_MLPN_VALUE_WS_BYTES = 256 << 20     # activation workspace cap of the layered value pre-pass


def mlpn_value_forward(params: TEN, spec: MlpSpecN, state_avg: TEN, state_std: TEN, states: TEN, out: Optional[TEN] = None) -> TEN:
    """CriticPPO(states).squeeze(-1) on the layered path.  The pass is cut into row chunks whose activations fit a fixed
    256 MiB workspace (the reference chunks this pre-pass too, `bs = 2 ** 10 // num_envs` time rows, AgentPPO.py:141-143:
    H x N x sum(dims) floats at 2048 x 4096 would be ~15 GB in one piece)."""
    rows = states.numel() // spec.S
    out = th.empty(states.shape[:-1], dtype=th.float32, device=states.device) if out is None else out
    per_row = max(1, spec.workspace_bytes(4096, False) // 4096)
    chunk = max(4096, (_MLPN_VALUE_WS_BYTES // per_row) // 4096 * 4096)
    c, n = spec.cdims
    s2, o1 = states.reshape(-1, spec.S), out.reshape(-1)
    for r0 in range(0, max(rows, 1), chunk):
        nr = min(chunk, rows - r0)
        if nr <= 0:
            break
        ws = _workspace(states.device, spec.workspace_bytes(nr, False))
        check(lib().erl_mlpn_value_forward_f32(ptr(params, th.float32), ptr(state_avg, th.float32), ptr(state_std, th.float32), c, n,
                                               ptr(s2, th.float32) + 4 * r0 * spec.S, nr, ptr(o1, th.float32) + 4 * r0, ptr(ws),
                                               ws.numel(), stream_ptr()),
              "erl_mlpn_value_forward_f32")
    return out


def mlpn_rollout_step(params: TEN, spec: MlpSpecN, state_avg: TEN, state_std: TEN, state: TEN, *, noise: Optional[TEN] = None,
                      seed: int = 0, counter: int = 0, out_state: Optional[TEN] = None, out_action: Optional[TEN] = None,
                      out_logprob: Optional[TEN] = None, out_env_action: Optional[TEN] = None) -> None:
    N = state.shape[0]
    ws = _workspace(state.device, spec.workspace_bytes(N, False))
    c, n = spec.cdims
    check(lib().erl_mlpn_rollout_step_f32(ptr(params, th.float32), ptr(state_avg, th.float32), ptr(state_std, th.float32), c, n,
                                          ptr(state, th.float32), N, ptr(noise), seed & (2 ** 64 - 1), counter & (2 ** 64 - 1),
                                          ptr(out_state), ptr(out_action), ptr(out_logprob), ptr(out_env_action), ptr(ws),
                                          ws.numel(), stream_ptr()),
          "erl_mlpn_rollout_step_f32")


def mlpn_ppo_step(actor_params: TEN, critic_params: TEN, act_avg: TEN, act_std: TEN, cri_avg: TEN, cri_std: TEN, spec: MlpSpecN,
                  states: TEN, actions: TEN, unmasks: TEN, logprobs: TEN, advantages: TEN, reward_sums: TEN, ids: TEN,
                  ratio_clip: float, lambda_entropy: float, inv_batch: float, flat_grad: TEN, objective: int = 0) -> None:
    """one PPO minibatch for networks of any depth; `flat_grad` receives [actor | critic | 3 objectives, 0]."""
    H, N = states.shape[0], states.shape[1]
    B = ids.numel()
    ws = _workspace(states.device, spec.workspace_bytes(B, True))
    c, n = spec.cdims
    check(lib().erl_mlpn_ppo_step_f32(ptr(actor_params, th.float32), ptr(critic_params, th.float32), ptr(act_avg), ptr(act_std),
                                      ptr(cri_avg), ptr(cri_std), c, n, ptr(states, th.float32), ptr(actions, th.float32),
                                      flag_ptr(unmasks), ptr(logprobs, th.float32), ptr(advantages, th.float32),
                                      ptr(reward_sums, th.float32), H, N, ptr(ids, th.int64), B, ratio_clip, lambda_entropy,
                                      inv_batch, int(objective), ptr(flat_grad, th.float32), ptr(ws), ws.numel(), stream_ptr()),
          "erl_mlpn_ppo_step_f32")


def mlpn_rollout_step_discrete(params: TEN, spec: MlpSpecN, state_avg: TEN, state_std: TEN, state: TEN, *,
                               uniform: Optional[TEN] = None, seed: int = 0, counter: int = 0, out_state: Optional[TEN] = None,
                               out_action: Optional[TEN] = None, out_logprob: Optional[TEN] = None,
                               out_env_action: Optional[TEN] = None) -> None:
    """one rollout step of the categorical policy: out_action (N,) int32, out_logprob (N,), out_env_action (N,) int64."""
    N = state.shape[0]
    ws = _workspace(state.device, spec.workspace_bytes(N, False))
    c, n = spec.cdims
    check(lib().erl_mlpn_rollout_step_discrete_f32(ptr(params, th.float32), ptr(state_avg, th.float32), ptr(state_std, th.float32), c, n,
                                                   ptr(state, th.float32), N, ptr(uniform, th.float32) if uniform is not None else None,
                                                   seed & (2 ** 64 - 1), counter & (2 ** 64 - 1), ptr(out_state),
                                                   ptr(out_action, th.int32) if out_action is not None else None, ptr(out_logprob),
                                                   ptr(out_env_action, th.int64) if out_env_action is not None else None, ptr(ws),
                                                   ws.numel(), stream_ptr()),
          "erl_mlpn_rollout_step_discrete_f32")


def mlpn_ppo_step_discrete(actor_params: TEN, critic_params: TEN, act_avg: TEN, act_std: TEN, cri_avg: TEN, cri_std: TEN,
                           spec: MlpSpecN, states: TEN, actions: TEN, unmasks: TEN, logprobs: TEN, advantages: TEN, reward_sums: TEN,
                           ids: TEN, ratio_clip: float, lambda_entropy: float, inv_batch: float, flat_grad: TEN) -> None:
    """one PPO minibatch of the categorical policy; actions (H, N) int32."""
    H, N = states.shape[0], states.shape[1]
    B = ids.numel()
    ws = _workspace(states.device, spec.workspace_bytes(B, True))
    c, n = spec.cdims
    check(lib().erl_mlpn_ppo_step_discrete_f32(ptr(actor_params, th.float32), ptr(critic_params, th.float32), ptr(act_avg), ptr(act_std),
                                               ptr(cri_avg), ptr(cri_std), c, n, ptr(states, th.float32), ptr(actions, th.int32),
                                               flag_ptr(unmasks), ptr(logprobs, th.float32), ptr(advantages, th.float32),
                                               ptr(reward_sums, th.float32), H, N, ptr(ids, th.int64), B, ratio_clip, lambda_entropy,
                                               inv_batch, ptr(flat_grad, th.float32), ptr(ws), ws.numel(), stream_ptr()),
          "erl_mlpn_ppo_step_discrete_f32")


# ------------------------------------------------------------------------------------------------
# SAC (erl_sac_*)
# ------------------------------------------------------------------------------------------------
class SacSpec:
    """shapes of ActorSAC / CriticEnsemble as flat fp32 blocks (include/erl_hip.h)."""

    def __init__(self, S: int, A: int, hidden: Sequence[int], num_ensembles: int, actor_variant: int = 0):
        """actor_variant: _hip.SAC_ACTOR_SAC (ActorSAC) or _hip.SAC_ACTOR_FIX (AgentModSAC's ActorFixSAC: same block sizes, a raw last
        encoder layer and the two one-layer decoders as the two row halves of the head)"""
        self.S, self.A, self.hidden, self.E = int(S), int(A), [int(h) for h in hidden], int(num_ensembles)
        self.actor_variant = int(actor_variant)
        self._c = (ctypes.c_int * len(self.hidden))(*self.hidden)
        pa, pc = ctypes.c_int64(0), ctypes.c_int64(0)
        check(lib().erl_sac_param_counts(self.S, self.A, self._c, len(self.hidden), self.E, ctypes.byref(pa), ctypes.byref(pc)),
              "erl_sac_param_counts")
        self.actor_count, self.critic_count = pa.value, pc.value

    def actor_slices(self):
        """(parameter name in ActorSAC, offset, shape) in flat order."""
        out, o, dims = [], 0, [self.S, *self.hidden]
        for i, (d_in, d_out) in enumerate(zip(dims[:-1], dims[1:])):
            out += [(f"net_s.{2 * i}.weight", o, (d_out, d_in)), (f"net_s.{2 * i}.bias", o + d_out * d_in, (d_out,))]
            o += d_out * d_in + d_out
        d_in, d_out = self.hidden[-1], 2 * self.A
        if self.actor_variant:        # ActorFixSAC: encoder_s.* | decoder_a_avg = rows [0, A), decoder_a_std = rows [A, 2A) of the head block
            A = self.A
            out = [(n.replace("net_s.", "encoder_s."), off, shp) for n, off, shp in out]
            out += [("decoder_a_avg.0.weight", o, (A, d_in)), ("decoder_a_std.0.weight", o + A * d_in, (A, d_in)),
                    ("decoder_a_avg.0.bias", o + d_out * d_in, (A,)), ("decoder_a_std.0.bias", o + d_out * d_in + A, (A,))]
            return out
        out += [("net_a.0.weight", o, (d_out, d_in)), ("net_a.0.bias", o + d_out * d_in, (d_out,))]
        return out

    def critic_slices(self):
        out, o = [], 0
        d_in, d_out = self.S + self.A, self.hidden[0]
        out += [("encoder_sa.0.weight", o, (d_out, d_in)), ("encoder_sa.0.bias", o + d_out * d_in, (d_out,))]
        o += d_out * d_in + d_out
        dims = [*self.hidden, 1]
        for e in range(self.E):
            for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
                out += [(f"decoder_q{e:02}.{2 * i}.weight", o, (b, a)), (f"decoder_q{e:02}.{2 * i}.bias", o + b * a, (b,))]
                o += b * a + b
        return out

    def workspace_bytes(self, B: int) -> int:
        return lib().erl_sac_workspace_bytes(self.S, self.A, self._c, len(self.hidden), self.E, B)


def sac_update(spec: SacSpec, actor: TEN, critic: TEN, target: TEN, alpha_log: TEN, moments: Sequence[TEN], batch: Sequence[TEN],
               step: int, *, gamma: float, target_entropy: float, tau: float, lr: float, max_norm: float, objs_out: TEN,
               noises: Optional[Tuple[TEN, TEN]] = None, seed: int = 0, counter: int = 0, betas=(0.9, 0.999), eps: float = 1e-8,
               is_weight: Optional[TEN] = None, td_error_out: Optional[TEN] = None, cum_reward: Optional[TEN] = None,
               lambda_fit_cum_r: float = 0.0, update_actor: bool = True, actor_step: int = 0, actor_target: Optional[TEN] = None) -> None:
    """one AgentSAC.update_objectives step after the sample; `moments` = (actor_m, actor_v, critic_m, critic_v, alpha_m,
    alpha_v); `batch` = (state, action, reward, undone, unmask, next_state); objs_out: float32[2] on the device.
    `cum_reward` (B,) + `lambda_fit_cum_r`: the critic's fit-the-batch's-mean-return term (AgentSAC.py:66-68)."""
    state, action, reward, undone, unmask, next_state = batch
    B = state.shape[0]
    ws = _workspace(state.device, spec.workspace_bytes(B))
    n_next, n_cur = (None, None) if noises is None else noises
    f32 = th.float32
    if spec.actor_variant or not update_actor or actor_step or actor_target is not None:
        # AgentModSAC's step (include/erl_hip.h, ErlSacOptions): ActorFixSAC's head, the actor skipped by the two-time-scale rule, the
        # actor optimiser's own step count, the actor target's soft update
        opt = _SacOptions(spec.actor_variant, int(bool(update_actor)), int(actor_step), 0, ptr(actor_target, f32))
        check(lib().erl_sac_update_opt_f32(ptr(actor, f32), ptr(critic, f32), ptr(target, f32), ptr(alpha_log, f32), *[ptr(m, f32) for m in moments],
                                           spec.S, spec.A, spec._c, len(spec.hidden), spec.E, ptr(state, f32), ptr(action, f32),
                                           ptr(reward, f32), ptr(undone, f32), ptr(unmask, f32), ptr(next_state, f32), ptr(is_weight),
                                           ptr(td_error_out), ptr(cum_reward), float(lambda_fit_cum_r), B, ptr(n_next), ptr(n_cur),
                                           seed & (2 ** 64 - 1), counter & (2 ** 64 - 1), gamma, target_entropy, tau, lr, betas[0], betas[1], eps,
                                           max_norm, step, ptr(objs_out, f32), ptr(ws), ws.numel(), ctypes.byref(opt), stream_ptr()),
              "erl_sac_update_opt_f32")
        return
    check(lib().erl_sac_update_f32(ptr(actor, f32), ptr(critic, f32), ptr(target, f32), ptr(alpha_log, f32), *[ptr(m, f32) for m in moments],
                                   spec.S, spec.A, spec._c, len(spec.hidden), spec.E, ptr(state, f32), ptr(action, f32),
                                   ptr(reward, f32), ptr(undone, f32), ptr(unmask, f32), ptr(next_state, f32), ptr(is_weight), ptr(td_error_out),
                                   ptr(cum_reward), float(lambda_fit_cum_r), B, ptr(n_next),
                                   ptr(n_cur), seed & (2 ** 64 - 1), counter & (2 ** 64 - 1), gamma, target_entropy, tau, lr, betas[0],
                                   betas[1], eps, max_norm, step, ptr(objs_out, f32), ptr(ws), ws.numel(), stream_ptr()),
          "erl_sac_update_f32")


class _SacOptions(ctypes.Structure):        # include/erl_hip.h ErlSacOptions
    _fields_ = [("actor_variant", ctypes.c_int32), ("update_actor", ctypes.c_int32), ("actor_step", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("actor_target_params", ctypes.c_void_p)]


class _RingSample(ctypes.Structure):        # include/erl_hip.h ErlRingSample
    _fields_ = [("buf_states", ctypes.c_void_p), ("buf_actions", ctypes.c_void_p), ("buf_rewards", ctypes.c_void_p),
                ("buf_undones", ctypes.c_void_p), ("buf_unmasks", ctypes.c_void_p), ("max_size", ctypes.c_int64), ("num_seqs", ctypes.c_int64),
                ("ids", ctypes.c_void_p), ("sample_len", ctypes.c_int64), ("out_ids0", ctypes.c_void_p), ("out_ids1", ctypes.c_void_p),
                ("row_floats", ctypes.c_int64)]


def sac_update_from_ring(spec: SacSpec, actor: TEN, critic: TEN, target: TEN, alpha_log: TEN, moments: Sequence[TEN], ring: Sequence[TEN],
                         ids: TEN, sample_len: int, stage: ReplayStage, step: int, *, gamma: float, target_entropy: float, tau: float, lr: float,
                         max_norm: float, objs_out: TEN, noises: Optional[Tuple[TEN, TEN]] = None, seed: int = 0, counter: int = 0,
                         betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """ReplayBuffer.sample(ids) + one AgentSAC.update_objectives step from ONE call (erl_sac_update_ring_f32): `ring` = the buffer's
    ReplayRing (interleaved block) or its five planar tensors (states, actions, rewards, undones, unmasks), `stage` receives the batch (stage.out / stage.ids: what replay_sample would have left)."""
    B = ids.numel()
    f32 = th.float32
    if isinstance(ring, ReplayRing):        # the interleaved block: one base pointer + its row width
        assert stage.B == B and not stage.discrete and (ring.S, ring.A) == (spec.S, spec.A)
        dev = ring.block.device
        rs = _RingSample(ptr(ring.block, f32), None, None, None, None, ring.max_size, ring.num_seqs, ptr(ids, th.int64), int(sample_len),
                         stage.p_ids0, stage.p_ids1, ring.row_floats)
    else:
        b_states, b_actions, b_rewards, b_undones, b_unmasks = ring
        max_size, num_seqs, S = b_states.shape
        assert stage.B == B and not stage.discrete and b_actions.dtype == th.float32
        dev = b_states.device
        rs = _RingSample(ptr(b_states, f32), ptr(b_actions, f32), ptr(b_rewards, f32), ptr(b_undones, f32), ptr(b_unmasks, f32), max_size, num_seqs,
                         ptr(ids, th.int64), int(sample_len), stage.p_ids0, stage.p_ids1, 0)
    ws = _workspace(dev, spec.workspace_bytes(B))
    n_next, n_cur = (None, None) if noises is None else noises
    check(lib().erl_sac_update_ring_f32(ptr(actor, f32), ptr(critic, f32), ptr(target, f32), ptr(alpha_log, f32), *[ptr(m, f32) for m in moments],
                                        spec.S, spec.A, spec._c, len(spec.hidden), spec.E, ctypes.addressof(rs), stage.p_state, stage.p_action,
                                        stage.p_reward, stage.p_undone, stage.p_unmask, stage.p_next, B, ptr(n_next), ptr(n_cur),
                                        seed & (2 ** 64 - 1), counter & (2 ** 64 - 1), gamma, target_entropy, tau, lr, betas[0], betas[1], eps,
                                        max_norm, step, ptr(objs_out, f32), ptr(ws), ws.numel(), stream_ptr()),
          "erl_sac_update_ring_f32")


def sac_update_ring_loop(spec: SacSpec, actor: TEN, critic: TEN, target: TEN, alpha_log: TEN, moments: Sequence[TEN], ring, ids_all: TEN,
                         sample_len: int, stage: ReplayStage, step0: int, *, gamma: float, target_entropy: float, tau: float, lr: float,
                         max_norm: float, objs_all: TEN, seed: int = 0, counter0: int = 0, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """`ids_all.shape[0]` steps of sac_update_from_ring from ONE C call (erl_sac_update_ring_loop_f32): step t uses ids_all[t], optimiser
    step step0 + t, noise counter counter0 + t, and writes objs_all[t]; `stage` ends up holding the last step's batch and ids0 / ids1."""
    T, B = ids_all.shape
    f32 = th.float32
    assert ids_all.is_contiguous() and objs_all.shape == (T, 2) and objs_all.is_contiguous() and stage.B == B and not stage.discrete
    if isinstance(ring, ReplayRing):
        assert (ring.S, ring.A) == (spec.S, spec.A)
        dev = ring.block.device
        rs = _RingSample(ptr(ring.block, f32), None, None, None, None, ring.max_size, ring.num_seqs, None, int(sample_len), stage.p_ids0,
                         stage.p_ids1, ring.row_floats)
    else:
        b_states, b_actions, b_rewards, b_undones, b_unmasks = ring
        max_size, num_seqs, S = b_states.shape
        dev = b_states.device
        rs = _RingSample(ptr(b_states, f32), ptr(b_actions, f32), ptr(b_rewards, f32), ptr(b_undones, f32), ptr(b_unmasks, f32), max_size, num_seqs,
                         None, int(sample_len), stage.p_ids0, stage.p_ids1, 0)
    ws = _workspace(dev, spec.workspace_bytes(B))
    check(lib().erl_sac_update_ring_loop_f32(ptr(actor, f32), ptr(critic, f32), ptr(target, f32), ptr(alpha_log, f32), *[ptr(m, f32) for m in moments],
                                             spec.S, spec.A, spec._c, len(spec.hidden), spec.E, ctypes.addressof(rs), ptr(ids_all, th.int64), T,
                                             stage.p_state, stage.p_action, stage.p_reward, stage.p_undone, stage.p_unmask, stage.p_next, B,
                                             seed & (2 ** 64 - 1), counter0 & (2 ** 64 - 1), gamma, target_entropy, tau, lr, betas[0], betas[1], eps,
                                             max_norm, int(step0), ptr(objs_all, f32), ptr(ws), ws.numel(), stream_ptr()),
          "erl_sac_update_ring_loop_f32")


def sac_explore_action(spec: SacSpec, actor: TEN, state: TEN, *, noise: Optional[TEN] = None, seed: int = 0, counter: int = 0,
                       out: Optional[TEN] = None, out_state: Optional[TEN] = None) -> TEN:
    """`out` (N, A): the action's destination (e.g. the rollout's row); `out_state` (N, S): a copy of `state` from the same launch."""
    N = state.shape[0]
    out = th.empty((N, spec.A), dtype=th.float32, device=state.device) if out is None else out
    ws = _workspace(state.device, spec.workspace_bytes(N))
    if spec.actor_variant:
        check(lib().erl_sac_explore_action_opt_f32(ptr(actor, th.float32), spec.S, spec.A, spec._c, len(spec.hidden), ptr(state, th.float32), N,
                                                   ptr(noise), seed & (2 ** 64 - 1), counter & (2 ** 64 - 1), ptr(out, th.float32),
                                                   ptr(out_state, th.float32) if out_state is not None else None, ptr(ws), ws.numel(),
                                                   spec.actor_variant, stream_ptr()),
              "erl_sac_explore_action_opt_f32")
        return out
    check(lib().erl_sac_explore_action_f32(ptr(actor, th.float32), spec.S, spec.A, spec._c, len(spec.hidden), ptr(state, th.float32), N,
                                           ptr(noise), seed & (2 ** 64 - 1), counter & (2 ** 64 - 1), ptr(out, th.float32),
                                           ptr(out_state, th.float32) if out_state is not None else None, ptr(ws), ws.numel(), stream_ptr()),
          "erl_sac_explore_action_f32")
    return out


# ------------------------------------------------------------------------------------------------
# environments
# ------------------------------------------------------------------------------------------------
def synenv_step(state: TEN, action: TEN, Ws: TEN, Wa: TEN, step_count: TEN, episode: TEN, reward: TEN, terminal: TEN,
                truncate: TEN, max_step: int, seed: int) -> None:
    N, S = state.shape
    A = action.shape[1]
    check(lib().erl_synenv_step_f32(ptr(state, th.float32), ptr(action, th.float32), ptr(Ws, th.float32), ptr(Wa, th.float32),
                                    ptr(step_count, th.int32), ptr(episode, th.int32), ptr(reward, th.float32),
                                    flag_ptr(terminal), flag_ptr(truncate), N, S, A, max_step, seed & (2 ** 64 - 1),
                                    stream_ptr()),
          "erl_synenv_step_f32")


def pendulum_step(phys: TEN, obs: TEN, action: TEN, step_count: TEN, episode: TEN, reward: TEN, terminal: TEN, truncate: TEN,
                  max_step: int, seed: int) -> None:
    N = phys.shape[0]
    check(lib().erl_pendulum_step_f32(ptr(phys, th.float32), ptr(obs, th.float32), ptr(action, th.float32),
                                      ptr(step_count, th.int32), ptr(episode, th.int32), ptr(reward, th.float32),
                                      flag_ptr(terminal), flag_ptr(truncate), N, max_step, seed & (2 ** 64 - 1), stream_ptr()),
          "erl_pendulum_step_f32")
