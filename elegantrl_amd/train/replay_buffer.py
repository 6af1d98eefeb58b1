"""Off-policy `ReplayBuffer` on the HIP ring-write / sample kernels (K8 / K9).

Same constructor, attributes and cursor behaviour as elegantrl/train/replay_buffer.py:11-134; the
cursor arithmetic (p, cur_size, if_full, add_size) is host-side integer code and reproduces the
reference bit for bit (including landing exactly on max_size, Appendix A12 of SURVEY.md).  The tensor
traffic goes through erl_replay_write_rows_f32 / erl_replay_sample_rows_f32 on ONE interleaved block (round 6: a transition and
its next state are consecutive bytes; the reference's five tensors are strided views of it), or erl_replay_write_f32 /
erl_replay_sample_f32 on five planar tensors (`args.replay_interleaved = False`; uint8 action rings of discrete agents:
erl_replay_*_discrete_f32).  Prioritised replay (`if_use_per=True`) keeps its per-sequence
sum / min trees on the device (csrc/per.hip, erl_per_*).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch as th

from .. import _hip
from .config import Config

TEN = th.Tensor


class ReplayBuffer:
    def __init__(self, max_size: int, state_dim: int, action_dim: int, gpu_id: int = 0, num_seqs: int = 1,
                 if_use_per: bool = False, if_discrete: bool = False, args: Optional[Config] = None):
        assert (action_dim < 256) or (not if_discrete)       # replay_buffer.py:50: a discrete action must fit a byte
        self.p = 0                 # write cursor (time row)
        self.if_full = False
        self.cur_size = 0
        self.add_size = 0
        self.max_size = int(max_size)
        self.num_seqs = int(num_seqs)
        self.device = th.device(f"cuda:{gpu_id}" if (th.cuda.is_available() and gpu_id >= 0) else "cpu")
        f32 = dict(dtype=th.float32, device=self.device)
        self.if_discrete = bool(if_discrete)
        self._stage = None
        # Continuous-action buffers on a HIP device keep ONE interleaved block (ops.ReplayRing, round 6): a row is [state | action |
        # reward | undone | unmask], sequence-major, so a sampled transition and its next state are consecutive bytes; the reference's
        # five attributes (replay_buffer.py:40-58) are strided views of it -- same shapes, dtypes, indexing and assignment behaviour, not
        # contiguous.  `args.replay_interleaved = False` (or a discrete-action ring, or no device) keeps five planar tensors.
        self._ring = None
        if (not if_discrete and self.device.type == "cuda" and getattr(args, "replay_interleaved", True) and self.max_size >= 2):
            from .. import ops
            with th.cuda.device(self.device):
                self._ring = ops.ReplayRing(self.max_size, self.num_seqs, state_dim, action_dim, self.device)
            r = self._ring
            self.states, self.actions, self.rewards, self.undones, self.unmasks = r.states, r.actions, r.rewards, r.undones, r.unmasks
        else:
            self.states = th.empty((self.max_size, self.num_seqs, state_dim), **f32)
            self.actions = (th.empty((self.max_size, self.num_seqs), dtype=th.uint8, device=self.device) if if_discrete
                            else th.empty((self.max_size, self.num_seqs, action_dim), **f32))          # replay_buffer.py:52-54
            self.rewards = th.empty((self.max_size, self.num_seqs), **f32)
            self.undones = th.empty((self.max_size, self.num_seqs), **f32)   # float flags, as in the reference
            self.unmasks = th.empty((self.max_size, self.num_seqs), **f32)
        self.cum_rewards = th.empty((self.max_size, self.num_seqs), **f32)
        self.ids0 = th.tensor((), dtype=th.long, device=self.device)
        self.ids1 = th.tensor((), dtype=th.long, device=self.device)
        # prioritised replay (replay_buffer.py:64-76): the reference keeps one CPU SumTree per sequence; here the trees of all
        # sequences live on the device (csrc/per.hip) behind `self.sum_trees` (an ops.PerTrees, not a list)
        self.if_use_per = bool(if_use_per)
        self.sum_trees = None
        self.per_alpha = getattr(args, "per_alpha", 0.6) if if_use_per else None     # alpha = (Uniform:0, Greedy:1)
        self.per_beta = getattr(args, "per_beta", 0.4) if if_use_per else None
        if if_use_per:
            from .. import ops
            self.sum_trees = ops.PerTrees(self.max_size, self.num_seqs, self.device)

    # ---- cursor arithmetic: replay_buffer.py:84-118 -------------------------------------------
    def _advance(self, add_size: int) -> int:
        """update p / if_full / cur_size for `add_size` new time rows; returns the row the write starts at."""
        assert 0 < add_size <= self.max_size, f"add_size={add_size} must be in (0, max_size={self.max_size}]"
        self.add_size = add_size
        start = self.p
        p = self.p + add_size
        if p > self.max_size:      # strictly greater: landing exactly on max_size is not "full" yet
            self.if_full = True
            p -= self.max_size
        self.p = p
        self.cur_size = self.max_size if self.if_full else self.p
        return start

    @_hip.on_device
    def update(self, items: Tuple[TEN, ...]):
        from .. import ops
        states, actions, rewards, undones, unmasks = items
        start = self._advance(rewards.shape[0])
        items = (states.contiguous(), actions.contiguous(), rewards.contiguous(), undones.contiguous(), unmasks.contiguous())
        if self._ring is not None:
            self._ring.write(items, start)
        else:
            ops.replay_write(self.states, self.actions, self.rewards, self.undones, self.unmasks, items, start)
        if self.if_use_per:                          # new rows enter with the maximum priority (replay_buffer.py:107-115)
            self.sum_trees.add_rows(start, self.add_size, 10.0)

    @_hip.on_device
    def sample(self, batch_size: int, ids: Optional[TEN] = None, reuse: bool = False) -> Tuple[TEN, TEN, TEN, TEN, TEN, TEN]:
        """(state, action, reward, undone, unmask, next_state) for ids drawn like the reference
        (th.randint(sample_len * num_seqs, (batch_size,))); `ids` can be injected for tests.
        reuse=True writes into a buffer-owned staging block that the NEXT reuse=True call overwrites (no allocator call, no
        view construction per sample) -- for loops that consume a batch before drawing the next, like the off-policy
        update_net; the default returns fresh tensors like the reference."""
        from .. import ops
        sample_len = self.cur_size - 1
        if ids is None:
            ids = th.randint(sample_len * self.num_seqs, size=(batch_size,), requires_grad=False, device=self.device)
        stage = None
        if reuse:
            B = ids.numel()
            if self._stage is None or self._stage.B != B:
                self._stage = ops.ReplayStage(B, self.states.shape[2], 1 if self.if_discrete else self.actions.shape[2],
                                              self.if_discrete, self.device)
            stage = self._stage
        if self._ring is not None:
            out, (self.ids0, self.ids1) = self._ring.sample(ids, sample_len, stage=stage)
        else:
            out, (self.ids0, self.ids1) = ops.replay_sample(self.states, self.actions, self.rewards, self.undones, self.unmasks,
                                                            ids, sample_len, stage=stage)
        return out

    def ring_for_fused_sample(self, batch_size: int):
        """((states, actions, rewards, undones, unmasks), sample_len, stage) for a consumer that does `sample` itself on given ids
        (ops.sac_update_from_ring: the gather inside the SAC step's first launch), or None where that does not apply (discrete ring, PER);
        `stage` is the block `sample(..., reuse=True)` writes to."""
        from .. import ops
        if self.if_discrete or self.if_use_per or self.cur_size < 2:
            return None
        if self._stage is None or self._stage.B != batch_size:
            self._stage = ops.ReplayStage(batch_size, self.states.shape[2], self.actions.shape[2], False, self.device)
        arrays = self._ring if self._ring is not None else (self.states, self.actions, self.rewards, self.undones, self.unmasks)
        return arrays, self.cur_size - 1, self._stage

    @_hip.on_device
    def sample_for_per(self, batch_size: int, uniform: Optional[TEN] = None):
        """Prioritised sample (replay_buffer.py:136-165): (state, action, reward, undone, unmask, next_state, is_weights,
        is_indices); `batch_size // num_seqs` stratified proportional draws per sequence.  The reference's SumTree does not run
        (DESIGN.md section 8); this follows the corrected restatement oracle/per_numpy.py (its header lists the deviations:
        full-depth trees, per-sequence batch share, is_indices = ids1 * cur_size + ids0, no draw on the last filled row).
        `uniform` (num_seqs, batch_size // num_seqs) in [0, 1) injects the random numbers (tests)."""
        from .. import ops
        assert self.if_use_per, "ReplayBuffer was built with if_use_per=False"
        assert batch_size % self.num_seqs == 0                                        # replay_buffer.py:144
        assert self.cur_size >= 2
        sub = batch_size // self.num_seqs
        if uniform is None:
            uniform = th.rand((self.num_seqs, sub), dtype=th.float32, device=self.device)
        # full ring: the newest row (p - 1) is followed in memory by the oldest one -- never drawn (oracle/per_numpy.py D6)
        is_indices, is_weights = self.sum_trees.sample(uniform, self.cur_size, self.per_beta, cursor=self.p if self.if_full else -1)
        if self._ring is not None:                                                      # ids0 = fmod, ids1 = div (:155-156)
            out, (self.ids0, self.ids1) = self._ring.sample(is_indices, self.cur_size)
        else:
            out, (self.ids0, self.ids1) = ops.replay_sample(self.states, self.actions, self.rewards, self.undones, self.unmasks,
                                                            is_indices, self.cur_size)
        return (*out, is_weights, is_indices)

    @_hip.on_device
    def td_error_update_for_per(self, is_indices: TEN, td_error: TEN):
        """priorities <- clamp(td_error, 1e-8, 10)^per_alpha for the sampled transitions (replay_buffer.py:167-179)."""
        assert self.if_use_per, "ReplayBuffer was built with if_use_per=False"
        ids0 = th.fmod(is_indices, self.cur_size)
        ids1 = th.div(is_indices, self.cur_size, rounding_mode="floor")
        self.sum_trees.update(ids0, ids1, td_error.detach().to(th.float32).reshape(-1).contiguous(), self.per_alpha)

    # ---- persistence: replay_buffer.py:181-211 (same file names and unrolled order) --------------
    def save_or_load_history(self, cwd: str, if_save: bool):
        named = ((self.states, "states"), (self.actions, "actions"), (self.rewards, "rewards"), (self.undones, "undones"))
        if if_save:
            for item, name in named:
                if self.cur_size == self.p:
                    buf_item = item[:self.cur_size].contiguous()     # (a view of the interleaved block would drag the whole block into the file)
                else:
                    buf_item = th.vstack((item[self.p:self.cur_size], item[0:self.p]))
                path = f"{cwd}/replay_buffer_{name}.pth"
                print(f"| buffer.save_or_load_history(): Save {path}", flush=True)
                th.save(buf_item, path)
        elif all(os.path.isfile(f"{cwd}/replay_buffer_{name}.pth") for _, name in named):
            sizes = []
            for item, name in named:
                path = f"{cwd}/replay_buffer_{name}.pth"
                print(f"| buffer.save_or_load_history(): Load {path}", flush=True)
                buf_item = th.load(path, map_location=self.device)
                item[:buf_item.shape[0]] = buf_item
                sizes.append(buf_item.shape[0])
            assert all(s == sizes[0] for s in sizes)
            self.cur_size = self.p = sizes[0]
            self.if_full = self.cur_size == self.max_size
            if self.if_use_per:        # the priorities are not part of the reference's file set: a resumed run starts every loaded
                from .. import ops    # transition at the maximum priority, like freshly written rows (replay_buffer.py:107-115)
                self.sum_trees = ops.PerTrees(self.max_size, self.num_seqs, self.device)
                self.sum_trees.add_rows(0, self.cur_size, 10.0)

    def update_cum_rewards(self, get_cumulative_rewards):
        if self.p >= self.add_size:
            p1, p0 = self.p, self.p - self.add_size
        else:
            p1 = self.max_size
            p0 = p1 - self.add_size
        self.cum_rewards[p0:p1, :] = get_cumulative_rewards(rewards=self.rewards[p0:p1, :], undones=self.undones[p0:p1, :])
