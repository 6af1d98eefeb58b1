"""TEST INFRASTRUCTURE (oracle) -- prioritised experience replay, restated on numpy.

What it follows: elegantrl/train/replay_buffer.py:136-179 (`sample_for_per`, `td_error_update_for_per`) and :226-299 (`SumTree`):
one sum tree per sequence (env) over the ring's time rows, proportional prioritisation with stratified draws
`values = (arange(n) + rand(n)) * total / n` (:287), new rows enter with the maximum priority 10 (:111), importance weights
`(leaf_value / min_leaf)^(-beta)` (:296-297), priorities `clamp(td_error, 1e-8, 10)^alpha` (:168).

PARITY UNPINNED against the reference for this row, on purpose: the reference's SumTree cannot run -- its descent and its
update both loop `depth - 2` levels (:240-283), so `get_leaf_id_and_value` stops above the leaves and `important_sampling`'s own
`assert 0 <= indices.min()` (:294) fires (executed on CPU for buf_len 8, 1000, 1024, 4096: all four assert).  This file is the
CORRECTED restatement the HIP kernels (csrc/per.hip) are bit-exact against, with these deviations, each marked below:
  D1  the descent and the update walk the FULL depth (implicit heap over L = next power of two >= max_size leaves);
  D2  every tree draws `batch_size // num_seqs` samples (the reference asks each tree for `batch_size` and then slices);
  D3  `is_indices = ids1 * cur_size + ids0`, so that the reference's own decode lines (:155-156) recover (ids0, ids1) -- the
      reference encodes `leaf + sub_batch_size * env_i`, which its decode does not invert;
  D4  the last filled row position (cur_size - 1) has no successor row for `states[ids0 + 1]` (:167): a draw that lands on it is
      moved to cur_size - 2 (the uniform `sample` excludes that position through sample_len = cur_size - 1, :121);
  D5  unwritten leaves carry priority 0 in the sum tree and +inf in the min tree (the reference's `tree[beg:end].min()` slices
      the written leaves; a min tree gives the same value without an O(n) pass per sample);
  D6  once the ring is full the newest row (write cursor p - 1) is followed in memory by the OLDEST row, so `states[ids0 + 1]`
      would pair it with unrelated data -- and new rows enter at the maximum priority, so a prioritised draw would pick that
      invalid pair far more often than the uniform sampler does (which has the same seam): a draw that lands
      on it is moved to p - 2 (row 1 when the newest row is row 0);
  D7  duplicate (row, sequence) pairs in one td-error update (stratified draws and the moves of D4 / D6 can repeat a
      transition): the LAST one in the list sets the priority (made explicit below; the device resolves it the same way).
"""
from __future__ import annotations

import numpy as np

F = np.float32


class PerTrees:
    def __init__(self, max_size: int, num_seqs: int, per_alpha: float = 0.6, per_beta: float = 0.4):
        self.max_size, self.num_seqs = max_size, num_seqs
        self.L = 1 << max(1, (max_size - 1).bit_length())          # D1: leaves of the implicit heap
        self.sum = np.zeros((num_seqs, 2 * self.L), F)             # node 1 = root, leaves at L + row
        self.min = np.full((num_seqs, 2 * self.L), np.inf, F)      # D5
        self.per_alpha, self.per_beta = per_alpha, per_beta

    # ---- update: SumTree.update_ids (:252-262), full depth (D1) ------------------------------------------------------
    def set(self, rows: np.ndarray, seqs: np.ndarray, prob: np.ndarray) -> None:
        """leaf (seqs[i], rows[i]) <- prob[i]; parents recomputed level by level as left + right in fp32."""
        prob = np.broadcast_to(np.asarray(prob, F), rows.shape)
        key = np.asarray(seqs, np.int64) * self.L + np.asarray(rows, np.int64)       # D7: keep the last occurrence of every leaf
        _, first_rev = np.unique(key[::-1], return_index=True)
        keep = np.sort(len(key) - 1 - first_rev)
        rows, seqs, prob = np.asarray(rows)[keep], np.asarray(seqs)[keep], prob[keep]
        node = rows.astype(np.int64) + self.L
        self.sum[seqs, node] = prob
        self.min[seqs, node] = prob
        while True:
            node = node >> 1
            if node[0] < 1:
                break
            self.sum[seqs, node] = self.sum[seqs, 2 * node] + self.sum[seqs, 2 * node + 1]
            self.min[seqs, node] = np.minimum(self.min[seqs, 2 * node], self.min[seqs, 2 * node + 1])

    def add_rows(self, start: int, add: int) -> None:
        """ReplayBuffer.update's PER part (:107-115): rows [start, start + add) mod max_size of EVERY sequence get priority 10."""
        rows = (start + np.arange(add)) % self.max_size
        r, q = np.meshgrid(rows, np.arange(self.num_seqs), indexing="ij")
        self.set(r.reshape(-1), q.reshape(-1), F(10.0))

    def td_error_update(self, ids0: np.ndarray, ids1: np.ndarray, td_error: np.ndarray) -> None:
        prob = np.power(np.clip(td_error.astype(F), F(1e-8), F(10.0)), F(self.per_alpha)).astype(F)          # :168
        self.set(ids0, ids1, prob)

    # ---- sample: SumTree.important_sampling (:285-298) + get_leaf_id_and_value (:264-283), corrected ---------------------
    def sample(self, uniform: np.ndarray, cur_size: int, cursor: int = -1):
        """uniform: (num_seqs, n) in [0, 1).  Returns (ids0, ids1, weights), each (num_seqs * n,), sequence-major (:145-151).
        `cursor`: the ring's write position p when the ring is full (D6), else -1."""
        newest = (cursor + self.max_size - 1) % self.max_size if (cursor >= 0 and cur_size == self.max_size and self.max_size >= 3) else -1
        Q, n = uniform.shape
        ids0 = np.empty((Q, n), np.int64)
        w = np.empty((Q, n), F)
        for q in range(Q):
            total = self.sum[q, 1]
            values = ((np.arange(n, dtype=F) + uniform[q].astype(F)) * (total / F(n))).astype(F)                 # :287
            for j in range(n):
                v, node = values[j], 1
                while node < self.L:                                                                          # D1
                    left = self.sum[q, 2 * node]
                    if v <= left:
                        node = 2 * node
                    else:
                        v = F(v - left)
                        node = 2 * node + 1
                row = node - self.L
                row = min(row, cur_size - 2)          # D4 (and fp32 round-off at the right edge cannot step into unwritten leaves)
                if row == newest:
                    row = newest - 1 if newest >= 1 else 1                                                       # D6
                ids0[q, j] = row
                w[q, j] = np.power(self.sum[q, self.L + row] / self.min[q, 1], F(-self.per_beta))             # :296-297
        ids1 = np.repeat(np.arange(Q, dtype=np.int64), n)
        return ids0.reshape(-1), ids1, w.reshape(-1)
