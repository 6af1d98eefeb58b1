/* erl_hip.h -- C ABI of liberl_hip.so: the MI355X (gfx950) hot path of an ElegantRL-compatible
 * vectorised actor-learner.
 *
 * The reference (AI4Finance-Foundation/ElegantRL) is 100% Python/PyTorch and has no FFI of its own; the
 * seam this library sits behind is the Python class protocol driven by elegantrl/train/run.py
 * (SURVEY.md section 8b).  Each entry point below names the reference lines whose arithmetic it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer would add under elegantrl/agents and
 * elegantrl/train.
 *
 * Conventions
 *  - plain C types only: raw device pointers, sizes, scalars, an opaque hipStream_t passed as void*.
 *  - every tensor is caller-allocated, caller-owned device memory (torch tensors in practice).  The library
 *    itself owns only: a last-error string, an RCCL communicator behind erl_comm_* handles, and one 8 MiB look-back table
 *    per (device, stream) that launches the single-pass GAE scan (allocated on first use, at most 16; see ERL_GAE_ALGO_LOOKBACK).
 *  - all work is enqueued on `stream`; nothing synchronises the host -- with ONE exception per device and process: the first
 *    minibatch-kernel launch that fills the chip (>= 256 workgroups; erl_ppo_step_f32 / erl_ppo_update*_f32 / erl_mlpn_ppo_step_f32)
 *    measures the kernel under two workgroup maps (30 extra launches between HIP events, ~1 ms, hipEventSynchronize) and, on a device
 *    that chose map 2, resolves the kernel's code range once (one small hipMalloc / hipStreamSynchronize / hipFree).  Results are
 *    bit-identical under every map.  ERL_K6_WG_MAP=0|1|2 or ERL_K6_NO_TUNE=1 in the environment skips the measurement; agents make
 *    that first launch in their warm-up, not in a timed or captured region (a capturing stream defers it).
 *  - layout at the seam is the reference's: time-major (H, N, .) row-major contiguous, fp32 values,
 *    1-byte flags (torch.bool), int64 indices.
 *  - return value: 0 = ok; ERL_EINVAL (-1) = bad argument; -(1000 + hipError_t) = HIP runtime error.
 *    erl_last_error_string() describes the most recent failure on the calling thread.
 */
#ifndef ERL_HIP_H
#define ERL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ERL_ABI_VERSION 19
#define ERL_API __attribute__((visibility("default")))
#define ERL_OK 0
#define ERL_EINVAL (-1)

/* limits of the fused MLP kernels (2 hidden layers, as Config.net_dims = [128, 128]) */
#define ERL_MAX_STATE_DIM 128
#define ERL_MAX_HIDDEN 128
#define ERL_MAX_ACTION_DIM 16
/* limits of the generic-shape path (erl_mlpn_*): hidden layers / layer width */
#define ERL_MAX_LAYERS 6
#define ERL_MAXN_WIDTH 4096

ERL_API int erl_abi_version(void);
ERL_API const char *erl_last_error_string(void);
/* number of compute units / max LDS per workgroup of the current device (needs a GPU). */
ERL_API int erl_device_info(int *num_cu, int *lds_bytes_per_block);

/* ---------------------------------------------------------------------------------------------
 * K3  GAE / lambda-return backward scan.
 * Replaces AgentPPO.get_advantages (elegantrl/agents/AgentPPO.py:207-232; the north star's
 * `get_reward_sum_gae`) plus `reward_sums = advantages + values` (:146).
 *   trunc = !unmask: r += V(s_t) (= values at that element), undone = 0            (:211-214)
 *   m = undone*gamma; v-trace branch (:223-227) or the alternative branch (:228-231)
 * rewards/values/adv/ret: (H, N) f32; undones/unmasks: (H, N) u8; next_value: (N,) = cri(last_state).
 * `ret` may be NULL.  With ERL_GAE_MUTATE the truncation fix-up is written back into
 * rewards/undones exactly like the reference does to its caller's tensors.
 * With ERL_GAE_STATS the kernel also leaves the raw sums needed by K4 in stats[0..4]
 * (see erl_adv_stats_f32).  `workspace` must hold erl_gae_workspace_bytes(H, N) bytes.
 * Algorithmic HBM traffic: 18 B per (t, n) element (14 B without ret).
 * ------------------------------------------------------------------------------------------- */
#define ERL_GAE_VTRACE 0x1      /* if_use_v_trace=True branch (the reference default) */
#define ERL_GAE_MUTATE 0x2      /* write the truncation fix-up back into rewards / undones */
#define ERL_GAE_STATS 0x4       /* also produce adv statistics (stats must be non-NULL) */
#define ERL_GAE_ALGO_AUTO 0x00
#define ERL_GAE_ALGO_EXACT 0x10    /* one thread per env, reference op order, bit-exact vs the oracle */
#define ERL_GAE_ALGO_CHUNKED 0x20  /* time-parallel two-pass affine scan (within 1e-5) */
#define ERL_GAE_ALGO_LOOKBACK 0x30 /* time-parallel single-pass scan (within 1e-5).  64 <= H <= 256: one workgroup per 32 envs
                                      holds the whole horizon, its 32 time chunks meet in LDS (no table, no waits); otherwise
                                      decoupled look-back over slabs, whose granule table lives in a library-owned buffer
                                      (nonce-tagged, never cleared) when it fits, else in `workspace` behind a memset */
#define ERL_GAE_ALGO_MASK 0xF0
ERL_API int64_t erl_gae_workspace_bytes(int64_t H, int64_t N);
ERL_API int erl_gae_scan_f32(float *rewards, uint8_t *undones, const uint8_t *unmasks, const float *values,
                     const float *next_value, float *adv, float *ret, int64_t H, int64_t N, float gamma,
                     float lam, int flags, double *stats, void *workspace, int64_t workspace_bytes,
                     void *stream);

/* Device-side faults that an asynchronous launch cannot return are counted in a pinned, host-mapped block, one counter
 * per source: (0) a look-back wait of ERL_GAE_ALGO_LOOKBACK that timed out because a predecessor slab never published --
 * the affected advantages are NaN; (1) a wait of the peer-to-peer gradient exchange for a peer's slice (a rank missing or
 * stalled) -- those summed gradients are invalid; (2) a clip + Adam grid wait that gave up (device shared with another
 * process) -- those parameter updates were skipped.  Call after the stream has been synchronised: returns the number of
 * faults since the last reset (0 = none), describes them per source in erl_last_error_string(), and clears the counters
 * when `reset` != 0.  Costs no GPU work. */
ERL_API int erl_async_fault_count(int reset);

/* n-step discounted return for the off-policy agents.  Replaces AgentBase.get_cumulative_rewards' scan
 * (elegantrl/agents/AgentBase.py:226-237):  masks = undones * gamma;
 *   for t = H-1 .. 0:  cum_rewards[t] = next_value = rewards[t] + masks[t] * next_value        (bit-exact op order)
 * rewards / undones / cum_rewards: (H, N) f32 (the replay buffer's float flags, replay_buffer.py:57); next_value: (N,)
 * = cri_target(last_state, act_target(last_state)), computed by the caller. */
ERL_API int erl_cum_rewards_f32(const float *rewards, const float *undones, const float *next_value, float *cum_rewards,
                        int64_t H, int64_t N, float gamma, void *stream);

/* ---------------------------------------------------------------------------------------------
 * K4  advantage normalisation.  Replaces AgentPPO.py:149:
 *       adv = (adv - adv.mean()) / (adv[::4, ::4].std() + 1e-5)      (unbiased std of the subsample)
 * erl_adv_stats_f32 writes raw sums so that data-parallel ranks can all-reduce them first:
 *   stats[0] = sum(adv)  stats[1] = H*N  stats[2] = sum(sub)  stats[3] = sum(sub^2)  stats[4] = count(sub)
 * erl_adv_normalize_f32 turns the sums into mean/std on the device and writes (adv-mean)/(std+1e-5).
 * `out` may alias `adv`.  workspace: erl_gae_workspace_bytes(H, N) bytes.
 * ------------------------------------------------------------------------------------------- */
ERL_API int erl_adv_stats_f32(const float *adv, int64_t H, int64_t N, double *stats, void *workspace,
                      int64_t workspace_bytes, void *stream);
ERL_API int erl_adv_normalize_f32(const float *adv, float *out, int64_t H, int64_t N, const double *stats,
                          void *stream);
/* the fold on its own: n_partials x 3 fp64 partial sums (sum adv | sum over [::4, ::4] | sum of squares over it) -> stats[0..4]
 * (stats[5..7] = 0), in index order.  For the per-workgroup partials the persistent rollouts' epilogue leaves. */
ERL_API int erl_adv_stats_fold_f32(const double *partials, int n_partials, int64_t H, int64_t N, double *stats, void *stream);

/* ---------------------------------------------------------------------------------------------
 * K5  minibatch index decomposition + gather.  Replaces AgentPPO.update_objectives' sampling
 * (AgentPPO.py:178-187):  ids0 = ids % H (time row), ids1 = ids // H (env column)  -- bit-exact --
 * then x[ids0, ids1] for states, actions, unmasks, logprobs, advantages, reward_sums.
 * Any out_* pointer may be NULL to skip that output.
 * ------------------------------------------------------------------------------------------- */
ERL_API int erl_split_ids_i64(const int64_t *ids, int64_t B, int64_t sample_len, int64_t *ids0, int64_t *ids1,
                      void *stream);
ERL_API int erl_ppo_gather_f32(const float *states, const float *actions, const uint8_t *unmasks,
                       const float *logprobs, const float *advantages, const float *reward_sums, int64_t H,
                       int64_t N, int S, int A, const int64_t *ids, int64_t B, float *out_state,
                       float *out_action, uint8_t *out_unmask, float *out_logprob, float *out_advantage,
                       float *out_reward_sum, int64_t *out_ids0, int64_t *out_ids1, void *stream);

/* ---------------------------------------------------------------------------------------------
 * K8  replay ring write.  Replaces ReplayBuffer.update's tensor copies
 * (elegantrl/train/replay_buffer.py:86-105): `add` time rows are appended at row `p`, wrapping at
 * max_size (rows [p, max_size) then [0, p+add-max_size)).  The cursor arithmetic (p, cur_size, if_full)
 * stays on the host in the Python class.  Flags arrive as torch.bool (flag_is_f32 = 0) or float32
 * (flag_is_f32 = 1) and are stored as float32 like the reference's buffers (:57-58).
 * ------------------------------------------------------------------------------------------- */
ERL_API int erl_replay_write_f32(float *buf_states, float *buf_actions, float *buf_rewards, float *buf_undones,
                         float *buf_unmasks, const float *states, const float *actions, const float *rewards,
                         const void *undones, const void *unmasks, int flag_is_f32, int64_t max_size,
                         int64_t num_seqs, int S, int A, int64_t p, int64_t add, void *stream);
/* discrete actions (if_discrete=True, replay_buffer.py:53-54): `actions` (add, num_seqs) int32 as the off-policy rollout
 * produces them (AgentBase.py:146), stored as uint8 in `buf_actions` (max_size, num_seqs) -- the low byte, like torch's
 * assignment of an int32 tensor into a uint8 one */
ERL_API int erl_replay_write_discrete_f32(float *buf_states, uint8_t *buf_actions, float *buf_rewards, float *buf_undones,
                                  float *buf_unmasks, const float *states, const int32_t *actions, const float *rewards,
                                  const void *undones, const void *unmasks, int flag_is_f32, int64_t max_size,
                                  int64_t num_seqs, int S, int64_t p, int64_t add, void *stream);

/* ---------------------------------------------------------------------------------------------
 * K9  replay sample.  Replaces ReplayBuffer.sample (replay_buffer.py:120-134) given the drawn ids:
 *   ids0 = ids % sample_len, ids1 = ids // sample_len (sample_len = cur_size - 1), bit-exact;
 *   returns states/actions/rewards/undones/unmasks[ids0, ids1] and next_state = states[ids0 + 1, ids1].
 * ------------------------------------------------------------------------------------------- */
ERL_API int erl_replay_sample_f32(const float *buf_states, const float *buf_actions, const float *buf_rewards,
                          const float *buf_undones, const float *buf_unmasks, int64_t max_size,
                          int64_t num_seqs, int S, int A, const int64_t *ids, int64_t B, int64_t sample_len,
                          float *out_state, float *out_action, float *out_reward, float *out_undone,
                          float *out_unmask, float *out_next_state, int64_t *out_ids0, int64_t *out_ids1,
                          void *stream);
/* the same for the uint8 action buffer of a discrete-action ring: out_action is (B,) uint8 */
ERL_API int erl_replay_sample_discrete_f32(const float *buf_states, const uint8_t *buf_actions, const float *buf_rewards,
                                   const float *buf_undones, const float *buf_unmasks, int64_t max_size,
                                   int64_t num_seqs, int S, const int64_t *ids, int64_t B, int64_t sample_len,
                                   float *out_state, uint8_t *out_action, float *out_reward, float *out_undone,
                                   float *out_unmask, float *out_next_state, int64_t *out_ids0, int64_t *out_ids1,
                                   void *stream);

/* ---------------------------------------------------------------------------------------------
 * K8 / K9 on the INTERLEAVED ring (ABI 18).  The reference keeps five planar tensors (max_size, num_seqs, .)
 * (elegantrl/train/replay_buffer.py:40-58); a random transition then touches six to seven 128-byte lines for 232 algorithmic bytes.
 * Here the ring is ONE block ring[num_seqs][max_size][RW] fp32, RW = erl_replay_row_floats(S, A) = S + A + 3 rounded up to 4 floats,
 * row = [state (S) | action (A) | reward | undone | unmask | pad], sequence-major so that states[ids0 + 1, ids1] (:133) is the head of the
 * next row: one sample reads RW + S consecutive floats.  Same arguments and results as erl_replay_write_f32 / erl_replay_sample_f32
 * otherwise (indices bit-exact, rows bit-exact copies); the Python class exposes the reference's attributes as strided views of the block.
 * sample_len <= max_size; an id that decodes to ids0 = max_size - 1 (no following row: the reference raises) gets its own row as next state.
 * ------------------------------------------------------------------------------------------- */
ERL_API int64_t erl_replay_row_floats(int S, int A);
ERL_API int erl_replay_write_rows_f32(float *ring, int64_t max_size, int64_t num_seqs, int S, int A, const float *states,
                              const float *actions, const float *rewards, const void *undones, const void *unmasks,
                              int flag_is_f32, int64_t p, int64_t add, void *stream);
ERL_API int erl_replay_sample_rows_f32(const float *ring, int64_t max_size, int64_t num_seqs, int S, int A, const int64_t *ids, int64_t B,
                               int64_t sample_len, float *out_state, float *out_action, float *out_reward, float *out_undone,
                               float *out_unmask, float *out_next_state, int64_t *out_ids0, int64_t *out_ids1, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Prioritised experience replay: the per-sequence SumTree of replay_buffer.py:226-299 as device-resident
 * implicit heaps, one sum tree and one min tree per sequence: (num_seqs, 2 L) fp32, L = next power of two >= max_size, node 1 =
 * root, leaf of time row r at L + r (erl_per_tree_floats floats each).  Parents are recomputed as left + right level by level
 * (deterministic).  The reference's SumTree does not run (its loops stop two levels short and its own assert fires); these entry
 * points implement the corrected restatement oracle/per_numpy.py, whose header lists the deviations.
 *   erl_per_add_rows_f32   ReplayBuffer.update's PER part (:107-115): rows [start, start + add) mod max_size of every sequence <- prob
 *   erl_per_update_f32     td_error_update_for_per (:167-179): leaf (ids1, ids0) <- clamp(td_error, 1e-8, 10)^per_alpha
 *   erl_per_sample_f32     sample_for_per / important_sampling (:136-151, :285-298): n_per_seq stratified draws per sequence from
 *                          uniform (num_seqs, n_per_seq) in [0,1); out_index = ids1 * cur_size + ids0 (decodes by the reference's
 *                          fmod / div), out_weight = (priority / min priority)^(-per_beta); a draw on row cur_size - 1 moves to
 *                          cur_size - 2 (it has no successor row); `cursor` = the ring's write position p when the ring is full
 *                          (< 0 otherwise): a draw on the newest row p - 1, whose successor slot holds the oldest data, moves to
 *                          p - 2.  Duplicate (ids0, ids1) pairs in one erl_per_update_f32 list: the highest list index wins;
 *                          pairs outside the trees are skipped.
 * ------------------------------------------------------------------------------------------- */
ERL_API int64_t erl_per_tree_floats(int64_t max_size, int64_t num_seqs);
ERL_API int erl_per_init_f32(float *sum_tree, float *min_tree, int64_t max_size, int64_t num_seqs, void *stream);
ERL_API int erl_per_add_rows_f32(float *sum_tree, float *min_tree, int64_t max_size, int64_t num_seqs, int64_t start, int64_t add,
                         float prob, void *stream);
ERL_API int erl_per_update_f32(float *sum_tree, float *min_tree, int64_t max_size, int64_t num_seqs, const int64_t *ids0,
                       const int64_t *ids1, const float *td_error, int64_t n, float per_alpha, void *stream);
ERL_API int erl_per_sample_f32(const float *sum_tree, const float *min_tree, int64_t max_size, int64_t num_seqs,
                       const float *uniform, int64_t n_per_seq, int64_t cur_size, int64_t cursor, float per_beta, int64_t *out_index,
                       float *out_weight, void *stream);

/* ---------------------------------------------------------------------------------------------
 * MLP parameter block used by K1/K2/K6/K7: one flat fp32 buffer per network, laid out as
 *   W1[h1][S] b1[h1] W2[h2][h1] b2[h2] W3[out][h2] b3[out] (+ action_std_log[A] for the actor)
 * i.e. nn.Linear.weight/.bias of build_mlp([S, h1, h2, out]) (elegantrl/agents/AgentBase.py:345-360)
 * in order; the Python side makes the nn.Module parameters views of this buffer.  state_avg /
 * state_std are the (S,) normalisation buffers of ActorPPO/CriticPPO (AgentPPO.py:357-361, :432-441).
 * Activation is exact-erf GELU.  Constraints: h1, h2 multiples of 32 and <= ERL_MAX_HIDDEN,
 * S <= ERL_MAX_STATE_DIM, A <= ERL_MAX_ACTION_DIM.
 * One wider shape class is accepted by erl_mlp_param_count, erl_ppo_slab_stride, K6 (erl_ppo_step_f32) and erl_ppo_update_dp_f32 only:
 * h1 == 256, h2 in {64, 128}, S <= 64, A <= 8 -- net_dims (256, 128) of the reference's LunarLander demo (examples/demo_A2C_PPO.py:117).
 * Its minibatch kernel streams W2 through LDS from images it builds from the parameters: a stand-alone erl_ppo_step_f32 call of this
 * shape needs critic_params == actor_params + erl_mlp_param_count(S, h1, h2, A, 1) (one flat block [actor | critic]); rollouts and value
 * pre-passes of the shape go through the erl_mlpn_* entry points.
 * ------------------------------------------------------------------------------------------- */
ERL_API int64_t erl_mlp_param_count(int S, int h1, int h2, int out, int with_std_log);

/* K2  value pre-pass: values[i] = CriticPPO(states[i]) for `rows` states (AgentPPO.py:141-143, :219-220). */
ERL_API int erl_value_forward_f32(const float *critic_params, const float *state_avg, const float *state_std, int S,
                          int h1, int h2, const float *states, int64_t rows, float *values, void *stream);

/* K1  one vectorised rollout step = ActorPPO.get_action + the three buffer stores
 * (AgentPPO.py:113-119, :368-376):  a = mean + exp(std_log)*eps,  logprob = Normal.log_prob(a).sum(1),
 * out_state_row <- state, out_action_row <- a (pre-tanh), out_logprob_row <- logprob,
 * out_action_env <- tanh(a) (convert_action_for_env, :388-390).  eps comes from `noise` (N, A) when
 * non-NULL (tests), otherwise from Philox4x32-10 keyed by (seed, counter, env, action-dim). */
ERL_API int erl_rollout_step_f32(const float *actor_params, const float *state_avg, const float *state_std, int S,
                         int h1, int h2, int A, const float *state, int64_t N, const float *noise,
                         uint64_t seed, uint64_t counter, float *out_state_row, float *out_action_row,
                         float *out_logprob_row, float *out_action_env, void *stream);

/* K1+K2 fused over a whole horizon, for the GPU-resident environments below: ONE launch runs all H steps of
 * AgentPPO._explore_vec_env (elegantrl/agents/AgentPPO.py:87-129) -- get_action, the buffer stores, tanh, env.step, the
 * reward / flag stores, `rewards *= reward_scale`, `undones = ~terminals`, `unmasks = ~truncates` (:126-128) -- and
 * evaluates CriticPPO on every visited state: out_values (H, N) is update_net's value pre-pass (:141-143) and
 * out_next_value (N) the bootstrap cri(last_state) (:219-220); both may be NULL.  A workgroup owns 16 envs for all H
 * steps (weights in registers / LDS for the whole rollout, state tile in LDS).  Step t draws its noise from
 * noise[t] ((H, N, A), tests) or Philox keyed by (seed, counter0 + t, env, action-dim) -- the keys H successive
 * erl_rollout_step_f32 calls would use -- and every step's arithmetic is that of erl_rollout_step_f32 (N <= 16384 form) +
 * erl_synenv_step_f32 / erl_pendulum_step_f32, so the six rollout buffers come out bit-identical to the per-step path.
 * The env's live state / counters are read at entry and written back at exit.  Needs erl_rollout_fused_supported(...):
 * state_dim <= 64 on top of the K1 constraints.
 * Epilogue (round 4; each output may be NULL): out_last_state (N, S) = a copy of the final state for `agent.last_state`
 * (AgentPPO.py:125: a tensor of the agent's own, not the env's live buffer); out_advantages / out_reward_sums (H, N) +
 * gae_stats (8 doubles) = AgentPPO.get_advantages (:207-232), reward_sums (:146) and the raw sums of the advantage
 * normalisation (:149) over the rollout just written -- exactly erl_gae_scan_f32(EXACT, STATS)'s outputs except that the
 * caller's rewards / undones are left as explore_env returns them (the truncation fix-up of get_advantages is applied by
 * erl_ppo_finish_f32).  The advantages are RAW: erl_ppo_update_dp_f32(adv_stats = gae_stats) normalises them at its row
 * loads.  They need out_values, out_next_value and gae_workspace (erl_rollout_gae_workspace_bytes(N) bytes): the sums are
 * left there as erl_rollout_gae_partials(N) per-workgroup fp64 partials (a fold inside the rollout launch costs three
 * dependent memory round trips at its end: measured +22-26 us); erl_adv_stats_fold_f32 -- or erl_ppo_update_dp_f32's
 * adv_partials argument -- folds them into the 8-double block (`gae_stats` is reserved, may be NULL).  gamma / lambda_gae /
 * use_v_trace as erl_gae_scan_f32. */
ERL_API int erl_rollout_fused_supported(int S, int h1, int h2, int A);
ERL_API int64_t erl_rollout_gae_workspace_bytes(int64_t N);
ERL_API int erl_rollout_gae_partials(int64_t N);     /* rows of 3 doubles the epilogue leaves in gae_workspace */
ERL_API int erl_rollout_synenv_f32(const float *actor_params, const float *critic_params, const float *act_avg,
                           const float *act_std, const float *cri_avg, const float *cri_std, int S, int h1, int h2, int A,
                           float *env_state, const float *Ws, const float *Wa, int32_t *step_count, int32_t *episode,
                           int max_step, uint64_t env_seed, int64_t N, int64_t H, const float *noise, uint64_t seed,
                           uint64_t counter0, float reward_scale, float *out_states, float *out_actions,
                           float *out_logprobs, float *out_rewards, uint8_t *out_undones, uint8_t *out_unmasks,
                           float *out_values, float *out_next_value, float *out_last_state, float *out_advantages,
                           float *out_reward_sums, double *gae_stats, double *gae_workspace, int64_t gae_workspace_bytes,
                           float gamma, float lambda_gae, int use_v_trace, void *stream);
ERL_API int erl_rollout_pendulum_f32(const float *actor_params, const float *critic_params, const float *act_avg,
                             const float *act_std, const float *cri_avg, const float *cri_std, int h1, int h2, float *phys,
                             float *obs, int32_t *step_count, int32_t *episode, int max_step, uint64_t env_seed, int64_t N,
                             int64_t H, const float *noise, uint64_t seed, uint64_t counter0, float reward_scale,
                             float *out_states, float *out_actions, float *out_logprobs, float *out_rewards,
                             uint8_t *out_undones, uint8_t *out_unmasks, float *out_values, float *out_next_value,
                             float *out_last_state, float *out_advantages, float *out_reward_sums, double *gae_stats,
                             double *gae_workspace, int64_t gae_workspace_bytes, float gamma, float lambda_gae,
                             int use_v_trace, void *stream);

/* K6  one PPO minibatch: gather (K5 indices) + critic fwd/bwd + actor fwd/bwd, both networks in one
 * launch.  Replaces AgentPPO.update_objectives up to (not including) the two optimizer steps
 * (AgentPPO.py:173-204) and ActorPPO.get_logprob_entropy (:378-386):
 *   obj_critic = mean((cri(s) - reward_sum)^2 * unmask)
 *   ratio = exp(logp_new - logp_old); surrogate = adv*ratio*where(adv > 0, 1-clip, 1+clip)
 *   actor loss = -(mean(surrogate*unmask) - lambda_entropy*mean(entropy*unmask))
 * Every workgroup owns 128 samples and leaves its gradient as one partial sum in `slabs`
 * (n_slabs x erl_ppo_slab_stride floats, n_slabs == erl_ppo_num_slabs(B) = ceil(B / 128)), to be summed in a
 * fixed order by erl_grad_reduce_f32.  inv_batch = 1/B (1/(B*world) is folded into K7 under data parallelism).
 * Slab / flat-gradient layout: [actor grads (Pa)] [critic grads (Pc)] [obj_critic, obj_surrogate,
 * obj_entropy, 0] [zeros up to erl_ppo_slab_stride] where Pa/Pc = erl_mlp_param_count(...) and erl_ppo_slab_stride =
 * Pa + Pc + 4 rounded up to 32 floats: rows start on 128-byte lines (the slab stores are non-temporal, whole lines). */
/* `objective` of the PPO minibatch entry points: which actor objective is differentiated (csrc/ppo_objective.h) */
#define ERL_PPO_OBJ_REFERENCE 0   /* AgentPPO.py:199   surrogate = adv*ratio*where(adv > 0, 1-clip, 1+clip) */
#define ERL_PPO_OBJ_CANONICAL 1   /* helloworld_PPO_single_file.py:337-339   min(adv*ratio, adv*clamp(ratio, 1-clip, 1+clip)) */
#define ERL_PPO_OBJ_A2C 2         /* AgentA2C.update_objectives (AgentPPO.py:296-303)   mean(adv * logp_a), no clip / mask / entropy */
/* Arithmetic of K6's five large products (both forward layers, the backward through W2, dW1, dW2).  PER CALL since ABI 17: the
 * `objective` argument of erl_ppo_step_f32 / erl_ppo_update_f32 / erl_ppo_update_dp_f32 is a mode word, ERL_PPO_MODE(objective, arith) =
 * objective | arith << 8; arith = ERL_PPO_ARITH_AUTO (a bare objective) takes the process-wide default that erl_ppo_set_arith sets, so
 * two agents of one process no longer share a setting:
 *   ERL_PPO_ARITH_F32    v_mfma_f32_32x32x2_f32 on fp32 operands;
 *   ERL_PPO_ARITH_SPLIT  every fp32 operand split into three bf16 parts (exactly: h + m + l == x), six partial products per
 *                        product on v_mfma_f32_32x32x16_bf16, fp32 accumulation -- fp32-equivalent (csrc/ppo_step_s3_impl.h);
 *                        shapes it does not cover take the fp32 kernel;
 *   ERL_PPO_ARITH_AUTO   the library default (environment ERL_K6_ARITH=f32|split overrides it).
 * Returns the previous setting. */
#define ERL_PPO_ARITH_AUTO 0
#define ERL_PPO_ARITH_F32 1
#define ERL_PPO_ARITH_SPLIT 2
#define ERL_PPO_MODE(objective, arith) ((objective) | ((arith) << 8))
ERL_API int erl_ppo_set_arith(int arith);
ERL_API int erl_ppo_arith_in_use(int S, int h1, int h2, int A);   /* ERL_PPO_ARITH_F32 or _SPLIT for this shape under the current setting */
/* Where the split-arithmetic minibatch kernels' workgroups run (csrc/ppo_step.h, k6_wg_map).  Workgroups go to the 8 XCDs round-robin by
 * linear id and, inside an XCD, to its 4 shader engines round-robin: map 0 (network = blockIdx.y) puts both networks' workgroups -- two
 * ~55 KB code paths -- behind every 64 KB instruction cache; map 1 = the actor's workgroups on XCDs 0-3, the critic's on 4-7; map 2 = the
 * actor's on shader engines 0-1 of every XCD, the critic's on engines 2-3 (one code path per instruction cache, every XCD still 16 + 16).
 * Results are bit-identical; which is faster depends on the box (about one in four of the pool has a slow instruction-cache miss path:
 * DESIGN.md section 4 "Instruction fetch and the workgroup map"; profiles/HISTORY.md "K6 in round 5"), so the first full-chip launch on a device measures map 0 against map 2 (back to back, the call's own
 * arguments, ~0.5 ms once per device and process) and the device keeps map 2 if it is 3 % faster.  ERL_K6_WG_MAP=0|1|2 forces a map.
 * *map = the map in use on `device` (-1: not decided yet), *us_map0 / *us_map2 = the per-launch times the decision saw (0: none).
 * The (256, h2[, h3]) kernels decide for themselves (their code is 125 KB per network): device | ERL_PPO_WG_FAMILY_WIDE asks for theirs. */
#define ERL_PPO_WG_FAMILY_WIDE 0x100
ERL_API int erl_ppo_wg_map_info(int device, int *map, double *us_map0, double *us_map2);
/* The form the last erl_ppo_update_f32 / erl_ppo_update_dp_f32 of this process took (ABI 19): 1 = one chain of launches per minibatch
 * (minibatch kernel over both networks, slab reduction, clip + Adam), 2 = TWO chains -- the actor's and the critic's minibatches share
 * nothing (own gradient, own clip norm, own Adam step: elegantrl/agents/AgentPPO.py:196-204, AgentBase.py:239-248), so each network runs
 * minibatch kernel -> slab reduction -> clip + Adam as its own chain of half-chip launches, the critic's on a library-owned second
 * stream forked from / joined into the caller's; 0 = no loop yet.  Bit-identical parameters either way.  Two chains are the default for
 * a single process on the split-arithmetic (128 | 64, h2) kernels where the device keeps workgroup map 0; ERL_PPO_CHAINS=1 forces one. */
ERL_API int erl_ppo_update_chains(void);
ERL_API int64_t erl_ppo_slab_stride(int S, int h1, int h2, int A);
ERL_API int erl_ppo_num_slabs(int64_t B);
ERL_API int erl_ppo_step_f32(const float *actor_params, const float *critic_params, const float *act_avg,
                     const float *act_std, const float *cri_avg, const float *cri_std, int S, int h1, int h2,
                     int A, const float *states, const float *actions, const uint8_t *unmasks,
                     const float *logprobs, const float *advantages, const float *reward_sums, int64_t H,
                     int64_t N, const int64_t *ids, int64_t B, float ratio_clip, float lambda_entropy,
                     float inv_batch, int objective, float *slabs, int n_slabs, void *stream);
ERL_API int erl_grad_reduce_f32(const float *slabs, int n_slabs, int64_t stride, float *flat_grad, void *stream);

/* K7  optimizer_backward's tail (elegantrl/agents/AgentBase.py:246-248) for up to 4 parameter groups in
 * one launch: per group, global-L2-norm clip (clip_grad_norm_: coef = min(1, max_norm/(norm+1e-6)))
 * then torch.optim.Adam defaults (betas 0.9/0.999, eps 1e-8).  Group g covers
 * [group_off[g], group_off[g] + group_len[g]) of the flat params/grads/m/v buffers (host arrays).
 * grad_scale multiplies every gradient first (1/world_size after an all-reduce SUM).
 * step = (step_base ? *step_base : 0) + step_offset is the 1-based Adam step of this call. */
ERL_API int erl_clip_adam_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq,
                      const int64_t *group_off, const int64_t *group_len, int n_groups, const int32_t *step_base,
                      int32_t step_offset, float lr, float beta1, float beta2, float eps, float max_norm,
                      float grad_scale, void *stream);

/* The optimiser tail of a PPO minibatch as TWO launches (csrc/grad_tail.hip; the default of erl_ppo_update_f32 /
 * erl_ppo_update_dp_f32 since round 3).  optimizer_backward = zero_grad / backward / clip_grad_norm_ / Adam.step
 * (elegantrl/agents/AgentBase.py:239-248); the backward's per-workgroup slabs come from erl_ppo_step_f32.
 *   launch 1  erl_grad_reduce_partials_f32: erl_grad_reduce_f32's sum (same association, bit for bit) + every 256-element
 *             workgroup's fp64 share of each parameter group's squared norm of (grad * grad_scale), kept in a library-owned
 *             per-(device, stream) table for the next launch on the same stream;
 *             erl_grad_sq_partials_f32: the partial norms alone, of a gradient row that is already summed (after a foreign
 *             all-reduce: RCCL / torch.distributed routes);
 *             erl_comm_reduce_exchange_f32: the reduction with the data-parallel exchange INSIDE the kernel when `comm` is a
 *             peer-to-peer communicator (push to the peers' stages, per-workgroup flags, rank-ordered sum: one launch),
 *             reduce -> ncclAllReduce -> partial norms on an RCCL communicator, erl_grad_reduce_partials_f32 for NULL;
 *   launch 2  erl_clip_adam_partials_f32: clip_grad_norm_ from the <= ceil(stride / 256) partial norms (fixed order), Adam
 *             on one element per thread.  `step` is the 1-based Adam step.
 * Every route through these leaves bit-identical parameters for the same summed gradient. */
ERL_API int erl_grad_reduce_partials_f32(const float *slabs, int n_slabs, int64_t stride, float *flat_grad,
                                 const int64_t *group_off, const int64_t *group_len, int n_groups, float grad_scale,
                                 void *stream);
ERL_API int erl_grad_sq_partials_f32(float *grads, int64_t stride, const int64_t *group_off, const int64_t *group_len,
                             int n_groups, float grad_scale, void *stream);
ERL_API int erl_clip_adam_partials_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t stride,
                               const int64_t *group_off, const int64_t *group_len, int n_groups, int32_t step, float lr,
                               float beta1, float beta2, float eps, float max_norm, float grad_scale, void *stream);
/* ... with a communicator: when an exchange on a peer-to-peer `comm` timed out earlier in this update loop (a peer missing:
 * the bounded wait of erl_comm_reduce_exchange_f32 / erl_comm_allreduce_sum_* gave up and its sums are garbage), the
 * communicator is poisoned and this launch SKIPS the update -- parameters and moments untouched -- until the host has read
 * the fault (erl_async_fault_count(1) reports "peer-to-peer gradient exchange" and clears the poison).  NULL / RCCL `comm`:
 * identical to erl_clip_adam_partials_f32. */
ERL_API int erl_comm_clip_adam_partials_f32(void *comm, float *params, const float *grads, float *exp_avg, float *exp_avg_sq,
                                    int64_t stride, const int64_t *group_off, const int64_t *group_len, int n_groups,
                                    int32_t step, float lr, float beta1, float beta2, float eps, float max_norm,
                                    float grad_scale, void *stream);

/* update_net's three returned objectives (AgentPPO.py:168-171): out3[j] = scale * mean over the n_rows minibatch rows of
 * grad_rows[k][offset + j] (the logged values K6 leaves behind the gradient; offset = Pa + Pc), one launch. */
ERL_API int erl_ppo_logs_mean_f32(const float *grad_rows, int64_t stride, int64_t offset, int n_rows, float scale, float *out3,
                          void *stream);
/* ... and, in the same launch, the side effect AgentPPO.get_advantages has on the caller's buffers (AgentPPO.py:211-214):
 * rewards[~unmasks] += values[~unmasks]; undones[~unmasks] = False over `total` = H * N elements.  The last launch of
 * update_net when the advantages came from the rollout's epilogue (which leaves rewards / undones untouched). */
ERL_API int erl_ppo_finish_f32(const float *grad_rows, int64_t stride, int64_t offset, int n_rows, float scale, float *out3,
                       float *rewards, uint8_t *undones, const uint8_t *unmasks, const float *values, int64_t total,
                       void *stream);

/* erl_grad_reduce_f32 + erl_clip_adam_f32 in ONE launch (host step only), for loops with nothing between the two: the same
 * gradient bit for bit (same summation order), written to flat_grad; the workgroup that finishes last derives the clip
 * coefficients from per-workgroup fp64 partial norms (fixed order) and applies Adam.  No workgroup waits on another. */
ERL_API int erl_reduce_clip_adam_f32(const float *slabs, int n_slabs, int64_t stride, float *flat_grad, float *params,
                             float *exp_avg, float *exp_avg_sq, const int64_t *group_off, const int64_t *group_len,
                             int n_groups, int32_t step, float lr, float beta1, float beta2, float eps, float max_norm,
                             float grad_scale, void *stream);

/* The same in the grid-wait form: every workgroup waits for the last partial norm (device counter) and updates its own
 * 256 elements from registers -- one launch, no single-workgroup Adam phase.  Valid only when the whole launch is resident
 * at once: erl_reduce_clip_adam_grid_ok(stride) (occupancy x compute units >= ceil(stride / 256)); EINVAL otherwise.  The
 * wait is bounded and reports through erl_async_fault_count.  Default optimiser tail of erl_ppo_update_f32 when it applies
 * (ERL_FUSED_TAIL=0 restores the two launches). */
ERL_API int erl_reduce_clip_adam_grid_ok(int64_t stride);
ERL_API int erl_reduce_clip_adam_grid_f32(const float *slabs, int n_slabs, int64_t stride, float *flat_grad, float *params,
                             float *exp_avg, float *exp_avg_sq, const int64_t *group_off, const int64_t *group_len,
                             int n_groups, int32_t step, float lr, float beta1, float beta2, float eps, float max_norm,
                             float grad_scale, void *stream);

/* The default two-launch tail (erl_grad_reduce_partials_f32 + erl_clip_adam_partials_f32) as ONE launch with the same bits (ABI 17,
 * csrc/grad_tail.hip tail_fused_kernel): the partial norms are published by agent-scope stores into a table double buffered by the
 * launch's parity and are their own flags (an unwritten entry holds a sentinel NaN); every workgroup polls them -- no arrival counter,
 * no fence, no second pass --, sums them in erl_clip_adam_partials_f32's order and applies clip + Adam to the 256 elements it has just
 * reduced, from registers.  Needs erl_tail_fused_ok(stride): rows up to 131 072 floats, every workgroup of the launch resident at once.
 * The wait is bounded; a workgroup whose wait times out SKIPS its 256 elements' update and reports through erl_async_fault_count -- the
 * others may already have stepped theirs: after such a fault the optimiser state is partially updated and must be restored, not used.  Single process only (a data-parallel
 * rank's exchange keeps the two launches).  erl_ppo_update_f32 uses it under ERL_FUSED_TAIL=3. */
ERL_API int erl_tail_fused_ok(int64_t stride);
ERL_API int erl_reduce_clip_adam_fused_f32(const float *slabs, int n_slabs, int64_t stride, float *flat_grad, float *params,
                             float *exp_avg, float *exp_avg_sq, const int64_t *group_off, const int64_t *group_len,
                             int n_groups, int32_t step, float lr, float beta1, float beta2, float eps, float max_norm,
                             float grad_scale, void *stream);

/* Whole PPO update in one call (single-process path): for k in [0, update_times):
 *   erl_ppo_step_f32(ids + k*B) -> erl_grad_reduce_f32 -> grads[k] -> erl_clip_adam_f32(step = first_step + k).
 * Replaces the minibatch loop of AgentPPO.update_net (AgentPPO.py:158-167).  flat_params / exp_avg / exp_avg_sq hold
 * [actor (Pa) | critic (Pc)]; grads: (update_times, erl_ppo_slab_stride) -- row k keeps minibatch k's summed gradient
 * and its three objective values in the tail; slabs: (erl_ppo_num_slabs(B), erl_ppo_slab_stride) scratch. */
ERL_API int erl_ppo_update_f32(float *flat_params, float *exp_avg, float *exp_avg_sq, const float *act_avg,
                       const float *act_std, const float *cri_avg, const float *cri_std, int S, int h1, int h2, int A,
                       const float *states, const float *actions, const uint8_t *unmasks, const float *logprobs,
                       const float *advantages, const float *reward_sums, int64_t H, int64_t N, const int64_t *ids,
                       int64_t B, int update_times, float ratio_clip, float lambda_entropy, int objective, float *slabs,
                       float *grads, int32_t first_step, float lr, float beta1, float beta2, float eps,
                       float max_norm, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Data-parallel exchange: one process per GPU, env shards one per rank, and ONE collective on the path --
 * the SUM all-reduce of the flat [actor | critic | 3 logged objectives] gradient row (203 KB at config 4) per minibatch,
 * enqueued by RCCL on the caller's stream between erl_grad_reduce_f32 and erl_clip_adam_f32.  The reference has no
 * collective here (its multi-GPU mode ships rollout data through host pipes, elegantrl/train/run.py:305-320).
 * RCCL is bound with dlopen at first use; without it these entry points return an error and everything else works.
 *   rank 0: erl_comm_unique_id(id) -> ship the ERL_COMM_ID_BYTES to every rank out of band (torch.distributed store)
 *   all   : erl_comm_init(id, rank, world, &comm)   (collective; binds the calling thread's current HIP device)
 * The communicator is library-owned state behind an opaque handle; destroy it before process exit. */
#define ERL_COMM_ID_BYTES 128
ERL_API int erl_comm_unique_id(uint8_t *out_id);
ERL_API int erl_comm_init(const uint8_t *id_bytes, int rank, int world_size, void **out_comm);
ERL_API int erl_comm_destroy(void *comm);
ERL_API int erl_comm_world_size(void *comm);
ERL_API int erl_comm_allreduce_sum_f32(void *comm, float *buf, int64_t count, void *stream);

/* One-shot peer-to-peer communicator (SURVEY 8e "better" option): the same handle type as above, so every erl_comm_* entry
 * point and erl_ppo_update_dp_f32 take it unchanged.  Every rank keeps a flag table [sender][workgroup] and a stage of
 * 2 halves x world rows x max_count floats in uncached device memory that the peers map through HIP IPC.  An exchange is
 * part of ONE launch (csrc/grad_tail.hip): a rank pushes its reduced slice into its row of every peer's stage (posted
 * remote stores over xGMI), raises a per-workgroup sequence flag on every peer, polls its own flags and sums the rows of
 * its own memory in rank order -- bit-identical sums on every rank, no round trip over the fabric.  world_size <= 8.
 *   all   : erl_comm_p2p_create(rank, world, max_count, &comm, handle)   (handle: ERL_P2P_HANDLE_BYTES, a hipIpcMemHandle_t)
 *   ship every rank's handle to every rank out of band (torch.distributed), rank-major
 *   all   : erl_comm_p2p_connect(comm, handles)                          (maps the peers' stages)
 * A peer that never publishes is reported through erl_async_fault_count (bounded wait), not a hang.  The Python side
 * selects this route only after a start-up self-test against RCCL passed on every rank (elegantrl_amd/parallel.py). */
#define ERL_P2P_HANDLE_BYTES 64
ERL_API int erl_comm_p2p_create(int rank, int world_size, int64_t max_count, void **out_comm, uint8_t *out_handle);
ERL_API int erl_comm_p2p_connect(void *comm, const uint8_t *handles);
/* polls (about a microsecond each) before a wait for a peer gives up and reports a fault; 0 = the default (2^24).  The
 * start-up self-test lowers it so that a route that does not work costs seconds, not a hang. */
ERL_API int erl_comm_p2p_set_spin(void *comm, uint32_t spins);

#define ERL_COMM_KIND_RCCL 0
#define ERL_COMM_KIND_P2P 1
ERL_API int erl_comm_kind(void *comm);                                                  /* -1 for NULL */
/* SUM all-reduce of doubles (the 5 advantage sums of AgentPPO.py:149 under data parallelism) on either kind */
ERL_API int erl_comm_allreduce_sum_f64(void *comm, double *buf, int64_t count, void *stream);
ERL_API int erl_comm_reduce_exchange_f32(void *comm, const float *slabs, int n_slabs, int64_t stride, float *flat_grad,
                                 const int64_t *group_off, const int64_t *group_len, int n_groups, float grad_scale,
                                 void *stream);

/* erl_ppo_update_f32 with the gradient all-reduce in the loop: ppo_step -> grad_reduce -> all-reduce(grads[k]) ->
 * clip_adam(grad_scale = 1/world), as the two launches erl_comm_reduce_exchange_f32 + erl_clip_adam_partials_f32.
 * comm == NULL degenerates to the single-process loop.  Every rank must call it with
 * the same update_times; ids are this rank's own minibatch indices into its own rollout shard.
 * adv_stats: NULL when `advantages` are normalised already (erl_adv_normalize_f32); else the 8-double block of raw sums
 * (erl_gae_scan_f32 / the rollout epilogue, all-reduced under data parallelism) and `advantages` are RAW: every minibatch
 * kernel applies (adv - mean) / (std(adv[::4, ::4]) + 1e-5) (AgentPPO.py:149) at its row load, in erl_adv_normalize_f32's
 * arithmetic -- one launch less per update.  adv_partials (n_partials x 3 doubles, the rollout epilogue's gae_workspace):
 * the sums are not folded yet; the loop folds them into adv_stats first (inside its weight-image launch where it has one). */
ERL_API int erl_ppo_update_dp_f32(float *flat_params, float *exp_avg, float *exp_avg_sq, const float *act_avg,
                          const float *act_std, const float *cri_avg, const float *cri_std, int S, int h1, int h2, int A,
                          const float *states, const float *actions, const uint8_t *unmasks, const float *logprobs,
                          const float *advantages, const float *reward_sums, int64_t H, int64_t N, const int64_t *ids,
                          int64_t B, int update_times, float ratio_clip, float lambda_entropy, int objective, float *slabs,
                          float *grads, int32_t first_step, float lr, float beta1, float beta2, float eps,
                          float max_norm, double *adv_stats, const double *adv_partials, int n_partials, void *comm,
                          void *stream);

/* ---------------------------------------------------------------------------------------------
 * Generic-shape path: build_mlp([S, d1, ..., dL, out]) with ANY number (<= ERL_MAX_LAYERS) and width of hidden layers
 * (elegantrl/agents/AgentBase.py:345-360; the reference's demos use (256, 128), (256, 128, 64), (256, 128, 128)).
 * dims = [S, d1, ..., dL, out], n_dims = L + 2.  Parameter block: W1 b1 ... WL bL Wout bout (+ action_std_log).
 * Dense layers are the library's own fp32 MFMA GEMMs with fused bias / GELU / gate epilogues; activations live in
 * `workspace` (erl_mlpn_workspace_bytes(dims, n_dims, rows, training)).  Same arithmetic as K1 / K2 / K6:
 *   erl_mlpn_value_forward_f32  = erl_value_forward_f32,  erl_mlpn_rollout_step_f32 = erl_rollout_step_f32,
 *   erl_mlpn_ppo_step_f32       = erl_ppo_step_f32 + erl_grad_reduce_f32 (writes the summed gradient
 *                                 [actor | critic | obj_critic, obj_surrogate, obj_entropy, 0] straight to flat_grad).
 * erl_mlpn_ppo_step_f32 with dims = [S <= 64, 256, 128, 64 | 128, A <= 8] (two of the demos' networks) runs as ONE fused kernel on the
 * bf16 matrix pipe with fp32-equivalent split arithmetic (csrc/ppo_step_wd_impl.h) + an image build + the slab reduction, on
 * library-owned buffers -- same arguments, same result; `workspace` is not used then; ERL_WIDE_FUSED=0 in the environment keeps the
 * layered step.  erl_mlpn_rollout_step_f32 with dims = [S <= 64, 256, 32..128, (32..128,) A <= 16] and N <= 16384 runs as ONE launch too
 * (csrc/rollout_wide.hip: the latency form of K1 for a 256-wide first layer); same arguments, `workspace` not used then.
 * erl_mlpn_value_forward_f32 with dims = [S <= 64, 256, 32..128, (32..128,) 1], any number of rows: ONE launch as well (persistent 16-row
 * tiles, the weights split once and kept in registers); same arguments, `workspace` not used then.
 * ------------------------------------------------------------------------------------------- */
ERL_API int64_t erl_mlpn_param_count(const int *dims, int n_dims, int with_std_log);
ERL_API int64_t erl_mlpn_workspace_bytes(const int *dims, int n_dims, int64_t rows, int training);
ERL_API int erl_mlpn_value_forward_f32(const float *params, const float *state_avg, const float *state_std, const int *dims,
                               int n_dims, const float *states, int64_t rows, float *values, void *workspace,
                               int64_t workspace_bytes, void *stream);
ERL_API int erl_mlpn_rollout_step_f32(const float *actor_params, const float *state_avg, const float *state_std,
                              const int *dims, int n_dims, const float *state, int64_t N, const float *noise,
                              uint64_t seed, uint64_t counter, float *out_state_row, float *out_action_row,
                              float *out_logprob_row, float *out_action_env, void *workspace,
                              int64_t workspace_bytes, void *stream);
ERL_API int erl_mlpn_ppo_step_f32(const float *actor_params, const float *critic_params, const float *act_avg,
                          const float *act_std, const float *cri_avg, const float *cri_std, const int *actor_dims,
                          int n_dims, const float *states, const float *actions, const uint8_t *unmasks,
                          const float *logprobs, const float *advantages, const float *reward_sums, int64_t H,
                          int64_t N, const int64_t *ids, int64_t B, float ratio_clip, float lambda_entropy,
                          float inv_batch, int objective, float *flat_grad, void *workspace, int64_t workspace_bytes,
                          void *stream);

/* Discrete-action sibling (AgentDiscretePPO / ActorDiscretePPO, elegantrl/agents/AgentPPO.py:305-320, :393-422): the actor
 * block has no action_std_log (erl_mlpn_param_count(dims, n_dims, 0)), dims[n_dims-1] = number of actions (<= 64).
 * Rollout: a ~ Categorical(softmax(logits)) by inverse CDF with one U[0,1) per env (`uniform` (N) injected, or
 * Philox4x32-10 keyed by (seed, counter, env)); stores the action as int32 (the reference's rollout dtype,
 * AgentPPO.py:103), its log-prob, and the int64 copy that goes to env.step (convert_action_for_env, :420-422).
 * Update: actions (H, N) int32; log-prob / entropy as torch.distributions.Categorical(probs) computes them
 * (logits = log(clamp(p, eps, 1 - eps))); the entropy is state dependent here, so its gradient flows into the network. */
#define ERL_MAX_DISCRETE_ACTIONS 64
ERL_API int erl_mlpn_rollout_step_discrete_f32(const float *actor_params, const float *state_avg, const float *state_std,
                              const int *dims, int n_dims, const float *state, int64_t N, const float *uniform,
                              uint64_t seed, uint64_t counter, float *out_state_row, int32_t *out_action_row,
                              float *out_logprob_row, int64_t *out_action_env, void *workspace,
                              int64_t workspace_bytes, void *stream);
ERL_API int erl_mlpn_ppo_step_discrete_f32(const float *actor_params, const float *critic_params, const float *act_avg,
                          const float *act_std, const float *cri_avg, const float *cri_std, const int *actor_dims,
                          int n_dims, const float *states, const int32_t *actions, const uint8_t *unmasks,
                          const float *logprobs, const float *advantages, const float *reward_sums, int64_t H, int64_t N,
                          const int64_t *ids, int64_t B, float ratio_clip, float lambda_entropy, float inv_batch,
                          float *flat_grad, void *workspace, int64_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * SAC: everything AgentSAC.update_objectives does after ReplayBuffer.sample
 * (elegantrl/agents/AgentSAC.py:50-86) in ONE call: target computation with the tanh-Gaussian actor (:167-199) and the
 * TARGET critic ensemble (:243-259), critic step + soft update (AgentBase.py:270-278), temperature step, actor step --
 * three clip + Adam steps (AgentBase.py:239-248).  hidden = net_dims (n_hidden <= ERL_MAX_LAYERS), E = num_ensembles.
 * Parameter blocks: actor = build_mlp([S, *hidden]) (GELU after every layer) + Linear(hidden[-1], 2A);
 * critic/target = Linear(S + A, hidden[0]) | E x build_mlp([*hidden, 1]).  The batch tensors are what
 * erl_replay_sample_f32 returned (B rows); is_weight (B) / td_error_out (B), both optional, are prioritised replay's importance
 * weights in and per-sample td errors out (AgentSAC.py:58-62).  eps_next / eps_cur (B, A) inject the two rsample() draws (tests); NULL ->
 * Philox keyed by (seed, counter).  objs_out: device float[2] = (obj_critic, obj_actor).  step = 1-based Adam step.
 * erl_sac_explore_action_f32 = ActorSAC.get_action (:179-185) for the off-policy rollout; state_out (N, S), may be NULL: the
 * rollout's `states[t] = state` (AgentBase.py:145) from the same launch (ABI 16).
 * ------------------------------------------------------------------------------------------- */
/* ReplayBuffer.sample (elegantrl/train/replay_buffer.py:120-134) handed to erl_sac_update_ring_f32 instead of its result: the ring's five
 * arrays (max_size, num_seqs, .), the drawn ids (B,) int64, sample_len = cur_size - 1; out_ids0 / out_ids1 (B,) int64 receive
 * ids % sample_len / ids // sample_len (:124-125; may be NULL). */
typedef struct ErlRingSample {
    const float *buf_states, *buf_actions, *buf_rewards, *buf_undones, *buf_unmasks;
    int64_t max_size, num_seqs;
    const int64_t *ids;
    int64_t sample_len;
    int64_t *out_ids0, *out_ids1;
    int64_t row_floats;     /* ABI 18: 0 = the five planar arrays above; > 0 = the interleaved ring (erl_replay_row_floats): buf_states is
                             * the block's base, the other four pointers are ignored */
} ErlRingSample;

/* AgentModSAC (elegantrl/agents/AgentSAC.py:89-165) on the same step: ErlSacOptions names what differs from AgentSAC (ABI 17).
 *   actor_variant        ERL_SAC_ACTOR_SAC = ActorSAC (:167-199); ERL_SAC_ACTOR_FIX = ActorFixSAC (:201-243): encoder build_mlp([S, *hidden])
 *                        with a RAW last layer, two one-layer decoders = rows [0, A) (mean) | [A, 2A) (log_std) of the same Linear(h, 2A)
 *                        block, log_std clamped to [-20, 2], log-prob AT the sample with the softplus form of the tanh correction
 *   update_actor         0: the two-time-scale rule (:148-158) skips the actor this step -- objs_out[1] = nan, alpha still clamped
 *   actor_step           the actor optimiser's own 1-based Adam step (it steps only when the actor is updated); <= 0: `step`
 *   actor_target_params  soft_update(act_target, act, tau) after the actor's step (:156); NULL: none
 * These take the layered step (dense layers on the library's MFMA GEMMs); erl_sac_update_f32 == a NULL options pointer. */
#define ERL_SAC_ACTOR_SAC 0
#define ERL_SAC_ACTOR_FIX 1
typedef struct ErlSacOptions {
    int32_t actor_variant;
    int32_t update_actor;
    int32_t actor_step;
    int32_t reserved;
    float *actor_target_params;
} ErlSacOptions;

ERL_API int erl_sac_param_counts(int S, int A, const int *hidden, int n_hidden, int E, int64_t *actor_count,
                         int64_t *critic_count);
ERL_API int64_t erl_sac_workspace_bytes(int S, int A, const int *hidden, int n_hidden, int E, int64_t B);
ERL_API int erl_sac_update_f32(float *actor_params, float *critic_params, float *target_params, float *alpha_log,
                       float *actor_m, float *actor_v, float *critic_m, float *critic_v, float *alpha_m, float *alpha_v,
                       int S, int A, const int *hidden, int n_hidden, int E, const float *state, const float *action,
                       const float *reward, const float *undone, const float *unmask, const float *next_state,
                       const float *is_weight, float *td_error_out, const float *cum_reward, float lambda_fit_cum_r,
                       int64_t B, const float *eps_next, const float *eps_cur, uint64_t seed, uint64_t counter,
                       float gamma, float target_entropy, float tau, float lr, float beta1, float beta2, float eps_adam,
                       float max_norm, int32_t step, float *objs_out, void *workspace, int64_t workspace_bytes,
                       void *stream);
ERL_API int erl_sac_update_opt_f32(float *actor_params, float *critic_params, float *target_params, float *alpha_log,
                           float *actor_m, float *actor_v, float *critic_m, float *critic_v, float *alpha_m, float *alpha_v,
                           int S, int A, const int *hidden, int n_hidden, int E, const float *state, const float *action,
                           const float *reward, const float *undone, const float *unmask, const float *next_state,
                           const float *is_weight, float *td_error_out, const float *cum_reward, float lambda_fit_cum_r,
                           int64_t B, const float *eps_next, const float *eps_cur, uint64_t seed, uint64_t counter,
                           float gamma, float target_entropy, float tau, float lr, float beta1, float beta2, float eps_adam,
                           float max_norm, int32_t step, float *objs_out, void *workspace, int64_t workspace_bytes,
                           const ErlSacOptions *opt, void *stream);
/* The off-policy rollout of AgentBase._explore_vec_env (elegantrl/agents/AgentBase.py:130-170) on the device-resident SynVecEnv as ONE
 * launch: H x [ActorSAC.get_action (AgentSAC.py:179-185), states[t] = state, actions[t] = action, env.step, reward / flag stores], then
 * `rewards *= reward_scale` and the two logical_not -- out_undones / out_unmasks are !terminal / !truncate.  A 16-env tile per workgroup
 * for all H steps, the actor's weights in registers / LDS for the whole rollout; per step the arithmetic of erl_sac_explore_action_f32
 * followed by erl_synenv_step_f32, so the five tensors, the final state and the env's counters are bit-identical to the per-step loop
 * under the same Philox keys (seed, counter0 + t, env, action-dim) / injected noise (H, N, A).  out_last_state (N, S) may be NULL.
 * Needs erl_sac_rollout_synenv_supported(...) (the fused SAC step's dims, N <= 4096).  ABI 16. */
ERL_API int erl_sac_rollout_synenv_supported(int S, int A, const int *hidden, int n_hidden, int64_t N);
ERL_API int erl_sac_rollout_synenv_f32(const float *actor_params, int S, int A, const int *hidden, int n_hidden, float *env_state,
                               const float *Ws, const float *Wa, int32_t *step_count, int32_t *episode, int max_step,
                               uint64_t env_seed, int64_t N, int64_t H, const float *noise, uint64_t seed, uint64_t counter0,
                               float reward_scale, float *out_states, float *out_actions, float *out_rewards,
                               uint8_t *out_undones, uint8_t *out_unmasks, float *out_last_state, void *stream);
/* ... on the device-resident PendulumVecEnv (S = 3, A = 1; erl_pendulum_step_f32's dynamics, operation for operation) */
ERL_API int erl_sac_rollout_pendulum_f32(const float *actor_params, const int *hidden, int n_hidden, float *phys, float *obs,
                                 int32_t *step_count, int32_t *episode, int max_step, uint64_t env_seed, int64_t N, int64_t H,
                                 const float *noise, uint64_t seed, uint64_t counter0, float reward_scale, float *out_states,
                                 float *out_actions, float *out_rewards, uint8_t *out_undones, uint8_t *out_unmasks,
                                 float *out_last_state, void *stream);
/* erl_sac_update_f32 with ReplayBuffer.sample in front of it, from ONE call: state / action / reward / undone / unmask / next_state are
 * the (B, .) staging block the sample is written to (what erl_replay_sample_f32 would have produced: bit-identical), everything else
 * as erl_sac_update_f32.  In the fused step the gather rides in the step's first launch (under its weight loads: one launch and one
 * kernel boundary less per update); elsewhere it is erl_replay_sample_f32 followed by erl_sac_update_f32.  cum_reward must be NULL
 * (lambda_fit_cum_r needs ids0 / ids1 before the step: sample first, then erl_sac_update_f32). */
ERL_API int erl_sac_update_ring_f32(float *actor_params, float *critic_params, float *target_params, float *alpha_log,
                            float *actor_m, float *actor_v, float *critic_m, float *critic_v, float *alpha_m, float *alpha_v,
                            int S, int A, const int *hidden, int n_hidden, int E, const ErlRingSample *ring, float *state,
                            float *action, float *reward, float *undone, float *unmask, float *next_state, int64_t B,
                            const float *eps_next, const float *eps_cur, uint64_t seed, uint64_t counter, float gamma,
                            float target_entropy, float tau, float lr, float beta1, float beta2, float eps_adam, float max_norm,
                            int32_t step, float *objs_out, void *workspace, int64_t workspace_bytes, void *stream);
/* n_steps of the above from one call (AgentBase.update_net's loop, AgentBase.py:172-189): step t samples with ids_all[t B .. (t + 1) B),
 * is optimiser step step0 + t, keys its Philox noise with counter0 + t and writes its two objectives to objs_all[2 t], [2 t + 1].  ABI 18. */
ERL_API int erl_sac_update_ring_loop_f32(float *actor_params, float *critic_params, float *target_params, float *alpha_log,
                                 float *actor_m, float *actor_v, float *critic_m, float *critic_v, float *alpha_m, float *alpha_v,
                                 int S, int A, const int *hidden, int n_hidden, int E, const ErlRingSample *ring,
                                 const int64_t *ids_all, int64_t n_steps, float *state, float *action, float *reward, float *undone,
                                 float *unmask, float *next_state, int64_t B, uint64_t seed, uint64_t counter0, float gamma,
                                 float target_entropy, float tau, float lr, float beta1, float beta2, float eps_adam, float max_norm,
                                 int32_t step0, float *objs_all, void *workspace, int64_t workspace_bytes, void *stream);
ERL_API int erl_sac_explore_action_f32(const float *actor_params, int S, int A, const int *hidden, int n_hidden,
                               const float *state, int64_t N, const float *noise, uint64_t seed, uint64_t counter,
                               float *action_out, float *state_out, void *workspace, int64_t workspace_bytes, void *stream);
/* ... with the actor variant named (ERL_SAC_ACTOR_FIX: ActorFixSAC.get_action, AgentSAC.py:217-224).  ABI 17. */
ERL_API int erl_sac_explore_action_opt_f32(const float *actor_params, int S, int A, const int *hidden, int n_hidden,
                                   const float *state, int64_t N, const float *noise, uint64_t seed, uint64_t counter,
                                   float *action_out, float *state_out, void *workspace, int64_t workspace_bytes,
                                   int actor_variant, void *stream);

/* measurement hook (bench.py's `roofline`; no reference counterpart): every_nth > 0 makes erl_ppo_step_f32 time every n-th K6
 * launch (1 = every launch, 0 = off) two ways -- a HIP-event bracket on the launch stream (contains the dispatch and completion
 * overhead of the bracket itself, 3-15 us depending on the box) and the kernel's own span, first workgroup in to last workgroup
 * out, on the device's constant-rate clock (what rocprofv3's kernel duration measures).  erl_k6_timing_read2 waits for the
 * sampled launches, returns both sums (ms) and the number of launches, and clears the lists; erl_k6_timing_read is the
 * event sum alone.  erl_k6_timing_null_bracket_us brackets an EMPTY launch the same way (median of `reps`, microseconds): the
 * part of an event bracket that is not the kernel. */
ERL_API void erl_k6_timing_enable(int every_nth);
ERL_API int erl_k6_timing_read(double *total_ms, int *launches);
ERL_API int erl_k6_timing_read2(double *event_ms, double *span_ms, int *launches);
ERL_API int erl_k6_timing_null_bracket_us(void *stream, int reps, double *median_us);
/* What the launches drained by the LAST erl_k6_timing_read2 say about the box (ABI 17).  Of every `every_nth` K6 launches one sits
 * inside an event bracket and one more (half a period later) is sampled WITHOUT a bracket; both leave one record per workgroup (entry /
 * exit on the constant-rate clock and on the shader clock, phase stamps: plain stores, folded by the host -- a first version that folded
 * them on the device with atomics cost the kernel 6 us), every other launch runs untouched.  `bracketed` selects the group:
 * 0 = the launches WITHOUT a bracket (the kernel as the loop runs it), 1 = the bracketed ones.  span_ms / launches = summed
 * first-workgroup-in to last-workgroup-out spans and their count; shader_mhz = the clock the launches actually ran at (shader cycles
 * per tick of the constant-rate clock, summed over every workgroup's own entry-to-exit interval: the chip clocks to its power budget,
 * so two boxes -- or two kernels on one box -- do not run the same clock); workgroup_us = a workgroup's mean duration;
 * phase_cycles[0 .. *n_phases) (at most max_phases written; *n_phases = 0 for kernels that stamp no phases) = mean shader cycles per
 * phase of an actor workgroup's first wave in ppo_step_s3_kernel: prologue | first layer forward | second layer forward | output layer
 * + objective + backward | staging + dW1 | staging + dW3 + staging | dW2 + logs + store drain; phase_workgroups = the number of
 * workgroups the means are over.  Any pointer may be NULL. */
/* diagnostics: the raw per-workgroup records of the group's last sampled launch (8 uint64 each: entry / exit on the constant-rate clock,
 * entry / exit on the shader clock, three words of phase stamps, HW_REG_HW_ID | HW_REG_XCC_ID << 32 = where the workgroup ran); actor
 * workgroups first.  Returns the number of workgroups copied (<= max_workgroups). */
ERL_API int erl_k6_timing_last_records(int bracketed, unsigned long long *out, int max_workgroups);
/* every sampled launch of the group the last erl_k6_timing_read2 drained: launch_index[i] = the launch's number since
 * erl_k6_timing_enable (a caller that knows its update loop's length knows where in the loop the launch sat: the first launch of a loop
 * finds the instruction caches cold), span_us[i] = its span.  Returns the number of launches copied (<= max_launches). */
ERL_API int erl_k6_timing_spans(int bracketed, long long *launch_index, double *span_us, int max_launches);
/* The same hook for the other kernels of the hot path (ABI 17): after erl_kernel_span_enable(n), every n-th launch of a tagged kernel
 * leaves one {entry, exit} record per workgroup on the device's constant-rate clock (plain stores; 2 M workgroup records between
 * enables); erl_kernel_span_read(tag) waits for the device and returns the summed first-workgroup-in to last-workgroup-out spans
 * (microseconds) of the tag's sampled launches and their number.  What rocprofv3's kernel duration measures, without a profiler and
 * without an event bracket around the launch.  erl_kernel_span_enable(0) turns it off; enabling again clears the records. */
#define ERL_SPAN_GAE 0             /* gae_exact_kernel / gae_lookback_kernel (erl_gae_scan_f32) */
#define ERL_SPAN_REPLAY_SAMPLE 1   /* replay_sample_kernel (erl_replay_sample_f32) */
#define ERL_SPAN_SAC_CRITIC_TRAIN 2 /* critic_tile_kernel<1>: the fused SAC step's critic training pass */
#define ERL_SPAN_SLAB_REDUCE 3     /* reduce_exchange_kernel: the PPO minibatch's slab reduction (+ exchange) */
#define ERL_SPAN_CLIP_ADAM 4       /* clip_adam_partials_kernel */
#define ERL_SPAN_ROLLOUT 5         /* rollout_fused_kernel: the persistent PPO rollout */
#define ERL_SPAN_TAGS 8
ERL_API void erl_kernel_span_enable(int every_nth);
ERL_API int erl_kernel_span_read(int tag, double *total_us, int *launches);
ERL_API int erl_k6_timing_clocks(int bracketed, double *span_ms, int *launches, double *shader_mhz, double *workgroup_us,
                         double *phase_cycles, int max_phases, int *n_phases, int *phase_workgroups);

/* ---------------------------------------------------------------------------------------------
 * GPU-resident synthetic environments for measurement; they implement the
 * env protocol of elegantrl (step -> state, reward, terminal, truncate; auto-reset).
 * ------------------------------------------------------------------------------------------- */
/* SynVecEnv: s' = s*Ws + a*Wa; reward = -mean(s'^2) - 0.01*mean(a^2); terminal = max|s'| > 10;
 * truncate = step_count >= max_step && !terminal; done rows reset to N(0,1) (Philox by seed/env/episode). */
ERL_API int erl_synenv_step_f32(float *state, const float *action, const float *Ws, const float *Wa,
                        int32_t *step_count, int32_t *episode, float *reward, uint8_t *terminal,
                        uint8_t *truncate, int64_t N, int S, int A, int max_step, uint64_t seed, void *stream);
/* Pendulum-v1 dynamics with elegantrl/envs/CustomGymEnv.py:24-44 scaling (action*2, reward*0.5).
 * phys: (N, 2) = (theta, theta_dot); obs out: (N, 3) = (cos, sin, theta_dot). 200-step truncation. */
ERL_API int erl_pendulum_step_f32(float *phys, float *obs, const float *action, int32_t *step_count, int32_t *episode,
                          float *reward, uint8_t *terminal, uint8_t *truncate, int64_t N, int max_step,
                          uint64_t seed, void *stream);

/* self-test of the MFMA tile helpers against scalar code; returns 0 and writes max |err| to *max_err
 * (host pointer).  Needs a GPU. */
ERL_API int erl_selftest_mfma(float *max_err);

#ifdef __cplusplus
}
#endif
#endif /* ERL_HIP_H */
